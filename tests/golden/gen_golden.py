#!/usr/bin/env python3
"""Generate the golden format fixtures by RUNNING the reference's own Python I/O modules.

Run in the build container only (the reference never travels to the GPU box):

    python tests/golden/gen_golden.py

It imports /root/reference/src/pyp/... with two throw-away stub modules (`toml`,
`jsonrpcclient`, created in a temp dir, SURVEY.md §8c recipe) and writes small data
fixtures (inputs + expected outputs) next to this script. Nothing of the reference's
source is copied: the fixtures are bytes/numbers the reference produced.

Reference entry points exercised:
  src/pyp/inout/metadata/cistem_star_file.py:632-776   Parameters.set_data/to_binary/from_binary
  src/pyp/inout/metadata/cistem_star_file.py:244-381   ExtendedParameters.to_binary/from_binary
  src/pyp/inout/metadata/frealign_parfile.py:660-697   Parameters.write_parameter_file
  src/pyp/inout/metadata/frealign_parfile.py:1481-1789 Parameters.from_file / format_from_parfile
  src/pyp/inout/image/mrc.py:537-560                   write / read / readHeaderFromFile
  src/pyp/analysis/geometry/core.py:211-234            get_degrees_from_matrix
  src/pyp/system/project_params.py:362-373             param() colon schedules
  src/pyp/analysis/image.py:320-417                    extract_background / normalize_image
"""
import json
import os
import sys
import tempfile
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/src"


def _stubs(tmp):
    with open(os.path.join(tmp, "toml.py"), "w") as f:
        f.write("def load(*a, **k):\n    raise NotImplementedError('stub')\nloads = dump = dumps = load\n")
    with open(os.path.join(tmp, "jsonrpcclient.py"), "w") as f:
        f.write("class Error: pass\nclass Ok: pass\ndef parse(*a, **k):\n    raise NotImplementedError('stub')\nrequest = parse\n")


def main():
    warnings.simplefilter("ignore")
    sys.dont_write_bytecode = True
    tmp = tempfile.mkdtemp(prefix="pypstub_")
    _stubs(tmp)
    sys.path[:0] = [tmp, REF]
    from pyp.inout.metadata import cistem_star_file as csf
    from pyp.inout.metadata import frealign_parfile as fp
    from pyp.inout.image import mrc as rmrc
    from pyp.analysis.geometry import core as geo
    from pyp.system import project_params as pp

    rng = np.random.default_rng(7)
    out = {}

    # ---- (1) .cistem main file: 8 rows x 32 columns ---------------------------------
    rows = 8
    data = np.zeros((rows, 32), dtype=np.float64)
    data[:, 0] = np.arange(1, rows + 1)                        # POSITION_IN_STACK
    data[:, 1:4] = rng.uniform(0, 360, (rows, 3)).round(3)     # PSI THETA PHI
    data[:, 4:6] = rng.normal(0, 3, (rows, 2)).round(3)        # shifts (A)
    data[:, 6] = rng.uniform(8000, 24000, rows).round(1)
    data[:, 7] = data[:, 6] + rng.uniform(-300, 300, rows).round(1)
    data[:, 8] = rng.uniform(0, 180, rows).round(2)
    data[:, 9] = 0.0
    data[:, 10] = np.arange(rows) // 3                          # film index
    data[:, 11] = 100.0
    data[:, 12] = -rng.uniform(100, 5000, rows).round(0)
    data[:, 13] = 0.5
    data[:, 14] = rng.uniform(0, 30, rows).round(4)
    data[:, 15] = 1.08
    data[:, 16] = 300.0
    data[:, 17] = 2.7
    data[:, 18] = 0.07
    data[:, 23] = rng.integers(100, 4000, rows)
    data[:, 24] = rng.integers(100, 4000, rows)
    data[:, 26] = np.arange(rows)                               # PIND
    data[:, 27] = np.arange(rows) % 3                           # TIND
    p = csf.Parameters()
    p.set_data(data=data)
    path = os.path.join(HERE, "params8.cistem")
    p.to_binary(output=path)
    back = csf.Parameters.from_file(path).get_data()
    np.save(os.path.join(HERE, "params8_data.npy"), data)
    np.save(os.path.join(HERE, "params8_readback.npy"), back)
    out["cistem_main_bytes"] = os.path.getsize(path)

    # ---- (1b) extended file ---------------------------------------------------------
    particles = {i: csf.Particle(i, 0.5 * i, -0.25 * i, 1.0 * i, 10.0 * i, 20.0 * i, 30.0 * i,
                                 100.0 + i, 200.0 + i, 50.0 + i, 3.5 * i, 100.0)
                 for i in range(4)}
    tilts = {t: {0: csf.Tilt(t, 0, 0.1 * t, -0.2 * t, -60.0 + 3 * t, 85.3)} for t in range(3)}
    ext = csf.ExtendedParameters()
    ext.set_data(particles=particles, tilts=tilts)
    epath = os.path.join(HERE, "params8_extended.cistem")
    ext.to_binary(epath)
    e2 = csf.ExtendedParameters.from_file(epath)
    out["ext_particles"] = {str(k): [v.particle_index, v.shift_x, v.shift_y, v.shift_z, v.psi, v.theta, v.phi,
                                     v.x_position_3d, v.y_position_3d, v.z_position_3d, v.score, v.occ]
                            for k, v in e2.get_particles().items()}
    out["ext_tilts"] = {str(t): {str(r): [v.tilt_index, v.region_index, v.shift_x, v.shift_y, v.angle, v.axis]
                                 for r, v in d.items()} for t, d in e2.get_tilts().items()}

    # ---- (2) .par text in the four writer templates ------------------------------------
    npar = 5
    base16 = np.zeros((npar, 16))
    base16[:, 0] = np.arange(1, npar + 1)
    base16[:, 1:4] = rng.uniform(0, 360, (npar, 3))
    base16[:, 4:6] = rng.normal(0, 5, (npar, 2))
    base16[:, 6] = 10000
    base16[:, 7] = [0, 0, 1, 1, 2]
    base16[:, 8] = rng.uniform(8000, 24000, npar)
    base16[:, 9] = base16[:, 8] + 100
    base16[:, 10] = rng.uniform(0, 180, npar)
    base16[:, 11] = 100
    base16[:, 12] = -rng.integers(100, 9000, npar)
    base16[:, 13] = 0.5
    base16[:, 14] = rng.uniform(0, 40, npar)
    base16[:, 15] = rng.normal(0, 1, npar)
    ext29 = np.zeros((npar, 29))
    ext29[:, 0] = np.arange(npar)
    ext29[:, 1] = rng.uniform(-60, 60, npar)
    ext29[:, 2] = rng.uniform(0, 100, npar)
    ext29[:, 3] = np.arange(npar) % 4
    ext29[:, 4:6] = rng.uniform(0, 1, (npar, 2))
    ext29[:, 6:26] = rng.normal(0, 1, (npar, 20))
    ext29[:, 26:29] = rng.uniform(0, 360, (npar, 3))
    base17 = np.insert(base16, 11, 0.0, axis=1)     # PSHIFT after ANGAST
    par_inputs = {
        "new": base16,
        "new_ext": np.hstack([base16, ext29]),
        "frealignx": base17,
        "frealignx_ext": np.hstack([base17, ext29]),
    }
    for key, arr in par_inputs.items():
        fn = os.path.join(HERE, f"par_{key}.par")
        fp.Parameters.write_parameter_file(fn, arr, parx=key.endswith("_ext"), frealignx=key.startswith("frealignx"))
        np.save(os.path.join(HERE, f"par_{key}_in.npy"), arr)
        rd = fp.Parameters.from_file(fn)
        np.save(os.path.join(HERE, f"par_{key}_readback.npy"), np.asarray(rd.data, dtype=np.float64))
        out[f"par_{key}_version"] = [rd.version, bool(rd.extended)]

    # ---- (3) MRC stack 4 x 8 x 8, float32 ----------------------------------------------
    stack = rng.normal(0, 1, (4, 8, 8)).astype(np.float32)
    mpath = os.path.join(HERE, "stack_4x8x8.mrc")
    rmrc.write(stack, mpath)
    np.save(os.path.join(HERE, "stack_4x8x8.npy"), stack)
    rb = rmrc.read(mpath)
    assert np.array_equal(rb, stack)
    h = rmrc.readHeaderFromFile(mpath)
    out["mrc_header"] = {k: (float(h[k]) if "float" in str(type(h[k])) else int(h[k]))
                         for k in ("nx", "ny", "nz", "mode", "mx", "my", "mz", "xlen", "ylen", "zlen",
                                   "mapc", "mapr", "maps", "amin", "amax", "amean", "rms", "byteorder", "nlabels")}

    # ---- (4) Euler convention: matrix -> (psi, theta, phi) ----------------------------------
    eul = []
    for psi, theta, phi in [(30, 40, 50), (0, 0, 0), (350, 179, 10), (123.4, 90, 271.5), (10, 1e-3, 20),
                            (200, 135, 45)]:
        c, s = np.cos, np.sin
        ps, th, ph = np.radians([psi, theta, phi])
        # the "left-handed" matrix written out at analysis/geometry/core.py:1194-1197
        m = np.array([
            [c(ph) * c(th) * c(ps) - s(ph) * s(ps), c(ph) * c(th) * s(ps) + s(ph) * c(ps), -c(ph) * s(th)],
            [-s(ph) * c(th) * c(ps) - c(ph) * s(ps), -s(ph) * c(th) * s(ps) + c(ph) * c(ps), s(ph) * s(th)],
            [s(th) * c(ps), s(th) * s(ps), c(th)]])
        got = geo.get_degrees_from_matrix(m)
        eul.append({"in": [psi, theta, phi], "matrix": m.tolist(), "out": [float(x) for x in got]})
    out["euler"] = eul

    # ---- (5) per-iteration colon schedules -----------------------------------------------
    out["param_schedule"] = {s: [pp.param(s, it) for it in range(2, 8)] for s in ("8:7:6", "4", "20:10")}

    # ---- (6) box normalisation of the extraction step (src/pyp/analysis/image.py:320-417) ---------
    from pyp.analysis import image as rimg
    boxes, outs, meta = [], [], []
    for boxsize, radius, pixel, binning in ((16, 10.0, 2.0, 1), (24, 30.0, 1.5, 2), (16, 100.0, 2.0, 1), (32, 20.0, 1.0, 1)):
        img = rng.normal(5.0, 2.0, (boxsize, boxsize))
        res = rimg.normalize_image(img.copy(), radius, pixel, binning)
        bg = rimg.extract_background(img.copy(), radius, pixel * binning)
        boxes.append(img); outs.append(res); meta.append([boxsize, radius, pixel, binning, float(bg[0]), float(bg[1])])
    np.savez(os.path.join(HERE, "normalize_image.npz"), **{f"in{i}": b for i, b in enumerate(boxes)}, **{f"out{i}": o for i, o in enumerate(outs)})
    out["normalize_image"] = meta

    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
