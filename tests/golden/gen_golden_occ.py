#!/usr/bin/env python3
"""Golden vectors for the class-occupancy update, produced by RUNNING the reference's own function.

    python tests/golden/gen_golden_occ.py        (build container only)

Reference entry point exercised: src/pyp/analysis/occupancies.py:67-214  occupancy_extended(..., local=False) on a toy
3-class, 2-image SPA data set written with the reference's .cistem codec (stub `toml` / `jsonrpcclient` modules as in
gen_golden.py).  Inputs (per-class OCC / LOGP / SIGMA columns) and the OCC / SIGMA columns the reference writes back are
stored in occupancy_3class.npz; nothing of the reference's source is copied.
"""
import os
import sys
import tempfile
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/src"


def main():
    warnings.simplefilter("ignore")
    sys.dont_write_bytecode = True
    tmp = tempfile.mkdtemp(prefix="pypstub_")
    with open(os.path.join(tmp, "toml.py"), "w") as f:
        f.write("def load(*a, **k):\n    raise NotImplementedError('stub')\nloads = dump = dumps = load\n")
    with open(os.path.join(tmp, "jsonrpcclient.py"), "w") as f:
        f.write("class Error: pass\nclass Ok: pass\ndef parse(*a, **k):\n    raise NotImplementedError('stub')\nrequest = parse\n")
    sys.path[:0] = [tmp, REF]
    from pyp.inout.metadata import cistem_star_file as csf
    from pyp.analysis import occupancies as occ

    rng = np.random.default_rng(11)
    K, images, per_image, it = 3, ["imgA", "imgB"], [7, 5], 4
    M = sum(per_image)
    logp = rng.normal(-4000.0, 6.0, (K, M)).round(2)
    logp[1, 3] = logp[0, 3] - 25.0                       # beyond the delta < 10 window
    logp[2, 3] = logp[0, 3] - 9.5
    sigma = rng.uniform(0.6, 1.4, (K, M)).round(3)
    occ_in = np.stack([rng.uniform(10, 90, M).round(2) for _ in range(K)])
    score = rng.uniform(5, 30, (K, M)).round(3)
    work = tempfile.mkdtemp(prefix="occ_")
    dataset = "toy"
    for k in range(K):
        folder = os.path.join(work, "%s_r%02d_%02d" % (dataset, k + 1, it - 1))
        os.makedirs(folder)
        off = 0
        for name, n in zip(images, per_image):
            data = np.zeros((n, 32))
            data[:, 0] = np.arange(1, n + 1)
            data[:, 1:4] = rng.uniform(0, 360, (n, 3)).round(2)
            data[:, 11] = occ_in[k, off:off + n]
            data[:, 12] = logp[k, off:off + n]
            data[:, 13] = sigma[k, off:off + n]
            data[:, 14] = score[k, off:off + n]
            data[:, 15] = 1.0
            data[:, 16] = 300.0
            data[:, 17] = 2.7
            data[:, 18] = 0.07
            data[:, 26] = np.arange(n)                       # PIND
            p = csf.Parameters()
            particles = {i: csf.Particle(i, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, float(data[i, 11])) for i in range(n)}
            tilts = {0: {0: csf.Tilt(0, 0, 0, 0, 0.0, 0.0)}}
            ext = csf.ExtendedParameters()
            ext.set_data(particles=particles, tilts=tilts)
            p.set_data(data=data, extended_parameters=ext)
            base = os.path.join(folder, "%s_r%02d.cistem" % (name, k + 1))
            p.to_binary(base, extended_output=base.replace(".cistem", "_extended.cistem"))
            off += n
    parameters = {"refine_iter": it, "data_mode": "spr", "refine_score_weighting": False}
    cwd = os.getcwd()
    os.chdir(work)
    try:
        occ.occupancy_extended(parameters, dataset, K, image_list=images, parameter_file_folders=work, local=False)
    finally:
        os.chdir(cwd)
    occ_out = np.zeros((K, M)); sig_out = np.zeros((K, M))
    for k in range(K):
        folder = os.path.join(work, "%s_r%02d_%02d" % (dataset, k + 1, it - 1))
        off = 0
        for name, n in zip(images, per_image):
            d = csf.Parameters.from_file(os.path.join(folder, "%s_r%02d.cistem" % (name, k + 1))).get_data()
            occ_out[k, off:off + n] = d[:, 11]; sig_out[k, off:off + n] = d[:, 13]
            off += n
    np.savez(os.path.join(HERE, "occupancy_3class.npz"), logp=logp, sigma=sigma, occ_in=occ_in, occ_out=occ_out, sigma_out=sig_out,
             per_image=np.array(per_image))
    print("occupancy golden written:", occ_out[:, :4].round(3))


if __name__ == "__main__":
    main()
