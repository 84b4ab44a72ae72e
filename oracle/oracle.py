"""ctypes front-end of the CPU oracle (oracle/ppm_oracle.c).

TEST INFRASTRUCTURE ONLY — imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg; never by pyp_amd/.  Config structs are the ones of include/ppm.h (any ctypes.Structure with that
layout is accepted and passed by reference).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libppm_oracle.so")
NCOL = 32


def build(force=False):
    src = os.path.join(_HERE, "ppm_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libppm_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.orc_reference_create.restype = C.c_void_p
        L.orc_reference_create.argtypes = [C.c_void_p, C.c_int, C.c_float]
        L.orc_reference_create_padded.restype = C.c_void_p
        L.orc_reference_create_padded.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_int]
        L.orc_reference_destroy.argtypes = [C.c_void_p]
        L.orc_refine_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_score_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_match_projections.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_preprocess.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]
        L.orc_band_dims.argtypes = [C.c_void_p] + [C.c_void_p] * 6
        L.orc_extract_slice.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_void_p]
        L.orc_symmetry_ops.argtypes = [C.c_char_p, C.c_void_p]
        L.orc_fft1d.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_insert_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_finalize.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_sva_align.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_sva_insert.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_csp_pose.argtypes = [C.c_double, C.c_double, C.c_void_p, C.c_void_p]
        L.orc_csp_pose.restype = None
        L.orc_csp_refine.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Reference:
    def __init__(self, vol, max_band_px, pad=1):
        vol = np.ascontiguousarray(vol, dtype=np.float32)
        self.n = vol.shape[0]
        self.h = lib().orc_reference_create_padded(_p(vol), self.n, float(max_band_px), int(pad))
        if not self.h:
            raise RuntimeError("oracle: reference_create failed")

    def close(self):
        if self.h and _lib is not None:
            _lib.orc_reference_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def band_dims(cfg):
    v = [C.c_int(), C.c_int(), C.c_int(), C.c_double(), C.c_int(), C.c_int()]
    if lib().orc_band_dims(C.byref(cfg), *[C.byref(x) for x in v]):
        raise ValueError("oracle: bad config")
    return dict(zip(("B", "n_orient", "Ns", "step", "RSx", "RSy"), [x.value for x in v]))


def refine_batch(ref, cfg, images, rows, ccf_mode=1):
    images = np.ascontiguousarray(images, dtype=np.float32)
    rows = np.ascontiguousarray(rows, dtype=np.float64)
    out = np.empty_like(rows)
    counts = np.zeros(3, dtype=np.int64)      # orientations, local score evaluations, local sample-evaluations (per particle)
    rc = lib().orc_refine_batch(ref.h, C.byref(cfg), _p(images), len(rows), _p(rows), _p(out), int(ccf_mode), _p(counts))
    if rc:
        raise RuntimeError(f"oracle: refine_batch failed ({rc})")
    return out, counts


def score_batch(ref, cfg, images, rows):
    images = np.ascontiguousarray(images, dtype=np.float32)
    rows = np.ascontiguousarray(rows, dtype=np.float64)
    sc = np.empty(len(rows), dtype=np.float64)
    rc = lib().orc_score_batch(ref.h, C.byref(cfg), _p(images), len(rows), _p(rows), _p(sc))
    if rc:
        raise RuntimeError(f"oracle: score_batch failed ({rc})")
    return sc


def match_projections(ref, cfg, rows):
    rows = np.ascontiguousarray(rows, dtype=np.float64)
    out = np.empty((len(rows), cfg.box, cfg.box), dtype=np.float32)
    rc = lib().orc_match_projections(ref.h, C.byref(cfg), _p(rows), len(rows), _p(out))
    if rc:
        raise RuntimeError(f"oracle: match_projections failed ({rc})")
    return out


def preprocess(cfg, img, mask_radius):
    d = band_dims(cfg)
    out = np.zeros((2 * d["B"] + 1, d["B"] + 1, 2), dtype=np.float32)
    img = np.ascontiguousarray(img, dtype=np.float32)
    wr = np.zeros(d["B"] + 2, dtype=np.float64)
    if lib().orc_preprocess(C.byref(cfg), _p(img), float(mask_radius), _p(out), _p(wr)):
        raise RuntimeError("oracle: preprocess failed")
    return out[..., 0] + 1j * out[..., 1], wr


def extract_slice(ref, cfg, psi, theta, phi):
    d = band_dims(cfg)
    out = np.zeros((2 * d["B"] + 1, d["B"] + 1, 2), dtype=np.float32)
    if lib().orc_extract_slice(ref.h, C.byref(cfg), psi, theta, phi, _p(out)):
        raise RuntimeError("oracle: extract_slice failed")
    return out[..., 0] + 1j * out[..., 1]


def symmetry_ops(sym):
    ops = np.zeros((60, 3, 3), dtype=np.float64)
    n = lib().orc_symmetry_ops(sym.encode(), _p(ops))
    if n < 1:
        raise ValueError(f"oracle: bad symmetry {sym}")
    return ops[:n]


def accum_floats(box):
    return 2 * box * box * (box // 2 + 1) * 3


def insert_batch(acc, counts, cfg, symmetry, images, rows):
    images = np.ascontiguousarray(images, dtype=np.float32)
    rows = np.ascontiguousarray(rows, dtype=np.float64)
    rc = lib().orc_insert_batch(_p(acc), _p(counts), C.byref(cfg), symmetry.encode(), _p(images), len(rows), _p(rows))
    if rc:
        raise RuntimeError(f"oracle: insert_batch failed ({rc})")


def finalize(acc, box, pixel, fcfg):
    h1 = np.empty((box, box, box), dtype=np.float32)
    h2 = np.empty_like(h1)
    fl = np.empty_like(h1)
    stats = np.zeros((box // 2 - 1, 7), dtype=np.float64)
    rc = lib().orc_finalize(_p(acc), box, float(pixel), C.byref(fcfg), _p(h1), _p(h2), _p(fl), _p(stats))
    if rc:
        raise RuntimeError(f"oracle: finalize failed ({rc})")
    return h1, h2, fl, stats


def fft1d(x, inverse=False):
    """The oracle's mixed-radix FFT on a complex vector (known-answer test against numpy.fft)."""
    a = np.ascontiguousarray(x, dtype=np.complex64).copy()
    if lib().orc_fft1d(_p(a), len(a), 1 if inverse else 0):
        raise ValueError("oracle: unsupported FFT length")
    return a


def csp_pose(tilt_angle, tilt_axis, particle):
    """(PSI, THETA, PHI, SHX, SHY) of a projection row from the tilt geometry and the stored particle parameters
    {psi, theta, phi, shift x, y, z} (restates csp_euler_angles, src/pyp/analysis/geometry/core.py:1081-1213)."""
    p = np.ascontiguousarray(particle, dtype=np.float64)
    out = np.zeros(5, dtype=np.float64)
    lib().orc_csp_pose(float(tilt_angle), float(tilt_axis), _p(p), _p(out))
    return out


def csp_refine(ref, cfg, csp_cfg, images, rows, particles, tilts):
    """Returns (rows, particles, tilts, evaluations) after constrained refinement (copies; inputs untouched)."""
    images = np.ascontiguousarray(images, dtype=np.float32)
    rows = np.array(rows, dtype=np.float64, order="C")
    particles = np.array(particles, dtype=np.float64, order="C")
    tilts = np.array(tilts, dtype=np.float64, order="C")
    nev = C.c_long(0)
    rc = lib().orc_csp_refine(ref.h, C.byref(cfg), C.byref(csp_cfg), _p(images), len(rows), _p(rows), _p(particles), len(particles),
                              _p(tilts), len(tilts), C.byref(nev))
    if rc:
        raise RuntimeError(f"oracle: csp_refine failed ({rc})")
    return rows, particles, tilts, nev.value


def sva_align(ref, cfg, volumes, wedges, poses):
    """Sub-tomogram alignment (orc_sva_align): returns (poses, scores, evaluations); inputs untouched."""
    volumes = np.ascontiguousarray(volumes, dtype=np.float32)
    wedges = np.ascontiguousarray(wedges, dtype=np.float32)
    poses = np.array(poses, dtype=np.float64, order="C")
    scores = np.zeros(len(poses), dtype=np.float64)
    nev = C.c_long(0)
    rc = lib().orc_sva_align(ref.h, C.byref(cfg), _p(volumes), len(poses), _p(wedges), _p(poses), _p(scores), C.byref(nev))
    if rc:
        raise RuntimeError(f"oracle: sva_align failed ({rc})")
    return poses, scores, nev.value


def sva_insert(acc, counts, cfg, volumes, wedges, poses, index=None):
    """Sub-tomogram average (orc_sva_insert): adds the aligned sub-volumes to `acc` (accum_floats(box) float32) and `counts` (2 int64) in place."""
    volumes = np.ascontiguousarray(volumes, dtype=np.float32)
    wedges = np.ascontiguousarray(wedges, dtype=np.float32)
    poses = np.ascontiguousarray(poses, dtype=np.float64)
    idx = None if index is None else np.ascontiguousarray(index, dtype=np.int64)
    rc = lib().orc_sva_insert(_p(acc), _p(counts), C.byref(cfg), _p(volumes), len(poses), _p(wedges), _p(poses), None if idx is None else _p(idx))
    if rc:
        raise RuntimeError(f"oracle: sva_insert failed ({rc})")
