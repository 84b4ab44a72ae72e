"""Data-driven dose weighting between refinement and reconstruction: per-exposure weights from the
mean score of the projections taken at that exposure, the side files `reconstruct3d` exchanges with the caller.

compute_global_weights restates src/pyp/inout/metadata/core.py:3039-3075: one float per exposure index (TIND, scanning
order) = mean SCORE of the rows with OCC > 0 at that index, -1.0 where there are none; pinned by the reference's own output
(tests/golden/golden_r03.json "global_weights", tests/test_golden_r03.py).  How the absent reconstruct3d turns
them into frequency weights is not visible in the reference; the rule used here is stated in include/ppm.h (ppm_recon_cfg):
exposure t is attenuated by q_t ^ (F min(1, (s / (transition s_Nyquist))^2)) with q_t = weight_t / max weight - build-defined,
parity unpinned.
"""
import os

import numpy as np

from .formats.cistem import COL


def compute_global_weights(rows):
    """Mean SCORE per TIND over the rows with OCCUPANCY > 0; -1.0 for exposure indices without rows."""
    rows = np.asarray(rows, dtype=np.float64)
    used = rows[rows[:, COL["OCCUPANCY"]] > 0.0]
    if len(used) == 0:
        return np.zeros(0)
    t = used[:, COL["TIND"]].astype(np.int64)
    n = int(t.max()) + 1
    cnt = np.bincount(t, minlength=n)
    tot = np.bincount(t, weights=used[:, COL["SCORE"]], minlength=n)
    return np.where(cnt > 0, tot / np.maximum(cnt, 1), -1.0)


def write_global_weights(path, weights):
    with open(path, "w") as f:
        f.write("\n".join(str(float(w)) for w in weights))


def read_global_weights(path):
    return np.loadtxt(path, ndmin=1)


def normalised(weights):
    """q_t in (0, 1]: weight over the best exposure's; 0 where the exposure has no data (not attenuated by the library)."""
    w = np.asarray(weights, dtype=np.float64)
    ok = w > 0
    if not ok.any():
        return np.zeros_like(w, dtype=np.float32)
    return np.where(ok, w / w[ok].max(), 0.0).astype(np.float32)


def weight_map(q, box, exponent, transition=1.0):
    """The weight of one exposure on the half plane [kx 0..box/2][ky wrapped 0..box-1] (what ppm_insert_batch applies)."""
    kx = np.arange(box // 2 + 1)[:, None].astype(np.float64)
    ky = np.fft.fftfreq(box, 1.0 / box)[None, :]
    k2 = kx * kx + ky * ky
    tr = transition if 0 < transition <= 1 else 1.0
    cap2 = (tr * box / 2) ** 2
    if not (0 < q < 1):
        return np.ones_like(k2)
    return np.exp(exponent * np.log(q) * np.minimum(k2, cap2) / cap2)


def write_weights_txt(path, q, box, exponent, transition=1.0):
    """`weights.txt` in the layout the caller's plotter reads (src/pyp/analysis/plot/pyp_frealign_plot_weights.py:15-35):
    per exposure, box values per column i = 1 .. box/2 with j = 0 .. box-1 outermost, then box/2 values for i = 0."""
    tmp = path + ".tmp%d" % os.getpid()
    with open(tmp, "w") as f:
        for qt in q:
            W = weight_map(float(qt), box, exponent, transition)         # [i][j]
            vals = [W[i, j] for j in range(box) for i in range(1, box // 2 + 1)] + [W[0, j] for j in range(box // 2)]
            f.write("\n".join("%.6f" % v for v in vals) + "\n")
    os.replace(tmp, path)


def write_scores_txt(path, weights):
    """`scores.txt`: one normalised mean score per exposure, -1 where there is none."""
    w = np.asarray(weights, dtype=np.float64)
    ok = w > 0
    out = np.where(ok, w / (w[ok].max() if ok.any() else 1.0), -1.0)
    tmp = path + ".tmp%d" % os.getpid()
    np.savetxt(tmp, out.reshape(-1, 1), fmt="%.6f")
    os.replace(tmp, path)
