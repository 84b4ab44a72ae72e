"""Constrained refinement rate at BASELINE config 4's per-series shape (41 tilts x P particles), one MI355X, resident stack."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyp_amd import host, synth
from pyp_amd.abi import RefineCfg, CspCfg, CSP_PARTICLES, CSP_MICROGRAPHS

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
npart = int(sys.argv[2]) if len(sys.argv) > 2 else 300
px = 2.0
tl = np.linspace(-60, 60, 41)
vol, stack, rows, parts, tilts = synth.make_tilt_series(n, npart, tl, pixel=px, snr=0.1, device="cuda")
torch.cuda.synchronize()
rng = np.random.default_rng(3)
p2 = parts.copy()
for i in range(len(p2)):
    N = synth.euler_matrix(-p2[i, 4], -p2[i, 5], -p2[i, 6])
    for k in range(3):
        N = N @ synth.rot_xyz(k, rng.normal(0, 2.0))
    p2[i, 4:7] = -synth.angles_from_matrix(N)
    p2[i, 1:4] += rng.normal(0, 1.0, 3)
rows2 = synth.csp_rows_from_params(rows, parts, tilts, p2, tilts)
cfg = RefineCfg.make(box=n, pixel_size=px, mask_radius=0.32 * n * px, res_high=px * n / (0.25 * n), res_signed_cc=30.0, global_search=0)
ref = host.Reference(vol, n / 2)
cc = CspCfg.make(CSP_PARTICLES, tol_angle=(8, 8, 8), tol_shift=4.0)
ref.csp_refine(cfg, cc, stack[:41 * 4], rows2[:41 * 4], p2, tilts)
host.profile(True, True)
t0 = time.time()
r3, p3, t3 = ref.csp_refine(cfg, cc, stack, rows2, p2, tilts)
dt = time.time() - t0
prof = host.profile_report()
def perr(a, b):
    return np.array([np.degrees(np.arccos(np.clip((np.trace(synth.euler_matrix(-x[4], -x[5], -x[6]).T @ synth.euler_matrix(-y[4], -y[5], -y[6])) - 1) / 2, -1, 1))) for x, y in zip(a, b)])
print("box %d, %d particles x 41 tilts = %d projections: particle mode %.2f s = %.0f particles/s (%.0f projections/s); prep %.1f ms, sweeps %.1f ms"
      % (n, npart, len(rows), dt, npart / dt, len(rows) / dt, prof["prep"]["ms"], prof["local"]["ms"]))
print("particle angle error %.2f -> %.2f deg (median), shift %.2f -> %.2f px" % (np.median(perr(p2, parts)), np.median(perr(p3, parts)),
      np.median(np.linalg.norm(p2[:, 1:4] - parts[:, 1:4], axis=1)), np.median(np.linalg.norm(p3[:, 1:4] - parts[:, 1:4], axis=1))))
cm = CspCfg.make(CSP_MICROGRAPHS, tol_angle=(1.5, 1.0, 0), tol_shift=4.0)
t0 = time.time()
ref.csp_refine(cfg, cm, stack, r3, p3, t3)
print("tilt mode %.2f s for 41 tilts" % (time.time() - t0))
