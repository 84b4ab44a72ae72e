"""Pose recovery against the synthetic ground truth at the BASELINE size (256^2, band 64 px), from scratch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyp_amd import host, synth
from pyp_amd.abi import RefineCfg
N, M, px = 256, 2000, 1.0
vol = synth.phantom(N)
ref = host.Reference(vol, N / 2)
for snr in (0.0, 0.2, 0.05, 0.02):
    _, stack, rows = synth.make_dataset(N, M, pixel=px, snr=snr, vol=vol, device="cuda", unique=M, batch=32)
    cfg = RefineCfg.make(box=N, pixel_size=px, mask_radius=0.32 * N * px, res_high=4.0, res_search=4.0, search_range_x=6.0, search_range_y=6.0, res_signed_cc=30.0)
    start = synth.cistem.default_rows(M, px, 300.0, 2.7, 0.07)
    for c in ("DEFOCUS_1", "DEFOCUS_2", "DEFOCUS_ANGLE"):
        start[:, synth.cistem.COL[c]] = rows[:, synth.cistem.COL[c]]
    out = ref.refine(cfg, stack, start)
    a, s = synth.angular_error_deg(out, rows), synth.shift_error_px(out, rows, px)
    print(f"SNR {snr}: angle median {np.median(a):.3f} deg, 95% {np.percentile(a,95):.2f}, within 1 deg {np.mean(a<1):.3f}; shift median {np.median(s):.3f} px, 95% {np.percentile(s,95):.2f}")
