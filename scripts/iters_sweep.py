"""Accuracy / speed of the compass iteration split (iters_hit, iters_final) at 256^2, SNR 0.05."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyp_amd import host, synth
from pyp_amd.abi import RefineCfg
N, M, px = 256, 4000, 1.0
vol = synth.phantom(N)
ref = host.Reference(vol, N / 2)
_, stack, rows = synth.make_dataset(N, M, pixel=px, snr=0.05, vol=vol, device="cuda", unique=M, batch=32)
start = synth.cistem.default_rows(M, px, 300.0, 2.7, 0.07)
for c in ("DEFOCUS_1", "DEFOCUS_2", "DEFOCUS_ANGLE"):
    start[:, synth.cistem.COL[c]] = rows[:, synth.cistem.COL[c]]
for tb, tc, k in ((3, 6, 20), (2, 7, 20), (2, 6, 20), (1, 8, 20), (3, 5, 20), (3, 6, 10), (2, 7, 10)):
    cfg = RefineCfg.make(box=N, pixel_size=px, mask_radius=0.32 * N * px, res_high=4.0, res_search=4.0, search_range_x=6.0, search_range_y=6.0,
                         res_signed_cc=30.0, iters_hit=tb, iters_final=tc, top_hits=k)
    ref.refine(cfg, stack[:500], start[:500])
    host.profile(True, True)
    t = time.perf_counter(); out = ref.refine(cfg, stack, start); dt = time.perf_counter() - t
    pr = host.profile_report()
    a, s = synth.angular_error_deg(out, rows), synth.shift_error_px(out, rows, px)
    print(f"Tb={tb} Tc={tc} K={k}: local {pr['local']['ms']/M*1e3:.2f} us/particle, total {M/dt:.0f}/s | angle median {np.median(a):.3f} 95% {np.percentile(a,95):.2f} within1 {np.mean(a<1):.3f} | shift med {np.median(s):.3f} | mean score {out[:,14].mean():.3f}")
