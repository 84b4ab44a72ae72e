"""PCIe-inclusive rate: the same refinement with the particle stack handed over as a HOST buffer (numpy)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyp_amd import host, synth
from pyp_amd.abi import RefineCfg
N, M, px = 256, 16000, 1.0
vol = synth.phantom(N)
_, stack, rows = synth.make_dataset(N, M, pixel=px, snr=0.05, vol=vol, device="cuda", unique=256, batch=32)
cfg = RefineCfg.make(box=N, pixel_size=px, mask_radius=0.32 * N * px, res_high=4.0, res_search=4.0, search_range_x=6.0, search_range_y=6.0, res_signed_cc=30.0)
ref = host.Reference(vol, N / 2)
start = synth.cistem.default_rows(M, px, 300.0, 2.7, 0.07)
for c in ("DEFOCUS_1", "DEFOCUS_2", "DEFOCUS_ANGLE"):
    start[:, synth.cistem.COL[c]] = rows[:, synth.cistem.COL[c]]
h = stack.cpu().numpy()
for name, imgs in (("device-resident", stack), ("host buffer", h)):
    ref.refine(cfg, imgs[:2000], start[:2000])
    t = time.perf_counter(); ref.refine(cfg, imgs, start); dt = time.perf_counter() - t
    print(f"{name}: {M / dt:.0f} particles/s")
