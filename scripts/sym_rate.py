"""Fourier insertion rate with point-group symmetry (every particle is inserted once per operator), 256^2, resident stack."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyp_amd import host, synth
from pyp_amd.abi import ReconCfg
N, px, M = 256, 1.0, int(sys.argv[1]) if len(sys.argv) > 1 else 20000
vol = synth.phantom(N)
_, stack, rows = synth.make_dataset(N, M, pixel=px, snr=0.05, vol=vol, device="cuda", unique=1024, batch=32)
torch.cuda.synchronize()
rc = ReconCfg(box=N, pixel_size=px, res_limit=2 * px, normalize=1, split_by_pind=0, mask_radius=0.32 * N * px)
for sym in ("C1", "C4", "D7", "O", "I"):
    acc = host.Accumulator(N, px, sym)
    acc.insert(rc, stack[:2000], rows[:2000])
    host.lib.load().ppm_device_sync()
    t = time.perf_counter(); acc.insert(rc, stack, rows); host.lib.load().ppm_device_sync(); dt = time.perf_counter() - t
    acc.close()
    nsym = {"C1": 1, "C4": 4, "D7": 14, "O": 24, "I": 60}[sym]
    print(f"{sym:3s} ({nsym:2d} operators): {M / dt:10.0f} particles/s = {M * nsym / dt / 1e6:.2f} M slice insertions/s", flush=True)
