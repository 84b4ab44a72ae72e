"""Throughput of the extraction kernel: 100k boxes of 256^2 from a resident 8k x 8k micrograph into a resident stack."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyp_amd import host
rows = cols = 8192; box = 256; M = 50000
mic = torch.randn((rows, cols), device="cuda", dtype=torch.float32)
rng = np.random.default_rng(1)
coords = np.stack([rng.uniform(0, cols, M), rng.uniform(0, rows, M)], axis=1)
out = torch.empty((M, box, box), device="cuda", dtype=torch.float32)
host.extract_boxes(mic, coords[:1000], box, 82.0, 1.0, out=out[:1000])
host.profile(True, True)
torch.cuda.synchronize(); t = time.perf_counter()
host.extract_boxes(mic, coords, box, 82.0, 1.0, out=out)
dt = time.perf_counter() - t
ms = host.profile_report()["extract"]["ms"]
byt = M * box * box * 4 * 2.0
print(f"extract: {M/dt:.0f} boxes/s wall, kernel {ms:.1f} ms -> {byt/ms/1e6:.0f} GB/s algorithmic (read + write 4 N^2 each)")
