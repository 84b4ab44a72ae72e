"""Sub-tomogram alignment rate at BASELINE config 5's geometry (192^3 sub-volumes), resident volumes, one MI355X."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyp_amd import host, synth
from pyp_amd.abi import SvaCfg

n, nv = int(sys.argv[1]) if len(sys.argv) > 1 else 192, int(sys.argv[2]) if len(sys.argv) > 2 else 64
vol, vols, poses, wedges = synth.make_subtomograms(n, nv, snr=0.1, device="cuda")
torch.cuda.synchronize()
cfg = SvaCfg.make(n, window=(0.33 * n, 0.33 * n, 0.33 * n), window_sigma=4.0, highpass=(0.05, 0.01), lowpass=(0.125, 0.05), tol_angle=10.0, tol_shift=10.0)
start = synth.perturb_poses(poses, 3.0, 2.0)
ref = host.Reference(vol, n / 2)
ref.sva_align(cfg, vols[:4], wedges[:4], start[:4])
host.profile(True, True)
t0 = time.time()
mode = sys.argv[3] if len(sys.argv) > 3 else "device"      # "host": pageable host memory; "pinned": page-locked (what sva.align_table reads into)
src = vols
if mode == "host":
    src = vols.cpu().numpy()
elif mode == "pinned":
    pb = host.PinnedBuffer(vols.numel())
    src = pb.array.reshape(tuple(vols.shape))
    src[...] = vols.cpu().numpy()
if src is not vols:
    del vols
    torch.cuda.empty_cache()
t0 = time.time()
out, sc = ref.sva_align(cfg, src, wedges, start)
dt = time.time() - t0
prof = host.profile_report()
print("box %d: %d sub-volumes in %.2f s = %.1f sub-volumes/s; prep %.1f ms, search %.1f ms per sub-volume" % (n, nv, dt, nv / dt, prof["prep"]["ms"] / nv, prof["local"]["ms"] / nv))
print("angle error before %.2f after %.2f deg (median); shift error before %.2f after %.2f px; mean score %.3f" % (
    np.median(synth.pose_angle_error(start, poses)), np.median(synth.pose_angle_error(out, poses)),
    np.median(np.linalg.norm(start[:, 9:] - poses[:, 9:], axis=1)), np.median(np.linalg.norm(out[:, 9:] - poses[:, 9:], axis=1)), sc.mean()))
