"""A/B of the drop-in executables' pipeline settings on ONE stack file: python scripts/dropin_ab.py [particles]
Prints wall time and the Pipeline line of refine3d / reconstruct3d for several environments."""
import os, subprocess, sys, tempfile, time, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyp_amd import synth
from pyp_amd.formats import cistem, mrc

M = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
N, PX = 256, 1.0
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = tempfile.mkdtemp(prefix="ppm_ab_", dir="/dev/shm")
try:
    vol, stack, rows = synth.make_dataset(N, M, pixel=PX, snr=0.05, device="cuda", unique=min(M, 2048), batch=32)
    mm = mrc.create(os.path.join(d, "p_stack.mrc"), (M, N, N), pixel_size=PX)
    for lo in range(0, M, 4096):
        mm[lo:lo + 4096] = stack[lo:lo + 4096].cpu().numpy()
    mm.flush(); del mm, stack
    torch.cuda.empty_cache()
    mrc.write(vol, os.path.join(d, "p_r01.mrc"), pixel_size=PX)
    start = cistem.default_rows(M, PX, 300.0, 2.7, 0.07)
    for c in ("DEFOCUS_1", "DEFOCUS_2", "DEFOCUS_ANGLE"):
        start[:, cistem.COL[c]] = rows[:, cistem.COL[c]]
    cistem.write_parameters(os.path.join(d, "p_r01.cistem"), start)
    rng = "%07d_%07d" % (1, M)
    refine = ["p_stack.mrc", "p_r01.cistem", "null", "p_r01.mrc", "statistics_r01.txt", "no", "no", f"p_r01_match.mrc_{rng}", f"p_r01_{rng}.cistem",
              f"p_r01_{rng}_changes.cistem", "C1", 1, M, 1, PX, 300, 0, 0.32 * N * PX, 0, PX * N / 64, 30.0, 8.0, 0.48 * N * PX, PX * N / 64, 15.0, 20, 6.0, 6.0,
              0, 0, 0, 0, 500, 50.0, 1, "yes", "no", "yes", "yes", "yes", "yes", "yes", "no", "no", "no", "yes", "no", "no", "no", "no"]
    recon = ["p_stack.mrc", f"p_r01_{rng}.cistem", "null", "p_r01.mrc", "p_map1.mrc", "p_map2.mrc", "output.mrc", "p_n1.res", "C1", 1, M, PX, 300, 0, 0.45 * N * PX,
             2 * PX, 0, 0, "no", 0, -1, "no", 0, 1, 1, "yes", "no", "no", "no", "no", "yes", "no", "no", "no", "no", "yes", "dump1.bin", "dump2.bin", 1]
    import json
    envs = json.loads(os.environ.get("PPM_AB_ENVS", '[{}, {"PPM_IO_CHUNK": "1024"}, {"PPM_IO_THREADS": "12"}, {"PPM_SYNC": "spin"}]'))
    for e in envs:
        for prog, script in (("refine3d", refine), ("reconstruct3d", recon)):
            t0 = time.time()
            rc = subprocess.run(f"{ROOT}/bin/{prog} << eot > {prog}.log 2>&1\n" + "\n".join(str(x) for x in script) + "\neot\n", shell=True, cwd=d,
                                env=dict(os.environ, **e)).returncode
            dt = time.time() - t0
            log = open(os.path.join(d, prog + ".log")).read()
            lines = [ln for ln in log.splitlines() if ln.startswith(("Timing:", "Pipeline:"))]
            print(e, prog, "rc", rc, "%.2f s = %.0f particles/s" % (dt, M / dt), flush=True)
            for ln in lines:
                print("    ", ln, flush=True)
            if rc:
                print(log[-800:])
finally:
    shutil.rmtree(d, ignore_errors=True)
