"""Why is the first read of a freshly written /dev/shm stack slow?  Writes a file (a) through a memory map, (b) with write(), then
times two pread passes over it from child processes (8 threads each)."""
import os, sys, time, subprocess, numpy as np
GB = int(sys.argv[1]) if len(sys.argv) > 1 else 8
child = r'''
import os, sys, time, numpy as np
from concurrent.futures import ThreadPoolExecutor
fn = sys.argv[1]; n = os.path.getsize(fn); fd = os.open(fn, os.O_RDONLY)
buf = [np.empty(64 << 20, np.uint8) for _ in range(8)]
def work(t):
    per = n // 8; pos = t * per; end = pos + per; mv = memoryview(buf[t])
    while pos < end:
        got = os.preadv(fd, [mv[:min(len(mv), end - pos)]], pos); pos += got
for k in range(2):
    t0 = time.time()
    with ThreadPoolExecutor(8) as ex: list(ex.map(work, range(8)))
    print("   pass %d: %.1f GB/s" % (k, n / (time.time() - t0) / 1e9), flush=True)
'''
for how in ("mmap", "write"):
    fn = "/dev/shm/ppm_probe_%s.bin" % how
    t0 = time.time()
    if how == "mmap":
        with open(fn, "wb") as f: f.truncate(GB << 30)
        mm = np.memmap(fn, dtype=np.uint8, mode="r+")
        for lo in range(0, GB << 30, 1 << 30): mm[lo:lo + (1 << 30)] = 7
        mm.flush(); del mm
    else:
        blk = np.full(1 << 30, 7, np.uint8)
        with open(fn, "wb") as f:
            for _ in range(GB): f.write(blk)
    print(how, "written in %.1f s" % (time.time() - t0), flush=True)
    for run in range(2):
        print("  child", run, flush=True)
        subprocess.run([sys.executable, "-c", child, fn])
    os.remove(fn)
