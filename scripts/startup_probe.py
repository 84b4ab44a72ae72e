"""Where a process's device start-up goes: python scripts/startup_probe.py  (ctypes only; one line per step, seconds)."""
import ctypes, os, time
t0 = time.time()
L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pyp_amd", "libpypmatch.so"))
t1 = time.time()
rc = L.ppm_init(0)
t2 = time.time()
L.ppm_accum_create.restype = ctypes.c_void_p
L.ppm_accum_create.argtypes = [ctypes.c_int, ctypes.c_float, ctypes.c_char_p, ctypes.c_void_p]
a = L.ppm_accum_create(256, 1.0, b"C1", None)
t3 = time.time()
L.ppm_device_alloc.restype = ctypes.c_void_p
L.ppm_device_alloc.argtypes = [ctypes.c_size_t]
p = L.ppm_device_alloc(2 << 30)
t4 = time.time()
L.ppm_host_alloc.restype = ctypes.c_void_p
L.ppm_host_alloc.argtypes = [ctypes.c_size_t]
h = L.ppm_host_alloc(256 << 20)
t5 = time.time()
print("dlopen %.3f  ppm_init %.3f (rc %d)  accum_create(256) %.3f  device_alloc(2 GB) %.3f  host_alloc(256 MB) %.3f" % (t1 - t0, t2 - t1, rc, t3 - t2, t4 - t3, t5 - t4))
