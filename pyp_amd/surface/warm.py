"""Device start-up in the background for the drop-in executables.  Creating the GPU context, loading the code object and
page-locking the staging buffers take 0.4 - 0.5 s of a process that lives for one or two seconds; started from bin/* BEFORE numpy
is imported, this thread overlaps them with interpreter start-up, the parsing of the answers and the reading of the parameter
files.  Nothing but ctypes is imported here.  A failure is not reported from this thread: the main thread's own ppm_init call
(pyp_amd.lib.init) meets the same condition and ends with the ERROR line."""
import ctypes
import os
import threading

PIN_BYTES = 256 << 20          # one staging buffer of cli._iter_image_chunks (set by start())
_lock_dev = None               # device whose advisory lock this thread takes BEFORE it creates the context (cli.gpu_lock adopts it)
_lock_fd = None
_lock_done = threading.Event()
_thread = None
_lock = threading.Lock()
_pool = []                     # page-locked buffers nobody has taken yet
_want = 3
_lib = None


def chunk_mb(default):
    try:
        return max(1, int(os.environ.get("PPM_IO_CHUNK_MB", default)))
    except ValueError:
        return int(default)


def start(pinned=3, mb=256, lock=True):
    """pinned: how many staging buffers of `mb` MB (PPM_IO_CHUNK_MB overrides) to page-lock ahead: refine3d / reconstruct3d stream
    the stack through three."""
    global _thread, _want, PIN_BYTES
    PIN_BYTES = chunk_mb(mb) << 20
    os.environ.setdefault("PPM_SYNC", "block")      # a spinning device wait starves the reader threads (measured: 29 k -> 53 k particles/s)
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "libpypmatch.so")
    if _thread is not None or not os.path.exists(so):
        return
    try:
        dev = int(os.environ.get("PPM_DEVICE", "0"))
    except ValueError:
        return
    _want = pinned

    global _lock_dev
    _lock_dev = dev if lock else None

    def run():
        global _lib, _lock_fd
        try:
            # the per-GPU lock first (PYP starts several processes per node, src/pyp/system/mpi.py:104): a process that waits for
            # the device holds neither a GPU context nor page-locked memory meanwhile - like the compiled executables
            try:
                if not lock:
                    raise OSError("no lock asked for")
                import fcntl
                old = os.umask(0)
                try:
                    fd = os.open(os.path.join(os.environ.get("PPM_LOCK_DIR", "/tmp"), "pyp_amd_gpu%d.lock" % dev), os.O_CREAT | os.O_RDWR, 0o666)
                finally:
                    os.umask(old)
                fcntl.flock(fd, fcntl.LOCK_EX)
                _lock_fd = fd
            except OSError:
                pass                                 # cli.gpu_lock reports what is wrong with the lock file
            finally:
                _lock_done.set()
            L = ctypes.CDLL(so)                      # the same dlopen handle pyp_amd.lib.load() gets later; ppm_init is idempotent
            if L.ppm_init(dev) != 0:
                return
            L.ppm_host_alloc.restype = ctypes.c_void_p
            L.ppm_host_alloc.argtypes = [ctypes.c_size_t]
            L.ppm_host_free.argtypes = [ctypes.c_void_p]
            _lib = L
            while True:
                with _lock:
                    if len(_pool) >= _want:
                        return
                p = L.ppm_host_alloc(PIN_BYTES)      # ~0.07 s each, the GIL released
                if not p:
                    return
                with _lock:
                    _pool.append(p)
        except Exception:
            pass
    _thread = threading.Thread(target=run)          # not a daemon: the interpreter waits for it rather than exit in the middle of hipInit
    _thread.start()


def adopt_lock(device):
    """The lock this thread took for `device` (a file descriptor the caller now owns and must unlock + close), or None."""
    global _lock_fd
    if _thread is None or _lock_dev != int(device):
        return None
    _lock_done.wait()
    fd, _lock_fd = _lock_fd, None
    return fd


def join_init():
    """Wait until the device is up (the buffers may still be on their way)."""
    while _thread is not None and _thread.is_alive() and _lib is None:
        _thread.join(0.002)


def limit(n):
    """No more than n buffers are needed (short ranges): the thread stops page-locking; release() frees any surplus."""
    global _want
    with _lock:
        _want = min(_want, int(n))


def take(nbytes):
    """A page-locked buffer of PIN_BYTES if one is ready and large enough, else None (the caller allocates its own)."""
    if nbytes > PIN_BYTES:
        return None
    with _lock:
        return _pool.pop() if _pool else None


def release():
    """Stop and free what was not taken (called when the executable is done or dies)."""
    global _want
    with _lock:
        _want = 0
    if _thread is not None:
        _thread.join()
    with _lock:
        while _pool:
            _lib.ppm_host_free(_pool.pop())
    global _lock_fd
    if _lock_fd is not None:            # nobody adopted the lock (the run ended before it reached the device)
        try:
            os.close(_lock_fd)
        except OSError:
            pass
        _lock_fd = None
