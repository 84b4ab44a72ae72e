"""Host-side objects over the C ABI: Reference (prepared 3-D reference), refine(), Accumulator
(half-map accumulators + finalise).  Mirrors what the refine3d / reconstruct3d / merge3d processes
do between reading their inputs and writing their outputs."""
import ctypes as C

import numpy as np

from . import lib
from .abi import NCOL, STATS_COLS, K_NAMES, FinalCfg, ReconCfg, RefineCfg  # noqa: F401


def _sync_producer(t):
    """The library works on its own HIP stream (include/ppm.h, "Stream ordering"): whatever torch has queued on the
    tensor's device (the copy or kernel that produced or zeroed it) must have finished before the raw pointer goes in."""
    import torch
    torch.cuda.current_stream(t.device).synchronize()


def _images_arg(images, n_img, box):
    """numpy array (host) or an object with data_ptr() on the GPU (torch tensor) -> (pointer, on_device, keepalive)."""
    if hasattr(images, "data_ptr") and hasattr(images, "is_cuda"):
        if not images.is_cuda:
            images = images.numpy()
        else:
            if not getattr(images, "ready", False):      # a torch tensor: wait for whatever torch queued on it (cli.DeviceImages arrives finished)
                _sync_producer(images)
            if str(images.dtype) != "torch.float32" or not images.is_contiguous():
                raise ValueError("ERROR: device image stack must be contiguous float32")
            if images.numel() != n_img * box * box:
                raise ValueError("ERROR: image stack size does not match the rows")
            return C.c_void_p(images.data_ptr()), 1, images
    a = np.ascontiguousarray(images, dtype=np.float32)
    if a.size != n_img * box * box:
        raise ValueError("ERROR: image stack size does not match the rows")
    return lib.ptr(a), 0, a


def _volumes_arg(volumes, n_vol, box):
    """numpy array (host) or a device array (torch tensor / anything with data_ptr(), is_cuda) -> (pointer, on_device, keepalive)."""
    if hasattr(volumes, "is_cuda") and volumes.is_cuda:
        if str(volumes.dtype) != "torch.float32" or not volumes.is_contiguous() or volumes.numel() != n_vol * box ** 3:
            raise ValueError("ERROR: device volumes must be contiguous float32 of V * box^3 elements")
        if not getattr(volumes, "ready", False):
            _sync_producer(volumes)
        return C.c_void_p(volumes.data_ptr()), 1, volumes
    a = np.ascontiguousarray(volumes.numpy() if hasattr(volumes, "numpy") else volumes, dtype=np.float32)
    if a.size != n_vol * box ** 3:
        raise ValueError("ERROR: volumes do not match the poses")
    return lib.ptr(a), 0, a


class Reference:
    """3-D reference prepared for projection matching up to `max_band_px` Fourier pixels."""

    def __init__(self, vol, max_band_px=None, device=0, pad=1, ring_weight=None):
        lib.init(device)
        vol = np.ascontiguousarray(vol, dtype=np.float32)
        if vol.ndim != 3 or len(set(vol.shape)) != 1:
            raise ValueError("ERROR: reference must be a cubic volume")
        self.n = vol.shape[0]
        band = self.n / 2 if max_band_px is None else float(max_band_px)
        w = None if ring_weight is None else np.ascontiguousarray(ring_weight, dtype=np.float32)
        self.h = lib.load().ppm_reference_create_weighted(lib.ptr(vol), self.n, band, int(pad), None if w is None else lib.ptr(w),
                                                          0 if w is None else int(w.size))
        if not self.h:
            raise lib.PpmError(lib.last_error())

    def refine(self, cfg, images, rows):
        rows = np.ascontiguousarray(rows, dtype=np.float64)
        if rows.ndim != 2 or rows.shape[1] != NCOL:
            raise ValueError("ERROR: rows must be (M, 32)")
        out = np.empty_like(rows)
        p, on_dev, keep = _images_arg(images, len(rows), cfg.box)
        lib.check(lib.load().ppm_refine_batch(self.h, C.byref(cfg), p, on_dev, len(rows), lib.ptr(rows), lib.ptr(out)))
        del keep
        return out

    def match_projections(self, cfg, rows):
        """Matching projections of refine3d (ppm_match_projections): float32 [M, box, box], the noise-free model of each row's particle."""
        rows = np.ascontiguousarray(rows, dtype=np.float64)
        if rows.ndim != 2 or rows.shape[1] != NCOL:
            raise ValueError("ERROR: rows must be (M, 32)")
        out = np.empty((len(rows), cfg.box, cfg.box), dtype=np.float32)
        lib.check(lib.load().ppm_match_projections(self.h, C.byref(cfg), lib.ptr(rows), len(rows), lib.ptr(out)))
        return out

    def csp_refine(self, cfg, csp_cfg, images, rows, particles, tilts):
        """Constrained refinement (ppm_csp_refine): returns updated copies (rows, particles, tilts)."""
        rows = np.array(rows, dtype=np.float64, order="C")
        particles = np.array(particles, dtype=np.float64, order="C")
        tilts = np.array(tilts, dtype=np.float64, order="C")
        if rows.ndim != 2 or rows.shape[1] != NCOL or particles.ndim != 2 or particles.shape[1] != 12 or tilts.ndim != 2 or tilts.shape[1] != 6:
            raise ValueError("ERROR: rows must be (M, 32), particles (P, 12), tilts (T, 6)")
        p, on_dev, keep = _images_arg(images, len(rows), cfg.box)
        lib.check(lib.load().ppm_csp_refine(self.h, C.byref(cfg), C.byref(csp_cfg), p, on_dev, len(rows), lib.ptr(rows), lib.ptr(particles),
                                            len(particles), lib.ptr(tilts), len(tilts)))
        del keep
        return rows, particles, tilts

    def sva_align(self, cfg, volumes, wedges, poses, accumulator=None, index=None):
        """Sub-tomogram alignment (ppm_sva_align): volumes (V, N, N, N) float32 (numpy or CUDA tensor), wedges (V, 2) tilt limits in
        degrees, poses (V, 12) = N row-major + shift.  Returns (refined poses, scores).  accumulator: an Accumulator of the same box -
        every chunk is added to the sub-tomogram average at its refined poses while it is in device memory (ppm_sva_align_average;
        half-map = parity of index, default 0 .. V-1)."""
        poses = np.array(poses, dtype=np.float64, order="C")
        if poses.ndim != 2 or poses.shape[1] != 12:
            raise ValueError("ERROR: poses must be (V, 12)")
        w = np.ascontiguousarray(wedges, dtype=np.float32).reshape(len(poses), 2)
        scores = np.zeros(len(poses), dtype=np.float64)
        p, on_dev, keep = _volumes_arg(volumes, len(poses), cfg.box)
        if accumulator is None:
            lib.check(lib.load().ppm_sva_align(self.h, C.byref(cfg), p, on_dev, len(poses), lib.ptr(w), lib.ptr(poses), lib.ptr(scores)))
        else:
            idx = None if index is None else np.ascontiguousarray(index, dtype=np.int64)
            if idx is not None and idx.shape != (len(poses),):
                raise ValueError("ERROR: index must be (V,)")
            if accumulator._ext is not None:
                _sync_producer(accumulator._ext)
            lib.check(lib.load().ppm_sva_align_average(self.h, accumulator.h, C.byref(cfg), p, on_dev, len(poses), lib.ptr(w), lib.ptr(poses), lib.ptr(scores),
                                                       None if idx is None else lib.ptr(idx)))
        del keep
        return poses, scores

    def note(self):
        """Remarks of the last refine() the caller should log ("" if none)."""
        return (lib.load().ppm_refine_note(self.h) or b"").decode(errors="replace")

    def last_counts(self):
        v = [C.c_long() for _ in range(4)]
        lib.check(lib.load().ppm_refine_last_counts(self.h, *[C.byref(x) for x in v]))
        return dict(zip(("n_global", "n_local", "samples_global", "samples_local"), [x.value for x in v]))

    def close(self):
        if getattr(self, "h", None):
            lib.load().ppm_reference_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Accumulator:
    """The two half-map accumulators of a reconstruction ({re, im, weight} per Fourier voxel)."""

    def __init__(self, box, pixel_size, symmetry="C1", device=0, ext_tensor=None):
        lib.init(device)
        self.box, self.pixel = int(box), float(pixel_size)
        self.nfloats = int(lib.load().ppm_accum_floats(self.box))
        self._ext = ext_tensor
        ext = None
        if ext_tensor is not None:
            if ext_tensor.numel() != self.nfloats or not ext_tensor.is_cuda or not ext_tensor.is_contiguous():
                raise ValueError("ERROR: external accumulator tensor has the wrong size or is not on the GPU")
            ext = C.c_void_p(ext_tensor.data_ptr())
        self.h = lib.load().ppm_accum_create(self.box, self.pixel, symmetry.encode(), ext)
        if not self.h:
            raise lib.PpmError(lib.last_error())

    def insert(self, cfg, images, rows):
        rows = np.ascontiguousarray(rows, dtype=np.float64)
        p, on_dev, keep = _images_arg(images, len(rows), self.box)
        if self._ext is not None:
            _sync_producer(self._ext)
        lib.check(lib.load().ppm_insert_batch(self.h, C.byref(cfg), p, on_dev, len(rows), lib.ptr(rows)))
        del keep

    def sva_insert(self, cfg, volumes, wedges, poses, index=None):
        """Sub-tomogram average (ppm_sva_insert): add aligned sub-volumes (V, N, N, N) float32 (numpy or CUDA tensor) with their wedges
        (V, 2) and poses (V, 12) to the half-map accumulators; index (V,) decides the half (parity), default 0 .. V-1."""
        poses = np.ascontiguousarray(poses, dtype=np.float64)
        if poses.ndim != 2 or poses.shape[1] != 12:
            raise ValueError("ERROR: poses must be (V, 12)")
        w = np.ascontiguousarray(wedges, dtype=np.float32).reshape(len(poses), 2)
        idx = None if index is None else np.ascontiguousarray(index, dtype=np.int64)
        if idx is not None and idx.shape != (len(poses),):
            raise ValueError("ERROR: index must be (V,)")
        p, on_dev, keep = _volumes_arg(volumes, len(poses), self.box)
        if self._ext is not None:
            _sync_producer(self._ext)
        lib.check(lib.load().ppm_sva_insert(self.h, C.byref(cfg), p, on_dev, len(poses), lib.ptr(w), lib.ptr(poses), None if idx is None else lib.ptr(idx)))
        del keep

    def counts(self):
        return [int(lib.load().ppm_accum_count(self.h, 0)), int(lib.load().ppm_accum_count(self.h, 1))]

    def set_counts(self, c0, c1):
        lib.load().ppm_accum_set_count(self.h, 0, int(c0))
        lib.load().ppm_accum_set_count(self.h, 1, int(c1))

    def reduce(self, comm, root=-1):
        """Sum this rank's accumulator and particle counters over the communicator (ppm_accum_reduce: RCCL on the library's
        stream; root < 0 = all-reduce).  `comm` comes from `make_comm` (or is the caller's own ncclComm_t pointer)."""
        if self._ext is not None:
            _sync_producer(self._ext)
        lib.check(lib.load().ppm_accum_reduce(self.h, C.c_void_p(comm), int(root)))

    def download(self):
        a = np.empty(self.nfloats, dtype=np.float32)
        if self._ext is not None:
            _sync_producer(self._ext)
        lib.check(lib.load().ppm_accum_download(self.h, lib.ptr(a)))
        return a

    def add(self, host):
        a = np.ascontiguousarray(host, dtype=np.float32)
        if a.size != self.nfloats:
            raise ValueError("ERROR: dump has the wrong size for this box")
        if self._ext is not None:
            _sync_producer(self._ext)
        lib.check(lib.load().ppm_accum_add(self.h, lib.ptr(a)))

    def finalize(self, fcfg):
        n = self.box
        h1 = np.empty((n, n, n), dtype=np.float32)
        h2 = np.empty_like(h1)
        fl = np.empty_like(h1)
        stats = np.zeros((n // 2 - 1, STATS_COLS), dtype=np.float64)
        if self._ext is not None:
            _sync_producer(self._ext)
        lib.check(lib.load().ppm_finalize(self.h, C.byref(fcfg), lib.ptr(h1), lib.ptr(h2), lib.ptr(fl), lib.ptr(stats)))
        return h1, h2, fl, stats

    def close(self):
        if getattr(self, "h", None):
            lib.load().ppm_accum_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def comm_unique_id():
    """128 opaque bytes (ncclUniqueId) made by ONE rank; hand them to the others by any means."""
    buf = (C.c_char * 128)()
    lib.check(lib.load().ppm_comm_unique_id(buf))
    return bytes(buf)


def make_comm(n_ranks, rank, unique_id, device=0):
    """An RCCL communicator over the ranks' GPUs (ppm_comm_create; collective: returns when all ranks have joined)."""
    lib.init(device)
    if len(unique_id) != 128:
        raise ValueError("ERROR: the communicator id must be 128 bytes")
    buf = (C.c_char * 128).from_buffer_copy(unique_id)
    h = lib.load().ppm_comm_create(int(n_ranks), int(rank), buf)
    if not h:
        raise lib.PpmError(lib.last_error())
    return h


def comm_count(comm):
    """Ranks of the communicator as RCCL itself reports them (ppm_comm_count = ncclCommCount)."""
    n = lib.load().ppm_comm_count(comm)
    if n < 0:
        raise lib.PpmError(lib.last_error())
    return int(n)


def destroy_comm(comm):
    if comm:
        lib.load().ppm_comm_destroy(C.c_void_p(comm))


def extract_boxes(micrograph, coords, box, radius_A, pixel_size, coordinate_binning=1, normalize=True, fix_empty=True,
                  out=None, device=0):
    """Crop + normalise particle boxes (src/pyp/extract/core.py:447-506 with src/pyp/analysis/image.py:406-417).
    micrograph: 2-D float32 numpy array or CUDA tensor; coords: (M, 2) of (box[0], box[1]) = (column, row) coordinates;
    out: optional CUDA tensor (M, box, box) to fill in place (resident stack).  Returns the stack."""
    lib.init(device)
    coords = np.ascontiguousarray(coords, dtype=np.float64).reshape(-1, 2)
    m = len(coords)
    radius_px = float(radius_A) / (float(pixel_size) * float(coordinate_binning))
    if hasattr(micrograph, "is_cuda") and micrograph.is_cuda:
        if str(micrograph.dtype) != "torch.float32" or not micrograph.is_contiguous() or micrograph.dim() != 2:
            raise ValueError("ERROR: device micrograph must be a contiguous 2-D float32 tensor")
        _sync_producer(micrograph)
        ip, idev, rows, cols, keep = C.c_void_p(micrograph.data_ptr()), 1, micrograph.shape[0], micrograph.shape[1], micrograph
    else:
        a = np.ascontiguousarray(micrograph, dtype=np.float32)
        if a.ndim != 2:
            raise ValueError("ERROR: micrograph must be 2-D")
        ip, idev, rows, cols, keep = lib.ptr(a), 0, a.shape[0], a.shape[1], a
    if out is not None:
        if not out.is_cuda or out.numel() != m * box * box or not out.is_contiguous():
            raise ValueError("ERROR: output stack tensor has the wrong size or is not on the GPU")
        _sync_producer(out)
        op, odev, res = C.c_void_p(out.data_ptr()), 1, out
    else:
        res = np.empty((m, box, box), dtype=np.float32)
        op, odev = lib.ptr(res), 0
    lib.check(lib.load().ppm_extract_boxes(ip, idev, int(rows), int(cols), lib.ptr(coords), m, int(box), float(coordinate_binning),
                                           radius_px, int(bool(normalize)), int(bool(fix_empty)), op, odev))
    del keep
    return res


class PinnedBuffer:
    """Page-locked host memory as a float32 numpy array (ppm_host_alloc): staging for uploads that overlap compute."""

    def __init__(self, n_floats, device=0, ptr=None):
        """ptr: memory already page-locked by ppm_host_alloc (surface/warm.py hands over what it prepared); owned from here on."""
        lib.init(device)
        self.n = int(n_floats)
        self.ptr = ptr or lib.load().ppm_host_alloc(self.n * 4)
        if not self.ptr:
            raise lib.PpmError(lib.last_error())
        self.array = np.ctypeslib.as_array((C.c_float * self.n).from_address(self.ptr))

    def close(self):
        if getattr(self, "ptr", None):
            self.array = None
            lib.load().ppm_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def profile(enable=True, reset=True):
    lib.load().ppm_profile_enable(1 if enable else 0)
    if reset:
        lib.load().ppm_profile_reset()


def profile_report():
    out = {}
    for i, name in enumerate(K_NAMES):
        ms, n = C.c_double(), C.c_long()
        lib.check(lib.load().ppm_profile_get(i, C.byref(ms), C.byref(n)))
        out[name] = {"ms": ms.value, "launches": n.value}
    return out
