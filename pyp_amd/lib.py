"""ctypes binding of libpypmatch.so (include/ppm.h).  There is no CPU fallback: if the shared
library is missing or no gfx950 device is visible, every entry point raises."""
import ctypes as C
import os

import numpy as np

from .abi import NCOL, STATS_COLS, K_NAMES, RefineCfg, ReconCfg, FinalCfg  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libpypmatch.so")

EXPORTS = [
    "ppm_init", "ppm_last_error", "ppm_version", "ppm_build_id", "ppm_device_mem_info", "ppm_reference_create", "ppm_reference_create_padded", "ppm_reference_create_weighted", "ppm_reference_destroy",
    "ppm_refine_batch", "ppm_refine_last_counts", "ppm_refine_note", "ppm_match_projections", "ppm_csp_refine", "ppm_sva_align", "ppm_accum_floats", "ppm_accum_create", "ppm_accum_destroy",
    "ppm_insert_batch", "ppm_accum_download", "ppm_accum_download_range", "ppm_accum_add", "ppm_accum_count", "ppm_accum_set_count",
    "ppm_finalize", "ppm_profile_enable", "ppm_profile_reset", "ppm_profile_get", "ppm_device_alloc",
    "ppm_device_free", "ppm_device_upload", "ppm_device_sync", "ppm_extract_boxes", "ppm_host_alloc", "ppm_host_free", "ppm_host_read",
    "ppm_comm_unique_id", "ppm_comm_create", "ppm_comm_count", "ppm_comm_destroy", "ppm_accum_reduce", "ppm_sva_insert", "ppm_sva_align_average",
]


class PpmError(RuntimeError):
    pass


_lib = None


def load():
    """dlopen the library and declare the prototypes (no GPU call is made here)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise PpmError(f"ERROR: {SO_PATH} is missing - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(there is no CPU fallback)")
    L = C.CDLL(SO_PATH)
    vp, ci, cf, cl = C.c_void_p, C.c_int, C.c_float, C.c_long
    L.ppm_init.argtypes = [ci]; L.ppm_init.restype = ci
    L.ppm_last_error.restype = C.c_char_p
    L.ppm_version.restype = C.c_char_p
    L.ppm_build_id.restype = C.c_char_p
    L.ppm_device_mem_info.argtypes = [vp, vp]; L.ppm_device_mem_info.restype = ci
    L.ppm_reference_create.argtypes = [vp, ci, cf]; L.ppm_reference_create.restype = vp
    L.ppm_reference_create_padded.argtypes = [vp, ci, cf, ci]; L.ppm_reference_create_padded.restype = vp
    L.ppm_reference_create_weighted.argtypes = [vp, ci, cf, ci, vp, ci]; L.ppm_reference_create_weighted.restype = vp
    L.ppm_reference_destroy.argtypes = [vp]; L.ppm_reference_destroy.restype = None
    L.ppm_refine_batch.argtypes = [vp, vp, vp, ci, ci, vp, vp]; L.ppm_refine_batch.restype = ci
    L.ppm_refine_last_counts.argtypes = [vp, vp, vp, vp, vp]; L.ppm_refine_last_counts.restype = ci
    L.ppm_refine_note.argtypes = [vp]; L.ppm_refine_note.restype = C.c_char_p
    L.ppm_match_projections.argtypes = [vp, vp, vp, ci, vp]; L.ppm_match_projections.restype = ci
    L.ppm_csp_refine.argtypes = [vp, vp, vp, vp, ci, ci, vp, vp, ci, vp, ci]; L.ppm_csp_refine.restype = ci
    L.ppm_sva_align.argtypes = [vp, vp, vp, ci, ci, vp, vp, vp]; L.ppm_sva_align.restype = ci
    L.ppm_sva_insert.argtypes = [vp, vp, vp, ci, ci, vp, vp, vp]; L.ppm_sva_insert.restype = ci
    L.ppm_sva_align_average.argtypes = [vp, vp, vp, vp, ci, ci, vp, vp, vp, vp]; L.ppm_sva_align_average.restype = ci
    L.ppm_accum_floats.argtypes = [ci]; L.ppm_accum_floats.restype = C.c_size_t
    L.ppm_accum_create.argtypes = [ci, cf, C.c_char_p, vp]; L.ppm_accum_create.restype = vp
    L.ppm_accum_destroy.argtypes = [vp]; L.ppm_accum_destroy.restype = None
    L.ppm_insert_batch.argtypes = [vp, vp, vp, ci, ci, vp]; L.ppm_insert_batch.restype = ci
    L.ppm_accum_download.argtypes = [vp, vp]; L.ppm_accum_download.restype = ci
    L.ppm_accum_download_range.argtypes = [vp, vp, C.c_size_t, C.c_size_t]; L.ppm_accum_download_range.restype = ci
    L.ppm_accum_add.argtypes = [vp, vp]; L.ppm_accum_add.restype = ci
    L.ppm_accum_count.argtypes = [vp, ci]; L.ppm_accum_count.restype = cl
    L.ppm_accum_set_count.argtypes = [vp, ci, cl]; L.ppm_accum_set_count.restype = None
    L.ppm_finalize.argtypes = [vp, vp, vp, vp, vp, vp]; L.ppm_finalize.restype = ci
    L.ppm_profile_enable.argtypes = [ci]; L.ppm_profile_enable.restype = None
    L.ppm_profile_reset.argtypes = []; L.ppm_profile_reset.restype = None
    L.ppm_profile_get.argtypes = [ci, vp, vp]; L.ppm_profile_get.restype = ci
    L.ppm_device_alloc.argtypes = [C.c_size_t]; L.ppm_device_alloc.restype = vp
    L.ppm_device_free.argtypes = [vp]; L.ppm_device_free.restype = None
    L.ppm_device_upload.argtypes = [vp, vp, C.c_size_t]; L.ppm_device_upload.restype = ci
    L.ppm_device_sync.argtypes = []; L.ppm_device_sync.restype = ci
    L.ppm_host_alloc.argtypes = [C.c_size_t]; L.ppm_host_alloc.restype = vp
    L.ppm_host_free.argtypes = [vp]; L.ppm_host_free.restype = None
    L.ppm_host_read.argtypes = [ci, C.c_longlong, vp, C.c_size_t, ci]; L.ppm_host_read.restype = ci
    L.ppm_extract_boxes.argtypes = [vp, ci, ci, ci, vp, ci, ci, C.c_double, C.c_double, ci, ci, vp, ci]; L.ppm_extract_boxes.restype = ci
    L.ppm_comm_unique_id.argtypes = [vp]; L.ppm_comm_unique_id.restype = ci
    L.ppm_comm_create.argtypes = [ci, ci, vp]; L.ppm_comm_create.restype = vp
    L.ppm_comm_count.argtypes = [vp]; L.ppm_comm_count.restype = ci
    L.ppm_comm_destroy.argtypes = [vp]; L.ppm_comm_destroy.restype = None
    L.ppm_accum_reduce.argtypes = [vp, vp, ci]; L.ppm_accum_reduce.restype = ci
    _lib = L
    return L


def last_error():
    return load().ppm_last_error().decode(errors="replace")


def check(rc):
    if rc != 0:
        raise PpmError(last_error() or f"ERROR: libpypmatch call failed ({rc})")


_inited = None


def init(device=0):
    global _inited
    if _inited != device:
        import sys
        w = sys.modules.get("pyp_amd.surface.warm")      # the executables start the device in a background thread (surface/warm.py)
        if w is not None:
            w.join_init()
        check(load().ppm_init(int(device)))
        _inited = device


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)
