"""MRC2000 image / stack / volume I/O as the PYP pipeline writes and reads it.

Byte layout follows the 1024-byte header the reference lists field by field at
src/pyp/inout/image/mrc.py:74-116 and fills in `write()` (:537-560,
`updateHeaderDefaults` :375-382, `updateHeaderUsingArray` :385-431): little-endian,
no extended header, data C-ordered with x fastest; a particle stack is nz = M sections.
"""
import os

import numpy as np

_MODE2DTYPE = {0: np.uint8, 1: np.int16, 2: np.float32, 4: np.complex64, 6: np.uint16}
_HEADER = np.dtype([
    ("nx", "<i4"), ("ny", "<i4"), ("nz", "<i4"), ("mode", "<i4"),
    ("nxstart", "<i4"), ("nystart", "<i4"), ("nzstart", "<i4"),
    ("mx", "<i4"), ("my", "<i4"), ("mz", "<i4"),
    ("xlen", "<f4"), ("ylen", "<f4"), ("zlen", "<f4"),
    ("alpha", "<f4"), ("beta", "<f4"), ("gamma", "<f4"),
    ("mapc", "<i4"), ("mapr", "<i4"), ("maps", "<i4"),
    ("amin", "<f4"), ("amax", "<f4"), ("amean", "<f4"),
    ("ispg", "<i4"), ("nsymbt", "<i4"), ("extra", "S100"),
    ("xorigin", "<f4"), ("yorigin", "<f4"), ("zorigin", "<f4"),
    ("map", "S4"), ("byteorder", "<i4"), ("rms", "<f4"), ("nlabels", "<i4"),
    ("labels", "S800"),
])
assert _HEADER.itemsize == 1024
_LITTLE_STAMP = 0x44440000


def make_header(shape, dtype=np.float32, pixel_size=None, stats=None):
    """Header record for an array of `shape` (…, ny, nx). `pixel_size` None -> 1.0 A/px (xlen = nx)."""
    shape = tuple(int(s) for s in shape)
    nx = shape[-1]
    ny = shape[-2] if len(shape) > 1 else 1
    nz = shape[-3] if len(shape) > 2 else 1
    h = np.zeros((), dtype=_HEADER)
    h["nx"], h["ny"], h["nz"] = nx, ny, nz
    h["mode"] = {np.dtype(np.uint8): 0, np.dtype(np.int16): 1, np.dtype(np.float32): 2,
                 np.dtype(np.complex64): 4, np.dtype(np.uint16): 6}[np.dtype(dtype)]
    h["mx"], h["my"], h["mz"] = nx, ny, nz
    ps = 1.0 if pixel_size is None else float(pixel_size)
    h["xlen"], h["ylen"], h["zlen"] = nx * ps, ny * ps, nz * ps
    h["alpha"] = h["beta"] = h["gamma"] = 90.0
    h["mapc"], h["mapr"], h["maps"] = 1, 2, 3
    h["map"] = b"MAP"          # the reference strips the blank: "MAP" + NUL (mrc.py:381, :483)
    h["byteorder"] = _LITTLE_STAMP
    if stats is not None:
        h["amin"], h["amax"], h["amean"], h["rms"] = stats
    return h


def write(a, filename, pixel_size=None):
    """Write ndarray `a` (2D image, 3D volume or stack) as MRC. float64 is stored as float32
    like the reference's numpy2mrc map (mrc.py:86-107)."""
    a = np.asarray(a)
    if a.dtype in (np.float64, np.int32, np.int64, np.uint32, np.uint64):
        a = a.astype(np.float32)
    # numpy.min/max/mean/std on the array in its own dtype, like the reference's arraystats.all
    # (src/pyp/inout/image/utils/arraystats.py:52-98) so the header bytes agree to the last bit
    s = np.abs(a) if a.dtype == np.complex64 else a
    stats = (np.min(s), np.max(s), np.mean(s), np.std(s)) if a.size else (0, 0, 0, 0)
    h = make_header(a.shape, a.dtype, pixel_size, stats)
    with open(filename, "wb") as f:
        f.write(h.tobytes())
        f.write(np.ascontiguousarray(a).astype(a.dtype.newbyteorder("<"), copy=False).tobytes())


def read_header(filename):
    """Return the header as a dict of python scalars plus 'dtype', 'shape', 'pixel_size'."""
    with open(filename, "rb") as f:
        raw = f.read(1024)
    if len(raw) < 1024:
        raise IOError(f"ERROR: {filename}: truncated MRC header")
    h = np.frombuffer(raw, dtype=_HEADER)[0]
    swapped = int(h["mode"]) not in _MODE2DTYPE
    if swapped:
        h = np.frombuffer(raw, dtype=_HEADER.newbyteorder(">"))[0]
        if int(h["mode"]) not in _MODE2DTYPE:
            raise IOError(f"ERROR: {filename}: unsupported MRC mode")
    d = {k: (h[k].item() if h[k].dtype.kind != "S" else bytes(h[k])) for k in _HEADER.names}
    dt = np.dtype(_MODE2DTYPE[d["mode"]]).newbyteorder(">" if swapped else "<")
    d["dtype"] = dt
    d["shape"] = (d["nz"], d["ny"], d["nx"]) if d["nz"] > 1 else ((d["ny"], d["nx"]) if d["ny"] > 1 else (d["nx"],))
    d["pixel_size"] = d["xlen"] / d["nx"] if d["nx"] > 0 and d["xlen"] > 0 else 1.0
    d["data_offset"] = 1024 + d["nsymbt"]
    return d


def read(filename, first=None, last=None):
    """Read the whole file, or sections first..last (0-based, inclusive) of a stack/volume."""
    h = read_header(filename)
    nz, ny, nx = h["nz"], h["ny"], h["nx"]
    lo = 0 if first is None else int(first)
    hi = nz - 1 if last is None else int(last)
    if lo < 0 or hi >= nz or hi < lo:
        raise IOError(f"ERROR: {filename}: section range {lo}..{hi} outside 0..{nz - 1}")
    sec = ny * nx * h["dtype"].itemsize
    need = h["data_offset"] + nz * sec
    if os.path.getsize(filename) < need:
        raise IOError(f"ERROR: {filename}: file shorter than header claims")
    with open(filename, "rb") as f:
        f.seek(h["data_offset"] + lo * sec)
        a = np.fromfile(f, dtype=h["dtype"], count=(hi - lo + 1) * ny * nx)
    a = a.astype(h["dtype"].newbyteorder("="), copy=False)
    if first is None and last is None:
        return a.reshape(h["shape"])
    return a.reshape((hi - lo + 1, ny, nx))


def mmap(filename):
    """Read-only memory map of the data block (stack: (nz, ny, nx))."""
    h = read_header(filename)
    return np.memmap(filename, dtype=h["dtype"], mode="r", offset=h["data_offset"],
                     shape=(h["nz"], h["ny"], h["nx"]))


def create(filename, shape, pixel_size=None, dtype=np.float32):
    """Pre-size a new MRC file and return a writable memory map of its data block (fill it section by section, `flush()`,
    then `set_statistics`): a stack larger than host memory is written without ever being held."""
    h = make_header(shape, dtype, pixel_size)
    shape = tuple(int(s) for s in shape)
    with open(filename, "wb") as f:
        f.write(h.tobytes())
        f.truncate(1024 + int(np.prod(shape)) * np.dtype(dtype).itemsize)
    return np.memmap(filename, dtype=np.dtype(dtype).newbyteorder("<"), mode="r+", offset=1024, shape=shape)


def set_statistics(filename, amin, amax, amean, rms):
    """Fill amin / amax / amean / rms of an existing file's header (after a streamed `create`)."""
    with open(filename, "r+b") as f:
        h = np.frombuffer(bytearray(f.read(1024)), dtype=_HEADER).copy()
        h["amin"], h["amax"], h["amean"], h["rms"] = amin, amax, amean, rms
        f.seek(0)
        f.write(h.tobytes())
