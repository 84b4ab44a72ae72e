"""cisTEM binary parameter files (`.cistem`, `_extended.cistem`) as PYP reads/writes them.

Layout restated from src/pyp/inout/metadata/cistem_star_file.py:114-123 (record dtypes),
:596-628 (the 32 standard columns and their order), :694-776 (main file read/write) and
:276-381 (the two-block extended file). Little-endian:

    int32 ncols; int32 nrows; ncols x {int64 column_code; int8 dtype_code}; rows packed.
"""
import os

import numpy as np

# dtype codes (cistem_star_file.py:17-27)
INTEGER, FLOAT, INTEGER_UNSIGNED = 2, 3, 9

# column codes (cistem_star_file.py:29-96)
POSITION_IN_STACK, IMAGE_IS_ACTIVE, PSI, X_SHIFT, Y_SHIFT = 1, 2, 4, 8, 16
DEFOCUS_1, DEFOCUS_2, DEFOCUS_ANGLE, PHASE_SHIFT, OCCUPANCY = 32, 64, 128, 256, 512
LOGP, SIGMA, SCORE, PIXEL_SIZE = 1024, 2048, 4096, 16384
MICROSCOPE_VOLTAGE, MICROSCOPE_CS, AMPLITUDE_CONTRAST = 32768, 65536, 131072
BEAM_TILT_X, BEAM_TILT_Y, IMAGE_SHIFT_X, IMAGE_SHIFT_Y = 262144, 524288, 1048576, 2097152
THETA, PHI = 4194304, 8388608
ORIGINAL_X_POSITION, ORIGINAL_Y_POSITION = 8589934592, 17179869184
IMIND, PIND, TIND, RIND, FIND, FSHIFT_X, FSHIFT_Y = 20, 15, 35, 70, 55, 11, 121
PSHIFT_X, PSHIFT_Y, PSHIFT_Z, PPSI, PTHETA, PPHI = 3, 9, 27, 81, 273, 819
ORIGINAL_X_POSITION_3D, ORIGINAL_Y_POSITION_3D, ORIGINAL_Z_POSITION_3D = 2457, 7371, 22113
PSCORE, POCC = 66339, 199017
TSHIFT_X, TSHIFT_Y, TILTANG, TILTAXIS = 7, 49, 343, 2401

# the 32 standard columns, in file order (cistem_star_file.py:596-628)
COLUMNS = [
    ("POSITION_IN_STACK", POSITION_IN_STACK, INTEGER_UNSIGNED), ("PSI", PSI, FLOAT), ("THETA", THETA, FLOAT),
    ("PHI", PHI, FLOAT), ("X_SHIFT", X_SHIFT, FLOAT), ("Y_SHIFT", Y_SHIFT, FLOAT),
    ("DEFOCUS_1", DEFOCUS_1, FLOAT), ("DEFOCUS_2", DEFOCUS_2, FLOAT), ("DEFOCUS_ANGLE", DEFOCUS_ANGLE, FLOAT),
    ("PHASE_SHIFT", PHASE_SHIFT, FLOAT), ("IMAGE_IS_ACTIVE", IMAGE_IS_ACTIVE, INTEGER),
    ("OCCUPANCY", OCCUPANCY, FLOAT), ("LOGP", LOGP, FLOAT), ("SIGMA", SIGMA, FLOAT), ("SCORE", SCORE, FLOAT),
    ("PIXEL_SIZE", PIXEL_SIZE, FLOAT), ("MICROSCOPE_VOLTAGE", MICROSCOPE_VOLTAGE, FLOAT),
    ("MICROSCOPE_CS", MICROSCOPE_CS, FLOAT), ("AMPLITUDE_CONTRAST", AMPLITUDE_CONTRAST, FLOAT),
    ("BEAM_TILT_X", BEAM_TILT_X, FLOAT), ("BEAM_TILT_Y", BEAM_TILT_Y, FLOAT),
    ("IMAGE_SHIFT_X", IMAGE_SHIFT_X, FLOAT), ("IMAGE_SHIFT_Y", IMAGE_SHIFT_Y, FLOAT),
    ("ORIGINAL_X_POSITION", ORIGINAL_X_POSITION, FLOAT), ("ORIGINAL_Y_POSITION", ORIGINAL_Y_POSITION, FLOAT),
    ("IMIND", IMIND, INTEGER), ("PIND", PIND, INTEGER), ("TIND", TIND, INTEGER), ("RIND", RIND, INTEGER),
    ("FIND", FIND, INTEGER), ("FSHIFT_X", FSHIFT_X, FLOAT), ("FSHIFT_Y", FSHIFT_Y, FLOAT),
]
NAMES = [c[0] for c in COLUMNS]
COL = {n: i for i, n in enumerate(NAMES)}
NCOL = len(COLUMNS)

PARTICLE_COLUMNS = [PIND, PSHIFT_X, PSHIFT_Y, PSHIFT_Z, PPSI, PTHETA, PPHI, ORIGINAL_X_POSITION_3D,
                    ORIGINAL_Y_POSITION_3D, ORIGINAL_Z_POSITION_3D, PSCORE, POCC]      # :247
TILT_COLUMNS = [TIND, RIND, TSHIFT_X, TSHIFT_Y, TILTANG, TILTAXIS]                      # :248
_KNOWN = {c[1]: c[2] for c in COLUMNS}
_KNOWN.update({c: FLOAT for c in PARTICLE_COLUMNS + TILT_COLUMNS})
_KNOWN.update({PIND: INTEGER, TIND: INTEGER, RIND: INTEGER, 8192: FLOAT})   # 8192 = SCORE_CHANGE
_NP = {INTEGER: "<i4", FLOAT: "<f4", INTEGER_UNSIGNED: "<u4"}
_COLREC = np.dtype([("code", "<i8"), ("dtype", "<i1")])


def _row_dtype(codes):
    return np.dtype([(str(c), _NP[_KNOWN[c]]) for c in codes])


def _write_block(f, codes, data):
    data = np.asarray(data, dtype=np.float64)
    if data.ndim != 2 or data.shape[1] != len(codes):
        raise ValueError(f"ERROR: data has {data.shape} but {len(codes)} columns declared")
    f.write(np.array([len(codes), data.shape[0]], dtype="<i4").tobytes())
    f.write(np.array([(c, _KNOWN[c]) for c in codes], dtype=_COLREC).tobytes())
    # every column is 4 bytes wide: one [M, ncols] word matrix filled per dtype group (a structured array filled column by
    # column costs 0.16 s per 100 k rows; this 25 ms).  float64 -> target casts are numpy's, like unstructured_to_structured
    words = data.astype("<f4").view("<u4")            # the float columns in one pass; the few integer columns are redone below
    for j, c in enumerate(codes):
        if _KNOWN[c] != FLOAT:
            words[:, j] = data[:, j].astype(_NP[_KNOWN[c]]).view("<u4")
    f.write(words.tobytes())


def _read_block(buf, pos):
    ncols, nrows = np.frombuffer(buf, dtype="<i4", count=2, offset=pos)
    pos += 8
    if ncols <= 0 or nrows < 0:
        raise IOError("ERROR: binary file is broken (bad dimensions)")
    hdr = np.frombuffer(buf, dtype=_COLREC, count=int(ncols), offset=pos)
    pos += 9 * int(ncols)
    codes = [int(c) for c in hdr["code"]]
    for c in codes:
        if c not in _KNOWN:
            raise IOError(f"ERROR: binary file contains unrecognized header. Column code = {c}")
    dt = _row_dtype(codes)
    need = int(nrows) * dt.itemsize
    if len(buf) - pos < need:
        raise IOError("ERROR: binary file is broken (short data block)")
    words = np.frombuffer(buf, dtype="<u4", count=int(nrows) * int(ncols), offset=pos).reshape(int(nrows), int(ncols))
    pos += need
    data = words.view("<f4").astype(np.float64)
    for j, c in enumerate(codes):
        if _KNOWN[c] != FLOAT:
            data[:, j] = words[:, j].view(_NP[_KNOWN[c]])
    return codes, data, pos


def write_parameters(filename, data):
    """Write an (M, 32) array in the standard column order to `filename` (.cistem)."""
    if not str(filename).endswith(".cistem"):
        raise ValueError(f"ERROR: output {filename} must have .cistem extension")
    data = np.asarray(data, dtype=np.float64)
    if data.ndim != 2 or data.shape[1] != NCOL or data.shape[0] == 0:
        raise ValueError(f"ERROR: expected (M>0, {NCOL}) array, got {data.shape}")
    tmp = str(filename) + ".tmp%d" % os.getpid()
    with open(tmp, "wb") as f:
        _write_block(f, [c[1] for c in COLUMNS], data)
    os.replace(tmp, filename)


def read_parameters(filename):
    """Return (M, ncols) float64 array; columns reordered to the standard 32 if the file has them."""
    with open(filename, "rb") as f:
        buf = f.read()
    if len(buf) < 8:
        raise IOError(f"ERROR: {filename}: binary file is broken")
    codes, data, _ = _read_block(buf, 0)
    std = [c[1] for c in COLUMNS]
    if codes == std:
        return data
    if set(std) <= set(codes):
        return data[:, [codes.index(c) for c in std]]
    raise IOError(f"ERROR: {filename}: missing standard columns")


def merge_parameters(filenames):
    """Concatenate range files and sort by POSITION_IN_STACK (cistem_star_file.py:655-692)."""
    arr = np.vstack([read_parameters(f) for f in filenames])
    return arr[np.argsort(arr[:, COL["POSITION_IN_STACK"]], kind="stable")]


def write_extended(filename, particles, tilts):
    """`particles`: (P, 12) array in PARTICLE_COLUMNS order; `tilts`: (T, 6) in TILT_COLUMNS order."""
    with open(filename, "wb") as f:
        for block, codes, data in ((PIND, PARTICLE_COLUMNS, particles), (TIND, TILT_COLUMNS, tilts)):
            f.write(np.array([block], dtype="<i8").tobytes())
            _write_block(f, codes, np.asarray(data, dtype=np.float64).reshape(-1, len(codes)))


def read_extended(filename):
    """Return {'particles': (P,12) array, 'tilts': (T,6) array}."""
    with open(filename, "rb") as f:
        buf = f.read()
    pos, out = 0, {}
    for _ in range(2):
        block = int(np.frombuffer(buf, dtype="<i8", count=1, offset=pos)[0])
        pos += 8
        codes, data, pos = _read_block(buf, pos)
        if block == PIND:
            out["particles"] = data
        elif block == TIND:
            out["tilts"] = data
        else:
            raise IOError(f"ERROR: {filename}: unknown block type {block}")
    return out


def default_rows(m, pixel_size, voltage_kv, cs_mm, amp_contrast):
    """Rows shaped like SPA from-scratch rows (src/pyp/inout/metadata/core.py:1324-1608)."""
    d = np.zeros((m, NCOL), dtype=np.float64)
    d[:, COL["POSITION_IN_STACK"]] = np.arange(1, m + 1)
    d[:, COL["OCCUPANCY"]] = 100.0
    d[:, COL["SIGMA"]] = 0.5
    d[:, COL["SCORE"]] = 0.5
    d[:, COL["PIXEL_SIZE"]] = pixel_size
    d[:, COL["MICROSCOPE_VOLTAGE"]] = voltage_kv
    d[:, COL["MICROSCOPE_CS"]] = cs_mm
    d[:, COL["AMPLITUDE_CONTRAST"]] = amp_contrast
    d[:, COL["PIND"]] = np.arange(m)
    return d
