"""FREALIGN text parameter files (`.par`): cclin (13 col), new (16), frealignx (17), each
optionally extended by 29 columns.

Templates and header lines restated from src/pyp/inout/metadata/frealign_parfile.py:62-135;
writer behaviour from `write_parameter_file` (:660-697); flavour is detected by the number
of columns of the first data line (`format_from_parfile`, :1578-1789); reading is whitespace
splitting of the non-`C` lines (`columns_from_parfile`, :1501-1575).
"""
import numpy as np

CCLIN, NEW, FREALIGNX = "cclin", "new", "frealignx"
_EXT = "%9d%9.2f%9.2f%9d%9.2f%9.2f" + "%10.4f" * 23
TEMPLATES = {
    CCLIN: "%7d%8.2f%8.2f%8.2f%10.2f%10.2f%8.0f%6d%9.1f%9.1f%8.2f%8.2f%10d%11.4f%8.2f%8.2f",
    NEW: "%7d%8.2f%8.2f%8.2f%10.2f%10.2f%8.0f%6d%9.1f%9.1f%8.2f%8.2f%10.0f%11.4f%8.2f%8.2f",
    FREALIGNX: "%7d%8.2f%8.2f%8.2f%10.2f%10.2f%8.0f%6d%9.1f%9.1f%8.2f%8.2f%8.2f%10.0f%11.4f%8.2f%8.2f",
}
NUM_COLS = {13: (CCLIN, False), 16: (NEW, False), 17: (FREALIGNX, False),
            42: (CCLIN, True), 45: (NEW, True), 46: (FREALIGNX, True)}

_NUMS16 = "C     1       2       3       4         5         6       7     8        9       10      11      12        13         14      15      16"
_NAMES16 = "C    NO     PSI   THETA     PHI       SHX       SHY     MAG  FILM      DF1      DF2  ANGAST     OCC      LOGP      SIGMA   SCORE  CHANGE"
_NUMS17 = "C     1       2       3       4         5         6       7     8        9       10      11      12      13        14         15      16      17"
_NAMES17 = "C    NO     PSI   THETA     PHI       SHX       SHY     MAG  FILM      DF1      DF2  ANGAST  PSHIFT     OCC      LOGP      SIGMA   SCORE  CHANGE"
_EXT_NAMES = ("   PTLIND    TILTAN    DOSEXX    SCANOR    CNFDNC    PTLCCX      AXIS     NORM0     NORM1     NORM2"
              + "".join("  MATRIX%02d" % i for i in range(16)) + "      PPSI    PTHETA      PPHI")


def _ext_nums(first):
    # column counters continue at width 9 for the first, 10 afterwards (frealign_parfile.py:113-125)
    return "%9d" % first + "".join("%10d" % i for i in range(first + 1, first + 29))


HEADERS = {
    (NEW, False): ["C FREALIGN NEW parameter file", _NUMS16, _NAMES16],
    (FREALIGNX, False): ["C FREALIGNX parameter file", _NUMS17, _NAMES17],
    (NEW, True): ["C FREALIGN EXTENDED NEW parameter file", _NUMS16 + _ext_nums(17), _NAMES16 + _EXT_NAMES],
    (FREALIGNX, True): ["C FREALIGN EXTENDED FREALIGNX parameter file", _NUMS17 + _ext_nums(18),
                        _NAMES17 + _EXT_NAMES],
}


def format_rows(data, version=NEW, extended=False):
    tmpl = TEMPLATES[version] + (_EXT if extended else "")
    return [tmpl % tuple(row.tolist()) for row in np.asarray(data, dtype=np.float64)]


def write(filename, data, version=NEW, extended=False, epilogue=()):
    """Write `data` (M x 13/16/17[+29]) with the reference's header lines; `epilogue` lines
    (already starting with 'C') are appended, e.g. the resolution table."""
    data = np.asarray(data, dtype=np.float64)
    want = {CCLIN: 13, NEW: 16, FREALIGNX: 17}[version] + (29 if extended else 0)
    if data.ndim != 2 or data.shape[1] != want:
        raise ValueError(f"ERROR: {version}{' extended' if extended else ''} needs {want} columns, got {data.shape}")
    hdr = HEADERS.get((version, extended), ["C FREALIGN CCLIN parameter file"])
    with open(filename, "w") as f:
        f.writelines(h + "\n" for h in hdr)
        f.writelines(r + "\n" for r in format_rows(data, version, extended))
        f.writelines(e.rstrip("\n") + "\n" for e in epilogue)


def read(filename):
    """Return (data float64 (M, ncols), version, extended, prologue, epilogue)."""
    prologue, epilogue, rows = [], [], []
    with open(filename) as f:
        for line in f.read().splitlines():
            if line.startswith("C"):
                (epilogue if rows else prologue).append(line)
            elif line.strip():
                if epilogue:        # comment lines in the middle belong to the body; keep only trailing ones
                    epilogue = []
                rows.append(line.split())
    if not rows:
        raise IOError(f"ERROR: {filename}: parameter file has no data lines")
    ncol = len(rows[0])
    if ncol not in NUM_COLS:
        raise IOError(f"ERROR: {filename}: unsupported number of columns {ncol}")
    if any(len(r) != ncol for r in rows):
        raise IOError(f"ERROR: {filename}: ragged parameter file")
    data = np.array(rows, dtype=np.float64)
    if np.isnan(data).any():
        raise IOError(f"ERROR: {filename}: parameter file has missing values NaN")
    version, extended = NUM_COLS[ncol]
    return data, version, extended, prologue, epilogue


# ---- conversion between .par rows and the 32-column .cistem layout -------------------------
def par_to_cistem(par, version, pixel_size, voltage_kv, cs_mm, amp_contrast):
    """Map NEW/FREALIGNX columns onto the standard .cistem columns (shifts stay in Angstrom,
    src/pyp/analysis/scores.py:693). MAG is dropped; FILM -> IMAGE_IS_ACTIVE (film index,
    cistem_star_file.py:1516-1524)."""
    from . import cistem as cs
    par = np.asarray(par, dtype=np.float64)
    m = par.shape[0]
    d = cs.default_rows(m, pixel_size, voltage_kv, cs_mm, amp_contrast)
    d[:, cs.COL["POSITION_IN_STACK"]] = par[:, 0]
    d[:, cs.COL["PSI"]], d[:, cs.COL["THETA"]], d[:, cs.COL["PHI"]] = par[:, 1], par[:, 2], par[:, 3]
    d[:, cs.COL["X_SHIFT"]], d[:, cs.COL["Y_SHIFT"]] = par[:, 4], par[:, 5]
    d[:, cs.COL["IMAGE_IS_ACTIVE"]] = par[:, 7]
    d[:, cs.COL["DEFOCUS_1"]], d[:, cs.COL["DEFOCUS_2"]], d[:, cs.COL["DEFOCUS_ANGLE"]] = par[:, 8], par[:, 9], par[:, 10]
    o = 11
    if version == FREALIGNX:
        d[:, cs.COL["PHASE_SHIFT"]] = par[:, 11]
        o = 12
    if version in (NEW, FREALIGNX):
        d[:, cs.COL["OCCUPANCY"]], d[:, cs.COL["LOGP"]] = par[:, o], par[:, o + 1]
        d[:, cs.COL["SIGMA"]], d[:, cs.COL["SCORE"]] = par[:, o + 2], par[:, o + 3]
    return d


def cistem_to_par(rows, version=NEW, mag=10000.0, change=None):
    from . import cistem as cs
    rows = np.asarray(rows, dtype=np.float64)
    m = rows.shape[0]
    n = {NEW: 16, FREALIGNX: 17}[version]
    p = np.zeros((m, n))
    p[:, 0] = rows[:, cs.COL["POSITION_IN_STACK"]]
    p[:, 1], p[:, 2], p[:, 3] = rows[:, cs.COL["PSI"]], rows[:, cs.COL["THETA"]], rows[:, cs.COL["PHI"]]
    p[:, 4], p[:, 5] = rows[:, cs.COL["X_SHIFT"]], rows[:, cs.COL["Y_SHIFT"]]
    p[:, 6] = mag
    p[:, 7] = rows[:, cs.COL["IMAGE_IS_ACTIVE"]]
    p[:, 8], p[:, 9], p[:, 10] = rows[:, cs.COL["DEFOCUS_1"]], rows[:, cs.COL["DEFOCUS_2"]], rows[:, cs.COL["DEFOCUS_ANGLE"]]
    o = 11
    if version == FREALIGNX:
        p[:, 11] = rows[:, cs.COL["PHASE_SHIFT"]]
        o = 12
    p[:, o], p[:, o + 1] = rows[:, cs.COL["OCCUPANCY"]], rows[:, cs.COL["LOGP"]]
    p[:, o + 2], p[:, o + 3] = rows[:, cs.COL["SIGMA"]], rows[:, cs.COL["SCORE"]]
    p[:, o + 4] = 0.0 if change is None else change
    return p
