"""Sub-tomogram averaging host side (BASELINE config 5): the `*_volumes.txt` tables 3DAVG exchanges with PYP,
the conversion between a table line (normal, 4 x 4 alignment matrix) and the particle pose the rest of the path uses, the
protocol XML fields that set the metric, and the alignment of all sub-volumes of a table against a reference on the GPU.

Table format: tab-separated, header `number lwedge uwedge posX posY posZ geomX geomY geomZ normalX normalY normalZ matrix[0..15]
magnification[0..2] cutOffset filename` (src/pyp/detect/tomo/core.py:352, src/pyp_main.py:1003); PYP reads normals from
columns 9-11, the matrix from 12-27 and re-uses cutOffset (31) for the correlation score (src/pyp/inout/metadata/core.py:2578-2610).
The line -> pose relation is the one the reference applies in spa_euler_angles (src/pyp/analysis/geometry/core.py:238-470) at
tilt 0 / axis 0 and is pinned by golden vectors it produced (tests/golden/golden_r02.json "sva_matrix_to_particle"):

    N = Sy R^-1 norm^-1 Sy,   norm = Rz(-normalZ) Rx(-normalX) Rz(-normalY),   Sy = diag(1, -1, 1)
    p = norm_rev^-1 R_rev^-1 tr,   tr = (-a0, a1, -a2),  a = R^-1 t,   R_rev[i][j] = (-1)^(i+j) R[j][i],  norm_rev = Rz(-nY) Rx(-nX) Rz(-nZ)

(R, t = rotation and translation of the matrix).  This is NOT a drop-in of MPI_Classification: its binary side files
(`*_averages.bin`) are not described anywhere in the reference; what is provided is the mode-3 iteration on the same tables:
"align all volumes to the reference" (align_table -> `*_alignments_to_reference_0.txt`, src/pyp_main.py:3104) and the average of the
aligned sub-volumes that becomes the next reference (`*_refined_selected_average_0.mrc` + `..._filtered.mrc`,
src/pyp/refine/tomo_avg/sub_tomo_avg.py:79-94, src/pyp_main.py:3076-3100; average_from_accumulator).
"""
import os
import xml.etree.ElementTree as ET

import numpy as np

from .abi import SvaCfg
from .formats import mrc
from .synth import rot_xyz

HEADER = ["number", "lwedge", "uwedge", "posX", "posY", "posZ", "geomX", "geomY", "geomZ", "normalX", "normalY", "normalZ"] + \
         ["matrix[%d]" % i for i in range(16)] + ["magnification[0]", "magnification[1]", "magnification[2]", "cutOffset", "filename"]
_SY = np.diag([1.0, -1.0, 1.0])


def read_volumes(path):
    """-> (table (V, 32) float64 in HEADER order without the file name, filenames list)."""
    rows, names = [], []
    with open(path) as f:
        for line in f:
            if not line.strip() or line.startswith("number"):
                continue
            parts = line.rstrip("\n").split("\t")
            if len(parts) < 33:
                parts = line.split()
            if len(parts) != 33:
                raise IOError(f"ERROR: {path}: expected 33 tab-separated columns, got {len(parts)}")
            rows.append([float(x) for x in parts[:32]])
            names.append(parts[32].strip())
    if not rows:
        raise IOError(f"ERROR: {path}: no sub-volumes listed")
    return np.array(rows, dtype=np.float64), names


def write_volumes(path, table, names):
    with open(path, "w") as f:
        f.write("\t".join(HEADER) + "\n")
        for r, n in zip(np.asarray(table, dtype=np.float64), names):
            f.write("%d\t" % int(r[0]) + "\t".join("%.6f" % x for x in r[1:32]) + "\t" + n + "\n")


def _norm_matrices(normal):
    nx, ny, nz = (float(x) for x in normal)
    norm = rot_xyz(2, -nz) @ rot_xyz(0, -nx) @ rot_xyz(2, -ny)
    norm_rev = rot_xyz(2, -ny) @ rot_xyz(0, -nx) @ rot_xyz(2, -nz)
    return norm, norm_rev


def _reverse(R):
    s = np.array([[1, -1, 1], [-1, 1, -1], [1, -1, 1]], dtype=np.float64)
    return (R.T) * s


def line_to_pose(normal, matrix):
    """(N (3, 3), p (3,)) of a table line: the pose convention of ppm_sva_align / the particle block."""
    m = np.asarray(matrix, dtype=np.float64).reshape(4, 4)
    R, t = m[:3, :3], m[:3, 3]
    norm, norm_rev = _norm_matrices(normal)
    N = _SY @ np.linalg.inv(R) @ np.linalg.inv(norm) @ _SY
    a = np.linalg.inv(R) @ t
    tr = np.array([-a[0], a[1], -a[2]])
    p = np.linalg.inv(norm_rev) @ np.linalg.inv(_reverse(R)) @ tr
    return N, p


def pose_to_matrix(N, p, normal):
    """The 16 matrix values of a table line that line_to_pose maps back to (N, p), for the line's normal."""
    norm, norm_rev = _norm_matrices(normal)
    R = np.linalg.inv(_SY @ np.asarray(N, dtype=np.float64).reshape(3, 3) @ _SY @ norm)
    tr = _reverse(R) @ norm_rev @ np.asarray(p, dtype=np.float64)
    a = np.array([-tr[0], tr[1], -tr[2]])
    m = np.eye(4)
    m[:3, :3], m[:3, 3] = R, R @ a
    return m.ravel()


def particle_from_pose(N, p):
    """Stored particle parameters (psi, theta, phi, shift x, y, z) of a pose (what spa_euler_angles hands to the particle block)."""
    from .synth import angles_from_matrix
    a = angles_from_matrix(np.asarray(N, dtype=np.float64).reshape(3, 3))
    return np.array([-a[0], -a[1], -a[2], p[0], p[1], p[2]])


def cfg_from_xml(xml_path, box, pixel_size=1.0, mode=None):
    """ppm_sva_cfg from a 3DAVG protocol file (src/pyp/refine/3DAVG/iteration_*_mode_*.xml as patched by parse_xml,
    src/pyp/refine/tomo_avg/sub_tomo_avg.py:318-465): image window, band-pass, missing wedge, search ranges of the mode's section."""
    root = ET.parse(xml_path).getroot()
    gen = root.find("general")
    m = int(gen.find("mode").text) if mode is None else int(mode)
    sec_name = ["mra", "class", "refine", "mra"][m]          # mode 0 (re-centring) uses the mra fields (:349-351)
    sec = root.find(sec_name)

    def val(tag, default=0.0):
        e = sec.find(f"{sec_name}_{tag}")
        return float(e.text) if e is not None and e.text not in (None, "") else default
    metric = gen.find("metric")

    def mval(tag, default):
        e = metric.find(tag) if metric is not None else None
        return int(float(e.text)) if e is not None and e.text not in (None, "") else default
    wedge = mval("use_missing_wedge", 1)
    # metric/alignment_mode (iteration_002_mode_3.xml:29-38): 0 = global rotation and translation search, 1 = refinement only,
    # 2 = translation only -> ppm_sva_cfg.search_mode 1 / 0 / 2; a protocol without the field refines
    search_mode = {0: 1, 1: 0, 2: 2}.get(mval("alignment_mode", 1), 0)
    return SvaCfg.make(box, pixel_size, window=(val("image_window_x"), val("image_window_y"), val("image_window_z")),
                       window_sigma=val("image_window_sigma"), highpass=(val("high_pass_cutoff"), val("high_pass_decay")),
                       lowpass=(val("low_pass_cutoff"), val("low_pass_decay")), use_missing_wedge=wedge,
                       tol_angle=val("out_of_plane_search_range"), tol_shift=val("shifts_tolerance"),
                       search_mode=search_mode, n_candidates=mval("number_of_candidate_peaks_to_search", 25))


def band_weights(cfg, n):
    """The protocol's band-pass (Gaussian roll-offs outside highpass .. lowpass, cycles per pixel) on the n^3 FFT grid - the weights
    the alignment metric applies (ppm_sva_cfg)."""
    k = np.fft.fftfreq(n)
    s = np.sqrt(k[:, None, None] ** 2 + k[None, :, None] ** 2 + k[None, None, :] ** 2)
    w = np.ones_like(s)
    if cfg.highpass_cutoff > 0:
        d = np.clip(cfg.highpass_cutoff - s, 0, None)
        w *= np.exp(-d * d / (2 * cfg.highpass_decay ** 2)) if cfg.highpass_decay > 0 else (d == 0)
    if cfg.lowpass_cutoff > 0:
        d = np.clip(s - cfg.lowpass_cutoff, 0, None)
        w *= np.exp(-d * d / (2 * cfg.lowpass_decay ** 2)) if cfg.lowpass_decay > 0 else (d == 0)
    return w


def filtered_map(vol, cfg):
    """A map as the alignment metric sees it: real-space window of the protocol, then its band-pass (what `Test_Metric_Filter` shows
    the user for the mode's settings, src/pyp/refine/tomo_avg/sub_tomo_avg.py:485-497; the `_filtered.mrc` twin of an average)."""
    n = vol.shape[0]
    c = np.abs(np.arange(n) - n // 2).astype(np.float64)
    win = np.ones((n, n, n))
    for axis, half in zip((2, 1, 0), cfg.window):           # window x, y, z on array axes 2, 1, 0
        if not half > 0:
            continue
        d = np.clip(c - half, 0, None)
        w1 = np.exp(-d * d / (2 * cfg.window_sigma ** 2)) if cfg.window_sigma > 0 else (d == 0).astype(np.float64)
        shape = [1, 1, 1]; shape[axis] = n
        win = win * w1.reshape(shape)
    f = np.fft.fftn((vol - vol.mean()) * win) * band_weights(cfg, n)
    return np.fft.ifftn(f).real.astype(np.float32)


def average_from_accumulator(acc, cfg, prefix, pixel_size=1.0, outer_radius=0.0):
    """Finalise the sub-tomogram average held by `acc` (host.Accumulator filled by sva_insert / align_table) and write
    `<prefix>.mrc` (FSC-weighted average of all sub-volumes: the next iteration's reference), `<prefix>_filtered.mrc` (the same through
    the protocol's window and band-pass), `<prefix>_half1.mrc` / `_half2.mrc` (odd / even sub-volumes) and `<prefix>_statistics.txt`
    (the 7-column table of merge3d: shell, resolution, ring radius, FSC, part-FSC, part-SSNR, rec-SSNR).  Returns (average, statistics)."""
    from .abi import FinalCfg
    from .surface.cli import format_statistics_rows
    h_even, h_odd, avg, stats = acc.finalize(FinalCfg(molecular_mass_kda=0.0, inner_radius=0.0, outer_radius=float(outer_radius), mask_falloff=0.0))
    mrc.write(avg, prefix + ".mrc", pixel_size=pixel_size)
    mrc.write(filtered_map(avg, cfg), prefix + "_filtered.mrc", pixel_size=pixel_size)
    mrc.write(h_odd, prefix + "_half1.mrc", pixel_size=pixel_size)
    mrc.write(h_even, prefix + "_half2.mrc", pixel_size=pixel_size)
    with open(prefix + "_statistics.txt", "w") as f:
        f.write("C  NO.   RESOL  RING RAD       FSC  Part_FSC Part_SSNR  Rec_SSNR\n")
        f.write(format_statistics_rows(stats))
    return avg, stats


def align_table(reference, table, names, cfg, base_dir=".", device=0, chunk=256, max_band_px=None, accumulator=None):
    """Align every sub-volume of a table to `reference` (N^3 array): returns the refined table (matrix columns replaced, the
    correlation score in cutOffset) and the scores.  Sub-volumes are read chunk by chunk (10 k x 192^3 is 283 GB) into two
    page-locked buffers, the next chunk by a reader thread while the current one is aligned; inside a call the library uploads
    2 GB at a time while it searches the previous 2 GB.  accumulator: a host.Accumulator of the same box - every chunk is added to
    the sub-tomogram average at its refined poses while it is still in DEVICE memory (ppm_sva_align_average; half-map = parity of the
    table's `number` column), so the volumes are read and uploaded once per iteration; finish with average_from_accumulator."""
    import threading
    from . import host
    n = int(cfg.box)
    ref = host.Reference(reference, n / 2 if max_band_px is None else max_band_px, device=device)
    out = np.array(table, dtype=np.float64, copy=True)
    scores = np.zeros(len(out))
    chunk = max(1, min(int(chunk), len(out)))
    bufs = [host.PinnedBuffer(chunk * n ** 3, device) for _ in range(2 if len(out) > chunk else 1)]

    L = host.lib.load()
    nthreads = max(1, min(16, int(os.environ.get("PPM_IO_THREADS", "8"))))

    def fill(b, lo, hi):
        vols = bufs[b].array[:(hi - lo) * n ** 3].reshape(hi - lo, n, n, n)
        for k in range(lo, hi):
            fn = names[k] if os.path.isabs(names[k]) else os.path.join(base_dir, names[k])
            h = mrc.read_header(fn)
            if h["shape"] != (n, n, n):
                raise ValueError(f"ERROR: {fn} is {h['shape']}, expected {n}^3")
            if h["dtype"] == np.dtype("<f4") and os.path.getsize(fn) >= h["data_offset"] + 4 * n ** 3:
                # little-endian float32 (what PYP's extraction writes): the library's reader pool fills the page-locked buffer directly
                fd = os.open(fn, os.O_RDONLY)
                try:
                    if L.ppm_host_read(fd, h["data_offset"], bufs[b].ptr + (k - lo) * 4 * n ** 3, 4 * n ** 3, nthreads) != 0:
                        raise IOError(f"ERROR: reading {fn} failed: " + host.lib.last_error())
                finally:
                    os.close(fd)
            else:
                vols[k - lo] = mrc.read(fn)
        return vols

    t = None
    try:
        cur, b = fill(0, 0, min(chunk, len(out))), 0
        for lo in range(0, len(out), chunk):
            hi = min(lo + chunk, len(out))
            nxt, err, t = [None], [None], None
            if hi < len(out):
                def work(bb=1 - b, a=hi, e=min(hi + chunk, len(out))):
                    try:
                        nxt[0] = fill(bb, a, e)
                    except Exception as ex:      # surfaced in the caller's thread below
                        err[0] = ex
                t = threading.Thread(target=work)
                t.start()
            poses = np.zeros((hi - lo, 12))
            for k in range(lo, hi):
                N, p = line_to_pose(out[k, 9:12], out[k, 12:28])
                poses[k - lo, :9], poses[k - lo, 9:] = N.ravel(), p
            got, sc = ref.sva_align(cfg, cur, out[lo:hi, 1:3].astype(np.float32), poses, accumulator=accumulator,
                                    index=out[lo:hi, 0].astype(np.int64) if accumulator is not None else None)
            for k in range(lo, hi):
                out[k, 12:28] = pose_to_matrix(got[k - lo, :9], got[k - lo, 9:], out[k, 9:12])
                out[k, 31] = sc[k - lo]
            scores[lo:hi] = sc
            if t is not None:
                t.join()
                t = None
                if err[0] is not None:
                    raise err[0]
                cur, b = nxt[0], 1 - b
    finally:
        if t is not None:           # an error above: the reader must be done with the pinned buffers before they are freed
            t.join()
        ref.close()
        for pb in bufs:
            pb.close()
    return out, scores


# ------------------------------------------------------------------------------------------------ classification (3DAVG mode 1)
class GpuBackend:
    """The two operations classification needs, on the GPU through the C ABI: the constrained correlation of aligned sub-volumes with a
    class average at their fixed poses (ppm_sva_align with zero search ranges = one score sweep) and class averages (ppm_sva_insert +
    ppm_finalize).  tests/ drive the same driver with the CPU oracle behind this interface."""

    def __init__(self, device=0):
        self.device = device

    def scores(self, reference, cfg, volumes, wedges, poses):
        from . import host
        ref = host.Reference(reference, cfg.box / 2, device=self.device)
        try:
            return ref.sva_align(cfg, volumes, wedges, poses)[1]
        finally:
            ref.close()

    def average(self, cfg, chunks, members):
        """chunks: iterable of (lo, hi, volumes, wedges, poses, index); members: boolean (V,) - the sub-volumes that enter.
        -> (average, average of the even-index members, of the odd-index members, counts [even, odd]); None where nothing entered."""
        from . import host
        from .abi import FinalCfg
        acc = host.Accumulator(cfg.box, 1.0, "C1", device=self.device)
        try:
            for lo, hi, vols, wedges, poses, index in chunks:
                sel = np.where(members[lo:hi])[0]
                if len(sel):
                    acc.sva_insert(cfg, np.ascontiguousarray(vols[sel]), wedges[sel], poses[sel], index[sel])
            counts = acc.counts()
            if counts[0] + counts[1] == 0:
                return None, None, None, counts
            h_even, h_odd, avg, _ = acc.finalize(FinalCfg(molecular_mass_kda=0.0, inner_radius=0.0, outer_radius=0.0, mask_falloff=0.0))
            return avg, (h_even if counts[0] else None), (h_odd if counts[1] else None), counts
        finally:
            acc.close()


def score_cfg(cfg):
    """The alignment settings with the search switched off: ppm_sva_align then only scores the given poses."""
    c = SvaCfg.from_buffer_copy(bytes(cfg))
    c.tol_angle, c.tol_shift, c.search_mode, c.max_iterations = 0.0, 0.0, 0, 0
    return c


def classify(chunks, n_vol, cfg, references, backend, iterations=4, start=None):
    """Multi-reference classification of ALIGNED sub-volumes - the `class` / `refine` steps of a 3DAVG iteration (protocol modes 1 and 2,
    src/pyp/refine/tomo_avg/sub_tomo_avg.py:435-449: the class averages `<dataset>_iteration_%03d_level_%d_average_%03d.mrc` of one round
    are the references of the next, :79-94) - in the metric the alignment uses.  From K class references: every sub-volume is scored
    against every class at its own pose over the protocol's band-pass and its measured wedge (the score sweep of ppm_sva_align), each
    class's scores are standardised over the data set (the class with more members has the cleaner average and would attract
    everybody otherwise) and the sub-volume joins the class it fits relatively best; the classes are averaged (ppm_sva_insert +
    ppm_finalize) and the averages become the references of the next pass, until nothing moves or `iterations` passes are done.  A
    sub-volume is never compared with an average it is part of: the accumulators keep even- and odd-index members apart (the half-maps
    of ppm_finalize) and it is scored against the half of the OTHER parity; a class too small to have both halves keeps the reference
    it came with.  `chunks()` yields (lo, hi, volumes, wedges, poses, index) over the data set - the volumes stream from wherever they
    live on every pass.  `start` (instead of references): a first assignment (V,) of class numbers - the first references are the
    half-averages of those classes (classify_unsupervised draws them at random).
    -> (classes (V,), scores (V, K) of the last pass, class averages [K] (None for an empty class), passes run)."""
    if start is not None:
        classes = np.asarray(start, dtype=np.int64).copy()
        if len(classes) != n_vol or classes.min() < 0:
            raise ValueError("ERROR: classification: the first assignment must name a class for every sub-volume")
        K = int(classes.max()) + 1
        refs = [None] * K
    else:
        K = len(references or [])
        if K < 1:
            raise ValueError("ERROR: classification needs at least one class reference")
        refs = [(np.asarray(r, dtype=np.float32),) * 2 for r in references]          # (scored by odd-index members, by even-index members)
        classes = np.full(n_vol, -1, dtype=np.int64)
    sc_cfg = score_cfg(cfg)
    scores = np.zeros((n_vol, K))
    averages = [None] * K
    done = 0
    parity = np.zeros(n_vol, dtype=bool)

    def reaverage():
        for k in range(K):
            avg, h_even, h_odd, counts = backend.average(cfg, chunks(), classes == k)
            averages[k] = avg
            if avg is not None and min(counts) >= 2:
                refs[k] = (h_even, h_odd)                # odd-index members are scored against the even half and vice versa
            elif refs[k] is None:
                raise ValueError("ERROR: classification: class %d of the first assignment has fewer than two members of each index parity" % k)
    if start is not None:
        reaverage()
    for it in range(max(1, int(iterations))):
        for lo, hi, vols, wedges, poses, index in chunks():
            odd = (np.asarray(index) % 2) != 0
            parity[lo:hi] = odd
            for k in range(K):
                if refs[k][0] is refs[k][1]:
                    scores[lo:hi, k] = backend.scores(refs[k][0], sc_cfg, vols, wedges, poses)
                else:
                    s_odd, s_even = (backend.scores(h, sc_cfg, vols, wedges, poses) for h in refs[k])
                    scores[lo:hi, k] = np.where(odd, s_odd, s_even)
        z = np.zeros_like(scores)
        for grp in (parity, ~parity):                  # the two parities are scored against different half-averages: standardised apart
            if grp.any():
                sd = scores[grp].std(axis=0)
                z[grp] = (scores[grp] - scores[grp].mean(axis=0)) / np.where(sd > 0, sd, 1.0)
        new = np.argmax(z, axis=1) if K > 1 else np.zeros(n_vol, dtype=np.int64)
        done = it + 1
        moved = int((new != classes).sum())
        classes = new
        reaverage()
        if moved == 0:
            break
    return classes, scores, averages, done


def random_assignment(index, n_classes, rng):
    """A first assignment for classify(start=...): neighbouring (even, odd) table indices share a class, so that every class has both
    half-averages from the start; the pairs are dealt out over the classes in random order (equal sizes up to one pair)."""
    index = np.asarray(index, dtype=np.int64)
    pair = index // 2
    ids = np.unique(pair)
    deal = np.empty(len(ids), dtype=np.int64)
    deal[rng.permutation(len(ids))] = np.arange(len(ids)) % n_classes
    return deal[np.searchsorted(ids, pair)]


def classify_unsupervised(chunks, n_vol, cfg, n_classes, backend, restarts=6, iterations=12, seed=0):
    """Classification without class references (what MPI_Classification's protocol mode 1 does from the aligned sub-volumes alone; its
    method - hierarchical clustering of pairwise comparisons - is in the absent binary, this one is build-defined): `restarts` random
    first assignments (random_assignment) are each iterated by classify(); the partition whose members fit their own class best - the
    mean CROSS-VALIDATED score of a sub-volume against the half-average of its class that does not contain it - is kept.  The
    cross-validation is what makes a random start workable: without it every sub-volume correlates best with the average it is
    part of and nothing moves.  On the two-structure mixtures of the tests six of ten random starts reach the true partition, the
    others end in mixed classes with a visibly lower objective (CHANGELOG.md round 4).
    -> (classes, scores, averages, passes, objective, objectives of all restarts)."""
    index = np.concatenate([np.asarray(ix, dtype=np.int64) for _, _, _, _, _, ix in chunks()])
    if len(index) != n_vol:
        raise ValueError("ERROR: classification: the chunks do not cover the data set")
    n_classes = int(n_classes)
    if n_classes < 1 or 4 * n_classes > n_vol:
        raise ValueError("ERROR: classification: %d classes need at least %d sub-volumes (two of each index parity per class)" % (n_classes, 4 * max(n_classes, 1)))
    rng = np.random.default_rng(seed)
    best, objectives = None, []
    for _ in range(max(1, int(restarts))):
        try:
            classes, sc, avgs, its = classify(chunks, n_vol, cfg, None, backend, iterations=iterations, start=random_assignment(index, n_classes, rng))
        except ValueError:                       # a start whose classes lack a half (odd data sets): draw again
            objectives.append(float("-inf"))
            continue
        obj = float(sc[np.arange(n_vol), classes].mean())
        objectives.append(obj)
        if best is None or obj > best[4]:
            best = (classes, sc, avgs, its, obj)
    if best is None:
        raise ValueError("ERROR: classification: no random start gave every class two members of each index parity")
    return best + (objectives,)


def table_chunks(table, names, n, base_dir=".", chunk=64):
    """A chunks() provider over a volumes table: sub-volume files read `chunk` at a time, poses from the table's matrices."""
    def gen():
        for lo in range(0, len(table), chunk):
            hi = min(lo + chunk, len(table))
            vols = np.empty((hi - lo, n, n, n), dtype=np.float32)
            poses = np.zeros((hi - lo, 12))
            for k in range(lo, hi):
                fn = names[k] if os.path.isabs(names[k]) else os.path.join(base_dir, names[k])
                v = mrc.read(fn)
                if v.shape != (n, n, n):
                    raise ValueError(f"ERROR: {fn} is {v.shape}, expected {n}^3")
                vols[k - lo] = v
                N, p = line_to_pose(table[k, 9:12], table[k, 12:28])
                poses[k - lo, :9], poses[k - lo, 9:] = N.ravel(), p
            yield lo, hi, vols, table[lo:hi, 1:3].astype(np.float32), poses, table[lo:hi, 0].astype(np.int64)
    return gen
