"""Drop-in `refine3d`, `reconstruct3d`, `local_merge3d`, `merge3d`: same stdin answer scripts, same
input / output files, same log sentinels as the binaries PYP finds in external/cistem2/
(src/pyp/system/utils.py:227-235), computing on the GPU through libpypmatch.

Failure contract: print a line containing ERROR, exit non-zero, create no output file.
Sentinels the caller greps: "Reconstruct3D: Normal termination" (frealign.py:4970), the table between
"Rec_SSNR" and "Merge3D: Normal termination" (frealign.py:2557-2567).
"""
import os
import struct
import sys
import time

import numpy as np

from ..abi import FinalCfg, ReconCfg, RefineCfg
from ..formats import cistem, mrc, parfile
from . import prompts

C = cistem.COL
DUMP_MAGIC = b"PPMDUMP1"


class gpu_lock:
    """PYP may still start `slurm_tasks` concurrent refine3d / reconstruct3d processes per node
    (src/pyp/system/mpi.py:104); they are serialised per GPU with an advisory file lock so that one process at a
    time owns the device memory (SURVEY.md §8b 'Threading')."""

    def __init__(self, device):
        self.device = int(device)
        self.path = os.path.join(os.environ.get("PPM_LOCK_DIR", "/tmp"), "pyp_amd_gpu%d.lock" % int(device))
        self.fd = None

    def __enter__(self):
        import fcntl
        w = sys.modules.get("pyp_amd.surface.warm")      # the start-up thread takes the lock before it creates the GPU context
        if w is not None:
            self.fd = w.adopt_lock(self.device)
            if self.fd is not None:
                return self
        try:        # world-writable lock file: several users may share the node's GPUs
            old = os.umask(0)
            try:
                self.fd = os.open(self.path, os.O_CREAT | os.O_RDWR, 0o666)
            finally:
                os.umask(old)
        except OSError as e:
            _die(f"ERROR: cannot open the GPU lock file {self.path}: {e} (set PPM_LOCK_DIR to a writable directory)")
        fcntl.flock(self.fd, fcntl.LOCK_EX)
        return self

    def __exit__(self, *a):
        import fcntl
        fcntl.flock(self.fd, fcntl.LOCK_UN)
        os.close(self.fd)
        return False


def _die(msg):
    if "ERROR" not in msg:
        msg = "ERROR: " + msg
    print(msg, flush=True)
    _release_warm()
    sys.exit(1)


def _release_warm():
    w = sys.modules.get("pyp_amd.surface.warm")
    if w:
        w.release()


def _unsupported(d, keys, prog):
    for k, bad in keys:
        if d.get(k) == bad:
            _die(f"ERROR: {prog}: option '{k}' = {d[k]} is not supported by this build")


def _refuse_unless(d, prog, defaults):
    """Answers this build neither uses nor can honour: any value other than the one PYP always sends ends the run with ERROR
    (fail loudly, never silently).  defaults: (key, the accepted value, what the answer would ask for)."""
    for key, accepted, what in defaults:
        if abs(float(d[key]) - accepted) > 1e-9:
            _die(f"ERROR: {prog}: answer '{key}' = {d[key]:g} is not supported by this build ({what}); only {accepted:g} is accepted")


def fraction_mask(positions, fraction):
    """refine3d answer 14 "fraction of particles to use" (frealign.py:3934 always sends 1): the particles that get refined.  A row is
    used when a fixed hash of its POSITION_IN_STACK, mapped to [0, 1), lies below the fraction, so the choice does not depend on how
    PYP splits the particle ranges; the others are written out unchanged.  Build-defined (the absent program draws random numbers)."""
    x = np.asarray(positions, dtype=np.uint64) & np.uint64(0xFFFFFFFF)
    x = (x ^ (x >> np.uint64(16))) * np.uint64(0x7FEB352D) & np.uint64(0xFFFFFFFF)
    x = (x ^ (x >> np.uint64(15))) * np.uint64(0x846CA68B) & np.uint64(0xFFFFFFFF)
    x = x ^ (x >> np.uint64(16))
    return (x.astype(np.float64) / 4294967296.0) < float(fraction)


def _select(rows, first, last):
    pos = rows[:, C["POSITION_IN_STACK"]]
    sel = np.where((pos >= first) & (pos <= last))[0]
    if len(sel) == 0:
        _die(f"ERROR: no rows with POSITION_IN_STACK in {first}..{last}")
    return sel


def _open_stack(stack_path, positions):
    mm = mrc.mmap(stack_path)
    if positions.max() > mm.shape[0] or positions.min() < 1:
        _die(f"ERROR: {stack_path}: stack has {mm.shape[0]} images, rows ask for {int(positions.max())}")
    if mm.shape[1] != mm.shape[2]:
        _die(f"ERROR: {stack_path}: particle images must be square")
    return mm


class DeviceImages:
    """A chunk of particle images already in device memory (what host._images_arg accepts besides numpy / torch arrays)."""
    is_cuda, dtype, ready = True, "torch.float32", True       # ready: the upload has completed before the chunk is handed out

    def __init__(self, ptr, n, box):
        self.ptr, self.n, self.box = ptr, n, box

    def data_ptr(self):
        return self.ptr

    def is_contiguous(self):
        return True

    def numel(self):
        return self.n * self.box * self.box

    @property
    def device(self):
        return None


def _iter_image_chunks(mm, positions, device, chunk=None, group=1, call_mb=0, chunk_mb=256):
    """Yield (lo, hi, images) over the range, `images` = hi - lo particle images ALREADY ON THE DEVICE.  Three stages run
    concurrently: a reader thread fills page-locked buffers from the stack file (pread by several worker threads: one kernel copy
    out of the page cache, no page faults, the GIL released), an upload thread moves them into one of two device buffers on the
    library's upload stream, and the caller computes on the other device buffer.  The host never holds more than three chunks (a
    500 k x 256^2 range is 131 GB).  The page-locked buffers hold `chunk_mb` MB each (PPM_IO_CHUNK_MB overrides; page-locking costs
    ~0.27 s per GB): the executables' start-up thread (surface/warm.py) prepares them while the inputs are parsed, otherwise the
    reader allocates them as it goes.  `call_mb`: that many MB of uploaded chunks are handed out together, as one contiguous device
    array (the insertion kernels want ~8 k particles per call: a call on 2 048 costs 11 ms instead of 1.4); `group` says the same
    as a number of chunks.  Measured at 100 k x 256^2 (scripts/dropin_ab.py): refine3d (compute-bound) is fastest with 64 MB
    chunks, reconstruct3d (I/O-bound: 8 reader threads want large requests) with 256 MB."""
    import queue
    import threading
    from concurrent.futures import ThreadPoolExecutor
    from .. import host, lib
    n, box = len(positions), mm.shape[1]
    try:
        chunk_mb = max(1, int(os.environ.get("PPM_IO_CHUNK_MB", chunk_mb)))
    except ValueError:
        pass
    if chunk is None:       # page-locked buffers of PPM_IO_CHUNK_MB: pinning costs ~0.27 s per GB and sits in front of the first chunks
        chunk = int(os.environ.get("PPM_IO_CHUNK", str(max(16, (chunk_mb << 20) // (box * box * 4)))))
    chunk = max(1, min(chunk, n))
    nchunks = (n + chunk - 1) // chunk
    if call_mb:             # the device call's size in MB instead of a chunk count
        group = int(os.environ.get("PPM_IO_GROUP", str(max(1, (call_mb << 20) // max(1, chunk * box * box * 4)))))
    group = max(1, min(int(group), nchunks))
    idx = positions.astype(np.int64) - 1
    contiguous = bool(np.all(np.diff(idx) == 1))
    L = lib.load()
    lib.init(device)
    sec = box * box * 4
    npin, ndev = min(3, nchunks), min(2, (nchunks + group - 1) // group)
    warm = sys.modules.get("pyp_amd.surface.warm")
    if warm:
        warm.limit(npin)
    pinned, dev = [None] * npin, [None] * ndev
    nread = max(1, min(16, int(os.environ.get("PPM_IO_THREADS", "8"))))
    # pread straight from the file when the data block is plain little-endian float32 (what PYP writes); otherwise through the map
    fd = None
    if contiguous and mm.dtype == np.dtype("<f4") and getattr(mm, "filename", None):
        fd = os.open(mm.filename, os.O_RDONLY)
    pool = ThreadPoolExecutor(nread) if nread > 1 else None
    off0 = int(getattr(mm, "offset", 0))

    native = os.environ.get("PPM_IO_READER", "native") != "python"

    def fill(buf, lo, hi):
        if fd is not None and native:      # the library's reader pool: nread concurrent preads into the page-locked buffer, the GIL released
            if L.ppm_host_read(fd, off0 + int(idx[lo]) * sec, buf.ptr, (hi - lo) * sec, nread) != 0:
                raise IOError(lib.last_error())
            return
        dst = buf.array[:(hi - lo) * box * box].reshape(hi - lo, box, box)

        def part(a, e):
            if fd is None:
                dst[a - lo:e - lo] = mm[idx[a]:idx[a] + (e - a)] if contiguous else mm[idx[a:e]]
                return
            view = memoryview(dst[a - lo:e - lo]).cast("B")
            pos, want, done = off0 + int(idx[a]) * sec, (e - a) * sec, 0
            while done < want:
                got = os.preadv(fd, [view[done:min(want, done + (256 << 20))]], pos + done)
                if got <= 0:
                    raise IOError(f"ERROR: short read from the particle stack at byte {pos + done}")
                done += got
        cuts = np.linspace(lo, hi, (nread if hi - lo >= 4 * nread else 1) + 1).astype(int)
        if pool is None or len(cuts) == 2:
            part(lo, hi)
        else:
            list(pool.map(lambda ae: part(*ae), zip(cuts[:-1], cuts[1:])))

    stats = _iter_image_chunks.stats = {"read": 0.0, "upload": 0.0, "wait_pinned": 0.0, "wait_device": 0.0, "wait_data": 0.0, "compute": 0.0, "chunks": 0}
    filled, ready = queue.Queue(), queue.Queue()
    pin_free = [threading.Event() for _ in range(npin)]
    dev_free = [threading.Event() for _ in range(ndev)]
    for ev in pin_free + dev_free:
        ev.set()
    stop = threading.Event()

    def wait(ev):
        while not ev.wait(0.05):
            if stop.is_set():
                return False
        return not stop.is_set()

    def reader():
        try:
            for k, lo in enumerate(range(0, n, chunk)):
                hi, slot = min(lo + chunk, n), k % npin
                ta = time.time()
                if not wait(pin_free[slot]):
                    return
                pin_free[slot].clear()
                if pinned[slot] is None:        # one the start-up thread page-locked while the inputs were read, else our own
                    pinned[slot] = host.PinnedBuffer(chunk * box * box, device, ptr=warm.take(chunk * sec) if warm else None)
                tb = time.time()
                fill(pinned[slot], lo, hi)
                stats["wait_pinned"] += tb - ta; stats["read"] += time.time() - tb
                filled.put((lo, hi, slot, None))
            filled.put(None)
        except BaseException as e:          # handed on: a reader failure must end in the ERROR line, not a traceback
            filled.put((0, 0, 0, e))

    def uploader():
        try:
            k = 0
            while True:
                item = filled.get()
                if item is None:
                    return
                lo, hi, pslot, err = item
                if err is not None:
                    ready.put((0, 0, 0, err))
                    return
                dslot, part = (k // group) % ndev, k % group
                ta = time.time()
                if part == 0:
                    if not wait(dev_free[dslot]):
                        return
                    dev_free[dslot].clear()
                    glo = lo
                if dev[dslot] is None:
                    dev[dslot] = L.ppm_device_alloc(group * chunk * sec)
                    if not dev[dslot]:
                        raise lib.PpmError(lib.last_error())
                tb = time.time()
                if L.ppm_device_upload(dev[dslot] + part * chunk * sec, pinned[pslot].ptr, (hi - lo) * sec) != 0:
                    raise lib.PpmError(lib.last_error())
                pin_free[pslot].set()
                stats["wait_device"] += tb - ta; stats["upload"] += time.time() - tb
                if part == group - 1 or hi == n:
                    ready.put((glo, hi, dslot, None))
                k += 1
        except BaseException as e:
            ready.put((0, 0, 0, e))

    threads = [threading.Thread(target=reader, daemon=True), threading.Thread(target=uploader, daemon=True)]
    for t in threads:
        t.start()
    try:
        lo = 0
        while lo < n:
            ta = time.time()
            a, hi, slot, err = ready.get()
            if err is not None:
                raise ValueError(f"ERROR: reading the particle stack failed: {err}")
            tb = time.time()
            yield a, hi, DeviceImages(dev[slot], hi - a, box)
            dev_free[slot].set()
            stats["wait_data"] += tb - ta; stats["compute"] += time.time() - tb; stats["chunks"] += 1
            lo = hi
    finally:
        # the consumer may have raised inside the loop body: the threads must be done with the buffers before they are freed
        stop.set()
        filled.put(None)
        for t in threads:
            t.join()
        if pool is not None:
            pool.shutdown(wait=True)
        if fd is not None:
            os.close(fd)
        for pb in pinned:
            if pb is not None:
                pb.close()
        for q in dev:
            if q:
                L.ppm_device_free(q)


def _pipeline_stats():
    st = getattr(_iter_image_chunks, "stats", None)
    if not st:
        return "-"
    return ("%d chunks; reader: read %.2f s, waited for a buffer %.2f s; uploader: copied %.2f s, waited for a buffer %.2f s; main thread: computed %.2f s, "
            "waited for data %.2f s" % (st["chunks"], st["read"], st["wait_pinned"], st["upload"], st["wait_device"], st["compute"], st["wait_data"]))


def _ssnr_ring_weights(n, stats_path, pixel):
    """'use statistics' = yes: figure-of-merit weighting of the reference rings, w = sqrt(2 pFSC / (1 + pFSC)) from the
    part-FSC column of statistics_rNN.txt (7 columns, src/pyp/postprocess/core.py:203-221), tabulated per Fourier pixel
    0 .. n/2 for ppm_reference_create_weighted (applied on the device while the reference cube is cut out)."""
    st = np.loadtxt(stats_path, comments=["C"], ndmin=2)
    if st.shape[1] < 5 or len(st) < 2:
        _die(f"ERROR: {stats_path}: statistics file needs 7 columns")
    res, pfsc = st[:, 1], np.clip(st[:, 4], 0.0, 1.0)
    w_tab = np.sqrt(2 * pfsc / (1 + pfsc))
    s = np.arange(n // 2 + 1) / (n * pixel)
    order = np.argsort(1.0 / res)
    return np.interp(s, (1.0 / res)[order], w_tab[order], left=1.0, right=float(w_tab[order][-1])).astype(np.float32)


# ------------------------------------------------------------------------------------------ refine3d
def refine_cfg_from_answers(d, box):
    """ppm_refine_cfg from the parsed refine3d answers (frealign.py:3918-3994; answer numbers in include/ppm.h).
    Focus mask (answers 29-32, used when answer 44 "apply 2D masking" is yes, frealign.py:3846-3849): the caller's X, Y, Z are
    Angstrom from the corner of the reference box (what a map viewer displays); the library takes them from the box centre."""
    focus = None
    if d.get("mask_2d"):
        half = 0.5 * box * d["pixel_size"]
        focus = (d["focus_x"] - half, d["focus_y"] - half, d["focus_z"] - half, d["focus_r"])
    return RefineCfg.make(focus=focus,
        box=box, pixel_size=d["pixel_size"], molecular_mass_kda=d["molecular_mass"], mask_radius=d["outer_radius"], res_low=d["res_low"],
        res_high=d["res_high"], res_signed_cc=d["res_signed_cc"], search_mask_radius=d["search_mask_radius"],
        res_search=d["res_search"], angular_step=d["angular_step"], top_hits=d["top_hits"], search_range_x=d["search_range_x"],
        search_range_y=d["search_range_y"], global_search=int(d["global_search"]), local_refine=int(d["local_refine"]),
        refine_psi=int(d["refine_psi"]), refine_theta=int(d["refine_theta"]), refine_phi=int(d["refine_phi"]),
        refine_x=int(d["refine_x"]), refine_y=int(d["refine_y"]), normalize=int(d["normalize"]), invert=int(d["invert"]),
        symmetry=d["symmetry"][:7], refine_defocus=int(d["refine_defocus"]), defocus_range=d["defocus_range"],
        defocus_step=d["defocus_step"], res_classification=d["res_classification"])


def refine3d_main(argv=None, stdin=None):
    t0 = time.time()
    os.environ.setdefault("PPM_SYNC", "block")      # a spinning device wait starves the reader threads (measured: 29 k -> 53 k particles/s)
    try:
        d = prompts.parse_refine3d(prompts.read_answers(stdin or sys.stdin))
    except prompts.PromptError as e:
        _die(str(e))
    print("\n        **   Welcome to Refine3D (MI355X / libpypmatch)   **\n")
    for k, v in d.items():
        print(f"{k:28s}: {v}")
    _unsupported(d, [("exclude_edges", True), ("normalize_reference", True), ("threshold_reference", True)], "refine3d")
    _refuse_unless(d, "refine3d", [("inner_radius", 0.0, "an inner mask radius")])
    if not 0.0 < d["fraction"] <= 1.0:
        _die(f"ERROR: refine3d: fraction of particles to use must be in (0, 1], got {d['fraction']:g}")
    if d["res_classification"] < 0:
        _die(f"ERROR: refine3d: classification resolution limit must be >= 0, got {d['res_classification']:g}")
    pad = int(round(d["padding"]))
    if abs(d["padding"] - pad) > 1e-6 or pad not in (1, 2, 4):
        _die("ERROR: refine3d: padding factor must be 1, 2 or 4")
    if d["mask_2d"] and not d["focus_r"] > 0:
        _die("ERROR: refine3d: 2D masking asked for with a focus mask of radius 0")
    for p in (d["stack"], d["input_params"], d["reference"]):
        if not os.path.exists(p):
            _die(f"ERROR: refine3d: input file {p} does not exist")
    px = d["pixel_size"]
    if d["surface"] == "par":
        par, version, ext, _, _ = parfile.read(d["input_params"])
        if version not in (parfile.NEW, parfile.FREALIGNX):
            _die("ERROR: refine3d: only NEW / FREALIGNX parameter files are supported")
        rows = parfile.par_to_cistem(par, version, px, d["voltage"], d["cs"], d["amplitude_contrast"])
    else:
        rows = cistem.read_parameters(d["input_params"])
    sel = _select(rows, d["first"], d["last"])
    rall = rows[sel]                        # every row of the range is written out; `use` marks the ones that are refined (answer 14)
    use = fraction_mask(rall[:, C["POSITION_IN_STACK"]], d["fraction"]) if d["fraction"] < 1.0 else np.ones(len(rall), dtype=bool)
    if not use.any():
        _die(f"ERROR: refine3d: fraction {d['fraction']:g} leaves no particle of {d['first']}..{d['last']} to refine")
    rin = rall[use]
    if d["fraction"] < 1.0:
        print(f"fraction of particles to use = {d['fraction']:g}: {len(rin)} of {len(rall)} rows are refined, the others are copied")
    mm = _open_stack(d["stack"], rin[:, C["POSITION_IN_STACK"]])
    box = mm.shape[1]
    vol = mrc.read(d["reference"]).astype(np.float32)
    if vol.shape != (box, box, box):
        _die(f"ERROR: refine3d: reference is {vol.shape}, particles are {box}^2")
    ring_w = None
    if d["use_statistics"]:
        if not os.path.exists(d["statistics"]):
            _die(f"ERROR: refine3d: statistics file {d['statistics']} does not exist")
        ring_w = _ssnr_ring_weights(box, d["statistics"], px)
    cfg = refine_cfg_from_answers(d, box)
    if d["use_priors"]:
        # answer 7 with the statistics of answer 3 (`<name>_stat.cistem`: means and variances of every column over the data set,
        # src/pyp_main.py:2667-2674): a Gaussian restraint on the refined parameters (include/ppm.h, ppm_refine_cfg.use_priors)
        if d["global_stats"] in ("null", "") or not os.path.exists(d["global_stats"]):
            _die(f"ERROR: refine3d: use priors = yes needs the global statistics file (answer 3), got '{d['global_stats']}'")
        st = cistem.read_parameters(d["global_stats"])
        if st.shape[0] < 2:
            _die(f"ERROR: refine3d: {d['global_stats']} must hold two rows (means, variances)")
        cols = [C["PSI"], C["THETA"], C["PHI"], C["X_SHIFT"], C["Y_SHIFT"]]
        cfg.use_priors = 1
        cfg.prior_mean[:] = [float(v) for v in st[0, cols]]
        cfg.prior_var[:] = [float(v) for v in st[1, cols]]
        print("priors: mean psi theta phi x y = %s, variance = %s" % (np.round(st[0, cols], 3).tolist(), np.round(st[1, cols], 3).tolist()))
    from .. import host, lib
    dev = int(os.environ.get("PPM_DEVICE", "0"))
    t1 = time.time()
    try:
        with gpu_lock(dev):
            if box * pad > 512:
                _die(f"ERROR: refine3d: padding factor {pad} needs a padded box of {box * pad} > 512")
            # the reference is prepared (upload, 3-D transform, slice bank) by a thread of its own while the pipeline below reads and
            # uploads the first images: different handles and streams (include/ppm.h)
            import threading
            made = {}

            def make_reference():
                try:
                    made["ref"] = host.Reference(vol, box / 2, device=dev, pad=pad, ring_weight=ring_w)
                except BaseException as e:         # noqa: BLE001 - reported by the main thread
                    made["err"] = e
            maker = threading.Thread(target=make_reference)
            maker.start()
            rout = np.empty_like(rin)
            chunks = _iter_image_chunks(mm, rin[:, C["POSITION_IN_STACK"]], dev, call_mb=512, chunk_mb=64)
            try:
                first = next(chunks)               # starts the reader / uploader threads and waits for the first images
            finally:
                maker.join()
            if "err" in made:
                chunks.close()
                raise made["err"]
            ref = made["ref"]
            t2 = time.time()
            lo, hi, imgs = first
            rout[lo:hi] = ref.refine(cfg, imgs, rin[lo:hi])
            for lo, hi, imgs in chunks:
                rout[lo:hi] = ref.refine(cfg, imgs, rin[lo:hi])
            t3 = time.time()
            note = ref.note()
            match_tmp = None
            if d["calc_match"]:         # answers 8 / 43: the model of every particle of the range at its refined pose, one section each
                # streamed chunk by chunk into a pre-sized file under a temporary name (a 500 k x 256^2 range is 131 GB); the name the
                # caller looks for appears only after the parameter outputs are written
                step = max(1, (256 << 20) // (box * box * 4))
                match_tmp = d["match_out"] + ".tmp%d" % os.getpid()
                out_mm = mrc.create(match_tmp, (len(rout), box, box), pixel_size=px)
                amin, amax, asum, asq = np.inf, -np.inf, 0.0, 0.0
                for lo in range(0, len(rout), step):
                    m = ref.match_projections(cfg, rout[lo:lo + step])
                    out_mm[lo:lo + len(m)] = m
                    amin, amax = min(amin, float(m.min())), max(amax, float(m.max()))
                    asum += float(m.sum(dtype=np.float64)); asq += float((m.astype(np.float64) ** 2).sum())
                out_mm.flush()
                del out_mm
                cnt = float(len(rout)) * box * box
                mrc.set_statistics(match_tmp, amin, amax, asum / cnt, float(np.sqrt(max(0.0, asq / cnt - (asum / cnt) ** 2))))
            ref.close()
    except (lib.PpmError, ValueError) as e:
        if "match_tmp" in locals() and match_tmp and os.path.exists(match_tmp):
            os.remove(match_tmp)
        _die(str(e))
    if note:
        print("\n" + note)
    if not use.all():                       # the rows that were not drawn keep their input values
        full = rall.copy()
        full[use] = rout
        rout, rin = full, rall
    changes = rout - rin
    changes[:, C["POSITION_IN_STACK"]] = rin[:, C["POSITION_IN_STACK"]]
    if d["surface"] == "par":
        # MAG and, for an extended file, the 29 trailing columns are not touched by the refinement: carried over from the input
        nstd = 17 if version == parfile.FREALIGNX else 16
        pout = parfile.cistem_to_par(rout, version, mag=par[sel, 6], change=changes[:, C["SCORE"]])
        if ext:
            pout = np.hstack([pout, par[sel, nstd:]])
        parfile.write(d["output_params"], pout, version=version, extended=ext)
    else:
        cistem.write_parameters(d["output_params"], rout)
        if d["output_changes"] not in ("/dev/null", "null"):
            cistem.write_parameters(d["output_changes"], changes)
    if match_tmp:
        os.replace(match_tmp, d["match_out"])
    print("\n   NO     PSI   THETA     PHI       SHX       SHY     SCORE   CHANGE")
    for r, c in zip(rout[:50], changes[:50]):
        print("%7d%8.2f%8.2f%8.2f%10.2f%10.2f%10.4f%9.4f" % (r[0], r[1], r[2], r[3], r[4], r[5], r[C["SCORE"]], c[C["SCORE"]]))
    print(f"\nRefined {len(rout)} particles in {time.time() - t0:.1f} s; mean score {rout[:, C['SCORE']].mean():.4f}")
    print(f"Timing: inputs {t1 - t0:.2f} s, device + reference {t2 - t1:.2f} s, particles {t3 - t2:.2f} s, outputs {time.time() - t3:.2f} s")
    print("Pipeline: " + _pipeline_stats())
    _release_warm()
    print("\nRefine3D: Normal termination\n", flush=True)
    return 0


# ------------------------------------------------------------------------------------------ dumps
def write_dump(path, box, pixel, count, data):
    """One accumulator half under a temporary name, renamed when complete.  The data block (201 MB at 256^3) is written by four
    threads at their own offsets, straight from the array (no intermediate bytes object)."""
    from concurrent.futures import ThreadPoolExecutor
    tmp = path + ".tmp%d" % os.getpid()
    data = np.ascontiguousarray(data, dtype="<f4")
    view = memoryview(data).cast("B")
    head = DUMP_MAGIC + struct.pack("<ifq", box, pixel, count)
    fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o666)
    try:
        os.pwrite(fd, head, 0)
        os.ftruncate(fd, len(head) + len(view))

        def part(a, e):
            while a < e:
                a += os.pwrite(fd, view[a:min(e, a + (64 << 20))], len(head) + a)
        cuts = [len(view) * k // 4 for k in range(5)]
        with ThreadPoolExecutor(4) as ex:
            list(ex.map(lambda ae: part(*ae), zip(cuts[:-1], cuts[1:])))
    except BaseException:
        os.close(fd)
        fd = None
        if os.path.exists(tmp):
            os.remove(tmp)
        raise
    finally:
        if fd is not None:
            os.close(fd)
    os.replace(tmp, path)


def read_dump(path):
    with open(path, "rb") as f:
        head = f.read(24)
        if len(head) < 24 or head[:8] != DUMP_MAGIC:
            _die(f"ERROR: {path} is not a libpypmatch dump file")
        box, pixel, count = struct.unpack("<ifq", head[8:])
        n = box * box * (box // 2 + 1) * 3
        data = np.fromfile(f, dtype="<f4", count=n)
    if data.size != n:
        _die(f"ERROR: {path}: dump file is truncated")
    return box, pixel, count, data


# ------------------------------------------------------------------------------------------ reconstruct3d
def apply_tilt_window(rows, lo, hi):
    """reconstruct3d answers 20 / 21 "min / max tilt-particle score": PYP's script sends 0 and -1 ("we just use occ as limit") and shows,
    commented out next to them, what the pair carries: csp_UseImagesForRefinementMin / Max, the window of tilt-image indices that enter
    (src/pyp/refine/frealign/frealign.py:1763-1766).  Rows whose TIND lies below `lo`, or above `hi` when hi >= 0, get OCCUPANCY 0
    here (in place) - the same window ppm_csp_cfg.tind_min / tind_max applies to the refinement.  Returns the number switched off."""
    t = rows[:, C["TIND"]]
    off = (t < lo) | ((t > hi) if hi >= 0 else np.zeros(len(rows), dtype=bool))
    off &= rows[:, C["OCCUPANCY"]] > 0
    rows[off, C["OCCUPANCY"]] = 0.0
    return int(off.sum())


def reconstruct3d_main(argv=None, stdin=None):
    t0 = time.time()
    os.environ.setdefault("PPM_SYNC", "block")
    try:
        d = prompts.parse_reconstruct3d(prompts.read_answers(stdin or sys.stdin))
    except prompts.PromptError as e:
        _die(str(e))
    print("\n        **   Welcome to Reconstruct3D (MI355X / libpypmatch)   **\n")
    for k, v in d.items():
        print(f"{k:28s}: {v}")
    _unsupported(d, [("center_mass", True),
                     ("threshold_reference", True), ("exclude_edges", True), ("split_even_odd", False), ("dump", False)], "reconstruct3d")
    if abs(d["padding"] - 1.0) > 1e-6:
        _die("ERROR: reconstruct3d: only padding factor 1 is supported")
    _refuse_unless(d, "reconstruct3d", [("inner_radius", 0.0, "an inner mask radius"), ("res_reference", 0.0, "a resolution limit for the input reconstruction"),
                                        ("smoothing", 1.0, "a smoothing factor")])
    if not d["input_params"].endswith(".cistem"):
        _die("ERROR: reconstruct3d: input parameters must be a .cistem file")
    if d["crop"]:
        # `refine_crop` (frealign.py:1717-1720) lets the CPU program transform cropped images to save time; the GPU path always
        # transforms the full box, so the answer changes nothing here - said in the log rather than refused
        print("NOTE: crop = yes has no effect: the full box is transformed")
    if d["likelihood_blurring"] and not os.path.exists(d["reference"]):
        _die(f"ERROR: reconstruct3d: likelihood blurring needs the reference {d['reference']}")
    for p in (d["stack"], d["input_params"]):
        if not os.path.exists(p):
            _die(f"ERROR: reconstruct3d: input file {p} does not exist")
    rows = cistem.read_parameters(d["input_params"])
    sel = _select(rows, d["first"], d["last"])
    rin = rows[sel].copy()
    px = d["pixel_size"]
    n_off = apply_tilt_window(rin, d["min_tilt_score"], d["max_tilt_score"])
    if n_off:
        print(f"tilt window {d['min_tilt_score']:g}..{d['max_tilt_score']:g}: {n_off} rows left out")
    used = rin[:, C["OCCUPANCY"]] > 0
    if d["adjust_scores"] and used.sum() > 10:
        # score vs defocus regression, removed before thresholding / weighting
        df = 0.5 * (rin[used, C["DEFOCUS_1"]] + rin[used, C["DEFOCUS_2"]])
        sc = rin[used, C["SCORE"]]
        if df.std() > 0:
            slope = np.polyfit(df, sc, 1)[0]
            rin[:, C["SCORE"]] -= slope * (0.5 * (rin[:, C["DEFOCUS_1"]] + rin[:, C["DEFOCUS_2"]]) - df.mean())
    if d["global_stats"] not in ("null", "") and os.path.exists(d["global_stats"]):
        score_avg = float(cistem.read_parameters(d["global_stats"])[0, C["SCORE"]])
    else:
        score_avg = float(rin[used, C["SCORE"]].mean()) if used.any() else 0.0
    mm = _open_stack(d["stack"], rin[:, C["POSITION_IN_STACK"]])
    box = mm.shape[1]
    rc = ReconCfg(box=box, pixel_size=px, res_limit=d["res_limit"], score_weight_bfactor=d["score_bfactor"] if d["score_weighting"] else 0.0,
                  score_average=score_avg, score_threshold=d["score_threshold"], normalize=int(d["normalize"]), invert=int(d["invert"]),
                  split_by_pind=int(d["per_particle_splitting"]), mask_radius=d["outer_radius"])
    if d["dose_weighting"]:
        # the five-line answer (frealign.py:1731-1753): external per-exposure weights or the ones this parameter file gives
        # (src/pyp/inout/metadata/core.py:3039-3075); side files weights.txt / scores.txt for the caller's plots
        from .. import dose
        wf = d["dose_weights_file"]
        gw = dose.read_global_weights(wf) if (wf != "/scratch/not_provided" and os.path.exists(wf)) else dose.compute_global_weights(rows)
        frames = max(1, len(np.unique(rows[:, C["FIND"]]))) if d["dose_multiply"] else 1
        q = dose.normalised(gw)
        rc.set_dose_weights(q, d["dose_fraction"] * frames, d["dose_transition"])
        if d["first"] == 1:      # PYP runs many ranges in one directory: the side files depend on the whole table only, the first range writes them
            dose.write_weights_txt("weights.txt", q, box, d["dose_fraction"] * frames, d["dose_transition"])
            dose.write_scores_txt("scores.txt", gw)
        print(f"dose weighting: {int((q > 0).sum())} exposures, exponent {d['dose_fraction'] * frames:g}, transition {d['dose_transition']:g}")
    from .. import host, lib
    dev = int(os.environ.get("PPM_DEVICE", "0"))
    t1 = time.time()
    try:
        with gpu_lock(dev):
            acc = host.Accumulator(box, px, d["symmetry"], device=dev)
            t2 = time.time()
            blur_ref = None
            if d["likelihood_blurring"]:
                vol = mrc.read(d["reference"]).astype(np.float32)
                if vol.shape != (box, box, box):
                    _die(f"ERROR: reconstruct3d: reference is {vol.shape}, particles are {box}^2")
                blur_ref = host.Reference(vol, box / 2, device=dev)
            for lo, hi, imgs in _iter_image_chunks(mm, rin[:, C["POSITION_IN_STACK"]], dev, group=1, call_mb=0 if d["likelihood_blurring"] else 2048):
                if blur_ref is None:
                    acc.insert(rc, imgs, rin[lo:hi])
                else:
                    n_blur = blurred_insert(acc, rc, blur_ref, imgs, rin[lo:hi], d)
                    print(f"likelihood blurring: rows {lo + 1}..{hi}: {n_blur:.2f} orientations inserted per particle on average")
            if blur_ref is not None:
                blur_ref.close()
                ok = (rin[:, C["OCCUPANCY"]] > 0) & ~(rin[:, C["SCORE"]] < d["score_threshold"])
                key = rin[:, C["PIND"] if d["per_particle_splitting"] else C["POSITION_IN_STACK"]].astype(np.int64) % 2
                acc.set_counts(int((ok & (key == 0)).sum()), int((ok & (key == 1)).sum()))      # particles, not inserted copies
            t3 = time.time()
            data = acc.download()
            counts = acc.counts()
            acc.close()
    except (lib.PpmError, ValueError) as e:
        _die(str(e))
    half = data.size // 2
    import threading
    failed = []

    def dump(path, count, part):          # a full disk must end in the ERROR line, from either thread
        try:
            write_dump(path, box, px, count, part)
        except OSError as e:
            failed.append(f"ERROR: reconstruct3d: could not write {path}: {e}")
    w2 = threading.Thread(target=dump, args=(d["dump_2"], counts[0], data[:half]))      # even keys -> map 2
    w2.start()
    dump(d["dump_1"], counts[1], data[half:])                                           # odd keys  -> map 1
    w2.join()
    if failed:
        for pth in (d["dump_1"], d["dump_2"]):
            if os.path.exists(pth):
                os.remove(pth)
        _die(failed[0])
    with open(d["res_file"], "w") as f:
        f.write("C Reconstruct3D (libpypmatch): particles %d..%d, inserted %d + %d\n" % (d["first"], d["last"], counts[1], counts[0]))
    print(f"\nInserted {counts[0] + counts[1]} of {len(rin)} particles in {time.time() - t0:.1f} s")
    print(f"Timing: inputs {t1 - t0:.2f} s, device {t2 - t1:.2f} s, particles {t3 - t2:.2f} s, dumps {time.time() - t3:.2f} s")
    print("Pipeline: " + _pipeline_stats())
    print("NOTE: the dump files are in libpypmatch's own format (PPMDUMP1): only this build's local_merge3d / merge3d read them "
          "(frealign.py:1852 consumers must be replaced together, INTEGRATION.md 1)")
    print("\nNormal termination, intermediate files dumped")
    _release_warm()
    print("\nReconstruct3D: Normal termination\n", flush=True)
    return 0


BLUR_NROT, BLUR_START, BLUR_STEP, BLUR_RANGE = 21, -10.0, 1.0, 20.0      # reconstruct_lblur_* defaults (config/pyp_config.toml:5995-6025)


def blurred_insert(acc, rc, ref, imgs, rows, d, nrot=BLUR_NROT, start=BLUR_START, step=BLUR_STEP, logp_range=BLUR_RANGE):
    """Likelihood blurring (answer "likelihood blurring" = yes, frealign.py:1772, :1817): every particle is inserted at `nrot`
    in-plane rotations psi + start + k step, each weighted by its likelihood against the reference, exp(LOGP_k - LOGP_max)
    normalised to sum 1 over the rotations whose LOGP lies within `logp_range` of the best (the others are dropped).  The
    script carries only yes / no: the grid is PYP's reconstruct_lblur_* defaults.  Build-defined (the absent program's rule
    is not visible).  Returns the mean number of rotations inserted per particle."""
    cfg = RefineCfg.make(box=int(rc.box), pixel_size=float(rc.pixel_size), mask_radius=float(rc.mask_radius), res_high=float(rc.res_limit),
                         global_search=0, local_refine=0, normalize=int(rc.normalize), invert=int(rc.invert))
    logp = np.empty((len(rows), nrot))
    for k in range(nrot):
        rk = rows.copy()
        rk[:, C["PSI"]] = np.mod(rk[:, C["PSI"]] + start + k * step, 360.0)
        logp[:, k] = ref.refine(cfg, imgs, rk)[:, C["LOGP"]]
    rel = logp - logp.max(axis=1, keepdims=True)
    w = np.where(rel >= -logp_range, np.exp(rel), 0.0)
    w /= w.sum(axis=1, keepdims=True)
    for k in range(nrot):
        if not np.any(w[:, k] > 1e-3):
            continue
        rk = rows.copy()
        rk[:, C["PSI"]] = np.mod(rk[:, C["PSI"]] + start + k * step, 360.0)
        rk[:, C["OCCUPANCY"]] = np.where(w[:, k] > 1e-3, rows[:, C["OCCUPANCY"]] * w[:, k], 0.0)
        acc.insert(rc, imgs, rk)
    return float((w > 1e-3).sum(axis=1).mean())


def _sum_dumps(seed1, seed2, n):
    tot = None
    box = pixel = None
    counts = [0, 0]
    for k in range(1, n + 1):
        for h, seed in ((0, seed1), (1, seed2)):
            p = prompts.dump_name(seed, k)
            if not os.path.exists(p):
                _die(f"ERROR: dump file {p} does not exist")
            b, px, cnt, data = read_dump(p)
            if box is None:
                box, pixel = b, px
                tot = [np.zeros_like(data, dtype=np.float64), np.zeros_like(data, dtype=np.float64)]
            if b != box:
                _die(f"ERROR: dump file {p} has box {b}, expected {box}")
            tot[h] += data
            counts[h] += cnt
    return box, pixel, counts, tot


def local_merge3d_main(argv=None, stdin=None):
    try:
        d = prompts.parse_local_merge3d(prompts.read_answers(stdin or sys.stdin))
    except prompts.PromptError as e:
        _die(str(e))
    print("\n        **   Welcome to LocalMerge3D (libpypmatch)   **\n")
    box, pixel, counts, tot = _sum_dumps(d["dump_seed_1"], d["dump_seed_2"], d["n_dumps"])
    write_dump(d["out_dump_1"], box, pixel, counts[0], tot[0].astype(np.float32))
    write_dump(d["out_dump_2"], box, pixel, counts[1], tot[1].astype(np.float32))
    print(f"Merged {d['n_dumps']} dump pairs ({counts[0]} + {counts[1]} particles)")
    print("\nLocalMerge3D: Normal termination\n", flush=True)
    return 0


def format_stats_table(stats):
    """Rows in the fixed widths the caller slices with numpy.genfromtxt(delimiter=[5,8,10,10,10,10,10])
    (frealign.py:2559): shell, resolution, ring radius, FSC, part-FSC, part-SSNR, rec-SSNR."""
    lines = []
    for s in stats:
        lines.append("%5d%8.2f%10.4f%10.4f%10.4f%10.4f%10.2f" % (int(s[0]), s[1], s[2], s[3], s[4], min(s[5], 99999.0), min(s[6], 999999.0)))
    return lines


def format_statistics_rows(stats):
    """Rows of <name>_statistics.txt: 7 x %14.5f, what numpy.savetxt writes when the reference rewrites the file
    (src/pyp/postprocess/core.py:219-221; pinned by tests/golden/golden_r02.json)."""
    return "".join("%14.5f%14.5f%14.5f%14.5f%14.5f%14.5f%14.5f\n" % tuple(s) for s in stats)


def merge3d_main(argv=None, stdin=None):
    t0 = time.time()
    try:
        d = prompts.parse_merge3d(prompts.read_answers(stdin or sys.stdin))
    except prompts.PromptError as e:
        _die(str(e))
    print("\n        **   Welcome to Merge3D (MI355X / libpypmatch)   **\n")
    for k, v in d.items():
        print(f"{k:28s}: {v}")
    box, pixel, counts, tot = _sum_dumps(d["dump_seed_1"], d["dump_seed_2"], d["n_dumps"])
    from .. import host, lib
    dev = int(os.environ.get("PPM_DEVICE", "0"))
    try:
        with gpu_lock(dev):
            acc = host.Accumulator(box, pixel, "C1", device=dev)
            acc.add(np.concatenate([tot[1], tot[0]]).astype(np.float32))      # dump 1 = odd keys = half index 1
            acc.set_counts(counts[1], counts[0])
            fc = FinalCfg(molecular_mass_kda=d["molecular_mass"], inner_radius=d["inner_radius"], outer_radius=d["outer_radius"], mask_falloff=0.0)
            h_even, h_odd, filt, stats = acc.finalize(fc)
            acc.close()
    except (lib.PpmError, ValueError) as e:
        _die(str(e))
    mrc.write(h_odd, d["half1"], pixel_size=pixel)
    mrc.write(h_even, d["half2"], pixel_size=pixel)
    mrc.write(filt, d["filtered"], pixel_size=pixel)
    with open(d["statistics"], "w") as f:
        f.write("C  NO.   RESOL  RING RAD       FSC  Part_FSC Part_SSNR  Rec_SSNR\n")
        f.write(format_statistics_rows(stats))
    print(f"\nParticles: {counts[0]} (map 1) + {counts[1]} (map 2); finalised in {time.time() - t0:.1f} s\n")
    print("   NO.   RESOL  RING RAD       FSC  Part_FSC Part_SSNR  Rec_SSNR")
    for line in format_stats_table(stats):
        print(line)
    print("\n\nMerge3D: Normal termination\n", flush=True)
    return 0
