"""Drop-in `csp`: constrained refinement of tilt-series particles and the extraction of their projections, behind the argv
PYP builds in create_csp_split_commands (src/pyp/system/local_run.py:306-467):

    csp <param.cistem> <param_extended.cistem> <mode> <first> <last> <flag> <images> <stack> > <log>

The program reads its settings from `.pyp_config.toml` in the working directory, as the reference's does
(src/pyp/align/core.py:1055: project_params.save_parameters right before the fan-out; CamelCase `csp_*` keys,
config/pyp_config.toml:6241-6641) and the current map from $PYP_SCRATCH/<data_set>_frames_CSP_01.mrc (align/core.py:916-931).

Modes (src/pyp/align/core.py:1015-1023 and the 2 -> 5, 3 -> 6 remaps of local_run.py:332-335, :413, :428):
   -2  extract the projections of particles first..last from the tilt series (<images> = .mrc) or from the movies of a
       frame list (<images> = frames_csp.txt, one movie per tilt: section FIND of movie IMIND) into <stack>
   -2.1  the same with every box replaced by the running average over the neighbouring frames of its particle
       (csp_produce_running_average, align/core.py:1000-1001; merged by the caller into <name>_stack_weighted_average.mrc)
    1 / 2 / 5  particle rotations / 3-D shifts / both, particles first..last (PIND); 8 / 7 = 1 / 2 (the caller's region codes)
    0 / 3 / 6  tilt angle + axis / image shifts / both, tilts first..last (TIND; last = -1: up to the end)
    4  one defocus offset per tilt (csp_ToleranceMicrographDefocus1 either side, 50 A steps), tilts first..last
Region-based ("patch") refinement (align/core.py:1106-1151, local_run.py:327-404): <param> is one of the
`<name>_regionNNNN.cistem` files of particle_cspt.split_parameter_file — the rows of the region's particles with RIND = the
region's index, the tilts of its extended file re-keyed (TIND, RIND) — and every job refines ONE particle (mode 5, first =
last = PIND) or ONE tilt of the region (modes 6 / 3 / 4, first = last = TIND), scored on the region's rows only.
Outputs of a refinement mode: <param>_<first:06d>_<last:06d>.cistem with the rows of the refined units only and its
_extended twin holding only the refined units' block entries (the caller merges them over the original,
src/pyp/refine/csp/particle_cspt.py:96-138 -> cistem_star_file.py:655-692).
Failure contract as for the other executables: a line containing ERROR, non-zero exit, no output file.
"""
import ast
import os
import sys
import time

import numpy as np

from ..abi import CSP_MICROGRAPHS, CSP_PARTICLES, CspCfg, RefineCfg
from ..formats import cistem, mrc
from .cli import _die, gpu_lock

C = cistem.COL


def read_flat_toml(path):
    """`key = value` lines as project_params.save_parameters writes them (a flat table; strings, numbers, booleans, lists)."""
    out = {}
    with open(path) as f:
        for raw in f:
            line = raw.strip()
            if not line or line.startswith("#") or line.startswith("["):
                continue
            k, sep, v = line.partition("=")
            if not sep:
                continue
            v = v.strip()
            try:
                if v in ("true", "false"):
                    val = v == "true"
                else:
                    val = ast.literal_eval(v)
            except (ValueError, SyntaxError):
                val = v.strip('"')
            out[k.strip().strip('"')] = val
    return out


def schedule(value, iteration):
    """Colon schedules "8:7:6" resolved like project_params.param (src/pyp/system/project_params.py:362-373)."""
    if isinstance(value, str) and ":" in value:
        parts = value.split(":")
        return float(parts[min(max(iteration - 2, 0), len(parts) - 1)])
    return float(value)


def _settings(p):
    it = int(p.get("refine_iter", 2))
    pixel = float(p["scope_pixel"]) * float(p.get("data_bin", 1)) * float(p.get("extract_bin", 1))
    return dict(
        iteration=it, pixel=pixel, box=int(p["extract_box"]), extract_bin=int(p.get("extract_bin", 1)),
        radius=float(schedule(p["particle_rad"], it)), mw=float(p.get("particle_mw", 0.0) or 0.0),
        res_low=schedule(p.get("refine_rlref", 0.0), it), res_high=schedule(p["refine_rhref"], it),
        res_signed=30.0 if not p.get("refine_fboost") else schedule(p.get("refine_fboost_lim", 30.0), it),
        tind_min=int(schedule(p.get("csp_UseImagesForRefinementMin", 0), it)), tind_max=int(schedule(p.get("csp_UseImagesForRefinementMax", -1), it)),
        step_tol=float(p.get("csp_OptimizerStepTolerance", 0.01)),
        tol_p_rot=(float(p.get("csp_ToleranceParticlesPsi", 30.0)), float(p.get("csp_ToleranceParticlesTheta", 30.0)), float(p.get("csp_ToleranceParticlesPhi", 30.0))),
        tol_p_shift=float(p.get("csp_ToleranceParticlesShifts", 20.0)),
        tol_m_rot=(float(p.get("csp_ToleranceMicrographTiltAngles", 1.5)), float(p.get("csp_ToleranceMicrographTiltAxisAngles", 1.0)), 0.0),
        tol_m_shift=float(p.get("csp_ToleranceMicrographShifts", 100.0)), tol_defocus=float(p.get("csp_ToleranceMicrographDefocus1", 750.0)),
        normalize=int(bool(p.get("reconstruct_norm", True))), invert=int(bool(p.get("refine_invert", False))),
        data_set=str(p.get("data_set", "")), running_frames=int(p.get("csp_running_average_frames", 2)))


def _out_names(param_file, first, last):
    base = param_file[:-len(".cistem")]
    return "%s_%06d_%06d.cistem" % (base, first, last), "%s_%06d_%06d_extended.cistem" % (base, first, last)


def split_command(command):
    """'<csp> a b c … > <log>' as create_csp_split_commands formats it (local_run.py:364-376, :392-404, :451-463) ->
    (argv incl. the program, log file)."""
    left, _, log = command.partition(" > ")
    return left.split(), log.strip()


def parse_argv(argv):
    """The eight positional arguments; raises ValueError with an ERROR line."""
    if len(argv) != 8:
        raise ValueError("ERROR: csp: usage: csp <param.cistem> <param_extended.cistem> <mode> <first> <last> <flag> <images> <stack>")
    param_file, ext_file, mode_s, first_s, last_s, flag, images, stack = argv
    try:
        mode, first, last = float(mode_s), int(first_s), int(last_s)
    except ValueError:
        raise ValueError(f"ERROR: csp: mode / first / last must be numbers, got {mode_s} {first_s} {last_s}")
    return dict(param_file=param_file, ext_file=ext_file, mode=int(mode) if mode == int(mode) else mode, first=first, last=last, flag=flag,
                images=images, stack=stack)


def merge_alignment_parameters(outputs, outputs_extended):
    """What the caller does with the per-job files (particle_cspt.py:96-138 -> cistem_star_file.py:655-692): rows of the
    outputs in the given (sorted) order stacked and sorted by POSITION_IN_STACK; particle entries and (TIND, RIND) tilt entries
    of the extended files merged in order, later files overwriting earlier ones (the ORIGINAL extended file goes first).
    Returns (rows, particles [P, 12], tilts [T, 6]) with the blocks in first-seen key order, as the dictionaries keep them."""
    rows = np.vstack([cistem.read_parameters(f) for f in outputs])
    rows = rows[np.argsort(rows[:, C["POSITION_IN_STACK"]])]
    particles, tilts = {}, {}
    for f in outputs_extended:
        e = cistem.read_extended(f)
        for p in e["particles"]:
            particles[int(p[0])] = p
        for t in e["tilts"]:
            tilts.setdefault(int(t[0]), {})[int(t[1])] = t
    pb = np.array(list(particles.values())).reshape(-1, len(cistem.PARTICLE_COLUMNS))
    tb = np.array([t for d in tilts.values() for t in d.values()]).reshape(-1, len(cistem.TILT_COLUMNS))
    return rows, pb, tb


def csp_main(argv=None):
    t0 = time.time()
    argv = list(sys.argv[1:] if argv is None else argv)
    try:
        a = parse_argv(argv)
    except ValueError as e:
        _die(str(e))
    param_file, ext_file, mode, first, last, flag, images, stack = (a[k] for k in ("param_file", "ext_file", "mode", "first", "last", "flag", "images", "stack"))
    print("\n        **   Welcome to CSP (MI355X / libpypmatch)   **\n")
    print(f"parameters {param_file}\nextended   {ext_file}\nmode {mode}  first {first}  last {last}  flag {flag}\nimages {images}\nstack  {stack}")
    for pth in (param_file, ext_file, ".pyp_config.toml"):
        if not os.path.exists(pth):
            _die(f"ERROR: csp: input file {pth} does not exist")
    if not param_file.endswith(".cistem"):
        _die("ERROR: csp: the parameter file must be a .cistem file")
    s = _settings(read_flat_toml(".pyp_config.toml"))
    rows = cistem.read_parameters(param_file)
    ext = cistem.read_extended(ext_file)
    particles, tilts = ext["particles"], ext["tilts"]
    if mode in (-2, -2.1):
        return _extract(s, rows, first, last, images, stack, t0, running=(mode == -2.1))
    # the caller's own codes for region-based refinement (align/core.py:1121-1133: 7 = particle shifts, 8 = particle rotations)
    # normally arrive already mapped to 2 / 1 (and then 2 -> 5 by local_run.py:334-335); accepted as given as well
    mode = {7: 2, 8: 1}.get(mode, mode)
    if mode not in (0, 1, 2, 3, 4, 5, 6):
        _die(f"ERROR: csp: unknown mode {mode}")
    unit = CSP_PARTICLES if mode in (1, 2, 5) else CSP_MICROGRAPHS
    rot, trans = mode in (0, 1, 5, 6), mode in (2, 3, 5, 6)
    key = C["PIND"] if unit == CSP_PARTICLES else C["TIND"]
    hi = last if last >= 0 else np.inf
    sel = np.where((rows[:, key] >= first) & (rows[:, key] <= hi))[0]
    if len(sel) == 0:
        _die(f"ERROR: csp: no rows with {'PIND' if unit == CSP_PARTICLES else 'TIND'} in {first}..{last}")
    # a particle sweep needs all rows of its particles (they are the selection); a tilt sweep all rows of its tilts
    rin = rows[sel].copy()
    if not os.path.exists(stack):
        _die(f"ERROR: csp: particle stack {stack} does not exist (run mode -2 first)")
    mm = mrc.mmap(stack)
    pos = rin[:, C["POSITION_IN_STACK"]].astype(np.int64)
    if pos.min() < 1 or pos.max() > mm.shape[0]:
        _die(f"ERROR: csp: {stack} has {mm.shape[0]} images, rows ask for {int(pos.max())}")
    imgs = np.ascontiguousarray(mm[pos - 1], dtype=np.float32)
    box = imgs.shape[1]
    refp = os.path.join(os.environ.get("PYP_SCRATCH", "."), f"{s['data_set']}_frames_CSP_01.mrc")
    if not os.path.exists(refp):
        _die(f"ERROR: csp: reference {refp} does not exist")
    vol = mrc.read(refp).astype(np.float32)
    if vol.shape != (box, box, box):
        _die(f"ERROR: csp: reference is {vol.shape}, particles are {box}^2")
    px = float(rin[0, C["PIXEL_SIZE"]]) if rin[0, C["PIXEL_SIZE"]] > 0 else s["pixel"]
    cfg = RefineCfg.make(box=box, pixel_size=px, molecular_mass_kda=s["mw"], mask_radius=s["radius"], res_low=s["res_low"], res_high=s["res_high"],
                         res_signed_cc=s["res_signed"], global_search=0, local_refine=1, normalize=s["normalize"], invert=s["invert"])
    cc = CspCfg.make(unit, refine_rotation=rot, refine_translation=trans, refine_defocus=int(mode == 4), defocus_range=s["tol_defocus"],
                     tol_angle=s["tol_p_rot"] if unit == CSP_PARTICLES else s["tol_m_rot"],
                     tol_shift=(s["tol_p_shift"] if unit == CSP_PARTICLES else s["tol_m_shift"]) / px,        # the tolerances are in Angstrom
                     step_tolerance=s["step_tol"], tind_min=s["tind_min"], tind_max=s["tind_max"], first=first, last=last)
    from .. import host, lib
    dev = int(os.environ.get("PPM_DEVICE", "0"))
    try:
        with gpu_lock(dev):
            ref = host.Reference(vol, box / 2, device=dev)
            rout, pout, tout = ref.csp_refine(cfg, cc, imgs, rin, particles, tilts)
            ref.close()
    except (lib.PpmError, ValueError) as e:
        _die(str(e))
    out_main, out_ext = _out_names(param_file, first, last)
    if unit == CSP_PARTICLES:
        keep = (pout[:, 0] >= first) & (pout[:, 0] <= hi)
        pblock, tblock = pout[keep], tout
    else:
        keep = (tout[:, 0] >= first) & (tout[:, 0] <= hi)
        pblock, tblock = pout, tout[keep]
    cistem.write_parameters(out_main, rout)
    cistem.write_extended(out_ext, pblock, tblock)
    print("\n   ROW    PIND  TIND     PSI   THETA     PHI       SHX       SHY     SCORE")
    for r in rout[:40]:
        print("%7d%8d%6d%8.2f%8.2f%8.2f%10.2f%10.2f%10.4f" % (r[0], r[C["PIND"]], r[C["TIND"]], r[1], r[2], r[3], r[4], r[5], r[C["SCORE"]]))
    print(f"\nRefined {int(keep.sum())} {'particles' if unit == CSP_PARTICLES else 'tilts'} over {len(rout)} projections in {time.time() - t0:.1f} s; "
          f"mean score {rout[:, C['SCORE']].mean():.4f} (was {rin[:, C['SCORE']].mean():.4f})")
    print("\nCSP: Normal termination\n", flush=True)
    return 0


def running_average(stack, rows, half_width, radius_px):
    """Mode -2.1 (csp_produce_running_average, src/pyp/align/core.py:1000-1001; the caller merges the outputs into
    frealign/<name>_stack_weighted_average.mrc, :1170): every row's box becomes the weighted average of the boxes of the SAME particle
    (PIND) in the SAME movie / tilt (IMIND) whose frame index FIND lies within `half_width` of the row's own, with weights
    exp(-d^2 / (2 (half_width / 2)^2)) of the frame distance d - one image per row, so the stack still matches the parameter file.
    The averaged boxes are normalised again on their background ring (mean 0, sigma 1 outside `radius_px`, like
    src/pyp/analysis/image.py:406-417).  Build-defined: the absent program's weights are not visible (the option is hidden in
    config/pyp_config.toml:6528-6534, "currently only for testing classification").  half_width = 0 returns the stack unchanged."""
    if half_width <= 0:
        return stack
    n, box = len(rows), stack.shape[1]
    key = np.stack([rows[:, C["PIND"]], rows[:, C["IMIND"]]], axis=1).astype(np.int64)
    fidx = rows[:, C["FIND"]].astype(np.int64)
    order = np.lexsort((fidx, key[:, 1], key[:, 0]))
    out = np.empty_like(stack)
    yy, xx = np.mgrid[:box, :box]
    bg = ((yy - box // 2) ** 2 + (xx - box // 2) ** 2) > radius_px * radius_px
    sig = max(half_width / 2.0, 1e-6)
    lo = 0
    while lo < n:
        hi = lo + 1
        while hi < n and np.array_equal(key[order[hi]], key[order[lo]]):
            hi += 1
        grp = order[lo:hi]
        f = fidx[grp]
        for j, fj in zip(grp, f):
            d = f - fj
            use = np.abs(d) <= half_width
            w = np.exp(-(d[use].astype(np.float64) ** 2) / (2 * sig * sig))
            avg = np.tensordot(w / w.sum(), stack[grp[use]].astype(np.float64), axes=1)
            b = avg[bg] if bg.any() else avg.ravel()
            sd = b.std()
            out[j] = ((avg - b.mean()) / (sd if sd > 0 else 1.0)).astype(np.float32)
        lo = hi
    return out


def _extract(s, rows, first, last, images, stack, t0, running=False):
    """Mode -2: boxes of `extract_box` pixels around (ORIGINAL_X_POSITION, ORIGINAL_Y_POSITION) of section IMIND of the tilt
    series, normalised like every PYP particle stack (src/pyp/analysis/image.py:406-417), written in row order.  Mode -2.1
    (`running`): the same boxes replaced by running frame averages (running_average; half-width csp_running_average_frames of
    .pyp_config.toml, default 2 = five frames)."""
    if not os.path.exists(images):
        _die(f"ERROR: csp: {'frame list' if str(images).endswith('.txt') else 'tilt series'} {images} does not exist")
    if s["extract_bin"] != 1:
        _die("ERROR: csp: extract_bin other than 1 is not supported")
    hi = last if last >= 0 else np.inf
    sel = np.where((rows[:, C["PIND"]] >= first) & (rows[:, C["PIND"]] <= hi))[0]
    if len(sel) == 0:
        _die(f"ERROR: csp: no rows with PIND in {first}..{last}")
    r = rows[sel]
    if str(images).endswith(".txt"):
        return _extract_frames(s, r, first, last, images, stack, t0, running)
    series = mrc.mmap(images)
    if series.ndim == 2:
        series = series[None]
    box, px = s["box"], s["pixel"]
    from .. import host, lib
    dev = int(os.environ.get("PPM_DEVICE", "0"))
    out = np.empty((len(r), box, box), dtype=np.float32)
    try:
        with gpu_lock(dev):
            lib.init(dev)
            for im in np.unique(r[:, C["IMIND"]].astype(np.int64)):
                if im < 0 or im >= series.shape[0]:
                    _die(f"ERROR: csp: row asks for section {im}, {images} has {series.shape[0]}")
                idx = np.where(r[:, C["IMIND"]].astype(np.int64) == im)[0]
                coords = np.stack([r[idx, C["ORIGINAL_X_POSITION"]], r[idx, C["ORIGINAL_Y_POSITION"]]], axis=1)
                out[idx] = host.extract_boxes(np.ascontiguousarray(series[im], dtype=np.float32), coords, box, s["radius"], px, device=dev)
    except (lib.PpmError, ValueError) as e:
        _die(str(e))
    if running:
        out = running_average(out, r, s["running_frames"], s["radius"] / px)
    mrc.write(out, stack, pixel_size=px)
    print(f"\nExtracted {len(r)} projections of particles {first}..{last} into {stack} in {time.time() - t0:.1f} s"
          + (f" (running averages over +-{s['running_frames']} frames)" if running else ""))
    print("\nCSP: Normal termination\n", flush=True)
    return 0


def _extract_frames(s, r, first, last, images, stack, t0, running=False):
    """Mode -2 from a frame list (`frames_csp.txt`, written by src/pyp/extract/core.py:620-625: one movie file per line, in
    tilt-series order): a row is cut out of section FIND of movie IMIND, at (ORIGINAL_X_POSITION + FSHIFT_X, ORIGINAL_Y_POSITION
    + FSHIFT_Y) rounded down like the box corner itself.  The reference's own use of the list lives in the absent binary; this
    assignment of the IMIND / FIND / FSHIFT columns (cistem_star_file.py:596-628) is build-defined."""
    with open(images) as f:
        movies = [ln.strip() for ln in f if ln.strip()]
    if not movies:
        _die(f"ERROR: csp: frame list {images} is empty")
    box, px = s["box"], s["pixel"]
    from .. import host, lib
    dev = int(os.environ.get("PPM_DEVICE", "0"))
    out = np.empty((len(r), box, box), dtype=np.float32)
    im_all, fr_all = r[:, C["IMIND"]].astype(np.int64), r[:, C["FIND"]].astype(np.int64)
    try:
        with gpu_lock(dev):
            lib.init(dev)
            for im in np.unique(im_all):
                if im < 0 or im >= len(movies):
                    _die(f"ERROR: csp: row asks for movie {im}, {images} lists {len(movies)}")
                if not os.path.exists(movies[im]):
                    _die(f"ERROR: csp: movie {movies[im]} of {images} does not exist")
                mv = mrc.mmap(movies[im])
                if mv.ndim == 2:
                    mv = mv[None]
                for fr in np.unique(fr_all[im_all == im]):
                    if fr < 0 or fr >= mv.shape[0]:
                        _die(f"ERROR: csp: row asks for frame {fr}, {movies[im]} has {mv.shape[0]}")
                    idx = np.where((im_all == im) & (fr_all == fr))[0]
                    coords = np.stack([r[idx, C["ORIGINAL_X_POSITION"]] + r[idx, C["FSHIFT_X"]],
                                       r[idx, C["ORIGINAL_Y_POSITION"]] + r[idx, C["FSHIFT_Y"]]], axis=1)
                    out[idx] = host.extract_boxes(np.ascontiguousarray(mv[fr], dtype=np.float32), coords, box, s["radius"], px, device=dev)
    except (lib.PpmError, ValueError) as e:
        _die(str(e))
    if running:
        out = running_average(out, r, s["running_frames"], s["radius"] / px)
    mrc.write(out, stack, pixel_size=px)
    print(f"\nExtracted {len(r)} frame projections of particles {first}..{last} from {len(movies)} movies into {stack} in {time.time() - t0:.1f} s"
          + (f" (running averages over +-{s['running_frames']} frames)" if running else ""))
    print("\nCSP: Normal termination\n", flush=True)
    return 0
