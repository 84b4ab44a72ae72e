"""Drop-in `csp`: constrained refinement of tilt-series particles and the extraction of their projections, behind the argv
PYP builds in create_csp_split_commands (src/pyp/system/local_run.py:306-467):

    csp <param.cistem> <param_extended.cistem> <mode> <first> <last> <flag> <images> <stack> > <log>

The program reads its settings from `.pyp_config.toml` in the working directory, as the reference's does
(src/pyp/align/core.py:1055: project_params.save_parameters right before the fan-out; CamelCase `csp_*` keys,
config/pyp_config.toml:6241-6641) and the current map from $PYP_SCRATCH/<data_set>_frames_CSP_01.mrc (align/core.py:916-931).

Modes (src/pyp/align/core.py:1015-1023 and the 2 -> 5, 3 -> 6 remaps of local_run.py:332-335, :413, :428):
   -2  extract the projections of particles first..last from the tilt series into <stack>
    1 / 2 / 5  particle rotations / 3-D shifts / both, particles first..last (PIND)
    0 / 3 / 6  tilt angle + axis / image shifts / both, tilts first..last (TIND; last = -1: up to the end)
    4  one defocus offset per tilt (csp_ToleranceMicrographDefocus1 either side, 50 A steps), tilts first..last
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
        data_set=str(p.get("data_set", "")))


def _out_names(param_file, first, last):
    base = param_file[:-len(".cistem")]
    return "%s_%06d_%06d.cistem" % (base, first, last), "%s_%06d_%06d_extended.cistem" % (base, first, last)


def csp_main(argv=None):
    t0 = time.time()
    argv = list(sys.argv[1:] if argv is None else argv)
    if len(argv) != 8:
        _die("ERROR: csp: usage: csp <param.cistem> <param_extended.cistem> <mode> <first> <last> <flag> <images> <stack>")
    param_file, ext_file, mode_s, first_s, last_s, flag, images, stack = argv
    try:
        mode, first, last = int(float(mode_s)), int(first_s), int(last_s)
    except ValueError:
        _die(f"ERROR: csp: mode / first / last must be integers, got {mode_s} {first_s} {last_s}")
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
    if mode == -2:
        return _extract(s, rows, first, last, images, stack, t0)
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


def _extract(s, rows, first, last, images, stack, t0):
    """Mode -2: boxes of `extract_box` pixels around (ORIGINAL_X_POSITION, ORIGINAL_Y_POSITION) of section IMIND of the tilt
    series, normalised like every PYP particle stack (src/pyp/analysis/image.py:406-417), written in row order."""
    if not str(images).endswith(".mrc"):
        _die("ERROR: csp: frame lists (frames_csp.txt) are not supported; give the tilt-series .mrc")
    if not os.path.exists(images):
        _die(f"ERROR: csp: tilt series {images} does not exist")
    if s["extract_bin"] != 1:
        _die("ERROR: csp: extract_bin other than 1 is not supported")
    hi = last if last >= 0 else np.inf
    sel = np.where((rows[:, C["PIND"]] >= first) & (rows[:, C["PIND"]] <= hi))[0]
    if len(sel) == 0:
        _die(f"ERROR: csp: no rows with PIND in {first}..{last}")
    r = rows[sel]
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
    mrc.write(out, stack, pixel_size=px)
    print(f"\nExtracted {len(r)} projections of particles {first}..{last} into {stack} in {time.time() - t0:.1f} s")
    print("\nCSP: Normal termination\n", flush=True)
    return 0
