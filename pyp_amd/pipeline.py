"""One refinement iteration held in memory: refine -> select -> reconstruct -> finalise (the work of
`split_refinement` (frealign.py:3014-3193), `shape_phase_residuals` (scores.py:300-761), `split_reconstruction` (:1622-1835),
`local_merge_reconstruction` / `merge_reconstructions` (:1838-2175) for one class), without the files in between.
Sharded runs call this once per rank and reduce the accumulator with `pyp_amd.dist.reduce_accumulators`."""
import numpy as np

from . import host, select
from .abi import FinalCfg, ReconCfg
from .formats.cistem import COL


def iteration(vol, stack, rows, refine_cfg, *, pixel_size, molecular_mass_kda, symmetry="C1", keep_fraction=1.0,
              score_weight_bfactor=0.0, outer_radius=None, split_by_pind=False):
    """vol: current reference (N^3 float32); stack: [M, N, N] float32 (numpy or a CUDA torch tensor); rows: [M, 32].
    Returns dict(rows=refined table incl. OCC after selection, half1, half2, filtered, stats=[N/2-1, 7] statistics table).
    """
    n = int(refine_cfg.box)
    ref = host.Reference(vol, n / 2)
    try:
        refined = ref.refine(refine_cfg, stack, rows)
    finally:
        ref.close()
    used = select.select_particles(refined, threshold=keep_fraction) if keep_fraction < 1.0 else refined
    score = used[used[:, COL["OCCUPANCY"]] > 0, COL["SCORE"]]
    rc = ReconCfg(box=n, pixel_size=pixel_size, res_limit=2 * pixel_size, score_weight_bfactor=score_weight_bfactor,
                  score_average=float(score.mean()) if score.size else 0.0, score_threshold=0.0, normalize=1, invert=0,
                  split_by_pind=1 if split_by_pind else 0, mask_radius=refine_cfg.mask_radius)
    acc = host.Accumulator(n, pixel_size, symmetry)
    try:
        acc.insert(rc, stack, used)
        fc = FinalCfg(molecular_mass_kda=molecular_mass_kda, inner_radius=0.0,
                      outer_radius=outer_radius if outer_radius is not None else 0.45 * n * pixel_size, mask_falloff=0.0)
        h1, h2, filt, stats = acc.finalize(fc)
    finally:
        acc.close()
    return dict(rows=used, half1=h1, half2=h2, filtered=filt, stats=np.asarray(stats))
