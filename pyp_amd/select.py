"""Score-based particle selection between refinement and reconstruction, numpy only.

Restates the single-particle branch of `shape_phase_residuals` (src/pyp/analysis/scores.py:300-761, called with
`scores=True` by `call_shape_phase_residuals`, :764-825): particles whose SCORE falls below a per-(orientation, defocus)-group
threshold, or outside the score / defocus / azimuth / frame windows, get OCCUPANCY 0, which is how `reconstruct3d`
(`ppm_insert_batch`) is told to skip them.  PARITY UNPINNED: the reference module cannot be imported under Python 3.10
(its import chain reaches an f-string that needs 3.12) and the tree holds no fixture for it; the tests check the rule on
hand-made tables.  Not restated: the tomography branch (per-particle mean scores over low tilts), the bimodal automatic
threshold (`threshold == 0`), match-stack sorting, the consistency filter and the plots.
"""
import math

import numpy as np

from .formats.cistem import COL

C_THETA, C_DF1, C_OCC, C_SCORE, C_TIND, C_POS = COL["THETA"], COL["DEFOCUS_1"], COL["OCCUPANCY"], COL["SCORE"], COL["TIND"], COL["POSITION_IN_STACK"]


def assign_groups(rows, angles, defocuses):
    """scores.py:253-270: orientation group from THETA mod 180, defocus group from DEFOCUS_1 between its floor(min) and
    ceil(max)."""
    ag = np.floor(np.mod(rows[:, C_THETA], 180.0) * angles / 180.0)
    if rows.shape[0] == 0:
        return ag, np.zeros_like(ag)
    mind, maxd = int(math.floor(rows[:, C_DF1].min())), int(math.ceil(rows[:, C_DF1].max()))
    if maxd == mind:
        dg = np.zeros(ag.shape)
    else:
        dg = np.round((rows[:, C_DF1] - mind) / (maxd - mind) * (defocuses - 1))
    return ag, dg


def select_particles(rows, threshold, angles=1, defocuses=1, mindefocus=0.0, maxdefocus=1.0e9, firstframe=0, lastframe=-1,
                     mintilt=-90.0, maxtilt=90.0, minazh=0.0, maxazh=180.0, minscore=0.0, maxscore=1.0, odd=False, even=False,
                     renumber=False):
    """Return a copy of the float64 [M, 32] table with OCCUPANCY zeroed for the rejected particles.

    threshold in (0, 1]: fraction of each group kept by score (scores.py:480-506); threshold > 1 never matches the
    reference's `cluster.ndim == 2` test and therefore removes nothing (:514-525), reproduced as such; threshold == 0
    (automatic bimodal cutoff) is not built and raises.
    """
    from scipy.ndimage import gaussian_filter
    out = np.array(rows, dtype=np.float64, copy=True)
    M = out.shape[0]
    if M == 0:
        return out
    if threshold == 0:
        raise ValueError("ERROR: automatic score threshold (reconstruct_cutoff 0) is not built; give a fraction in (0, 1]")
    sc = out[:, C_SCORE]
    ag, dg = assign_groups(out, angles, defocuses)
    thr = np.full((angles, defocuses), np.nan); lo = np.full((angles, defocuses), np.nan); hi = np.full((angles, defocuses), np.nan)
    for g in range(angles):
        for f in range(defocuses):
            cluster = (ag == g) & (dg == f)
            size = 1
            while cluster.sum() < 100 and M > 100:          # widen small groups (:420-434)
                cluster = (ag >= g - size) & (ag <= g + size) & (dg >= f - size) & (dg <= f + size)
                size += 1
            prs = sc[cluster]
            if prs.size == 0:
                continue
            if threshold <= 1:
                # the reference indexes with the length of the boolean mask (= M); identical for one group, clamped here
                # so that several groups cannot index past the group's end
                k = min(int((M - 1) * (1 - threshold)), prs.size - 1)
                thr[g, f] = np.sort(prs)[k]
            lo[g, f] = prs.min() + minscore * (prs.max() - prs.min()) if minscore < 1 else minscore
            hi[g, f] = prs.max() - (1 - maxscore) * (prs.max() - prs.min()) if maxscore <= 1 else maxscore
    thr = gaussian_filter(thr, sigma=1)                        # :560 (a NaN group spreads to its neighbours, as there)
    occ = out[:, C_OCC]
    for g in range(angles):
        for f in range(defocuses):
            grp = (ag == g) & (dg == f)
            with np.errstate(invalid="ignore"):
                bad = (sc < thr[g, f]) | (sc < lo[g, f]) | (sc > hi[g, f])
            occ[grp & bad] = 0.0
    occ[(out[:, C_DF1] < mindefocus) | (out[:, C_DF1] > maxdefocus)] = 0.0                       # :640-645
    if maxazh < 180 or minazh > 0:                                                                 # :649-658
        az = np.mod(out[:, C_THETA], 180.0)
        occ[(az < minazh) | (az > maxazh)] = 0.0
    if lastframe > -1:                                                                             # :663-669
        occ[(out[:, C_TIND] < firstframe) | (out[:, C_TIND] > lastframe)] = 0.0
    if 0.0 < mintilt or 0.0 > maxtilt:                                                             # :672-677, tilt angle 0 for SPA
        occ[:] = 0.0
    if odd:
        occ[::2] = 0.0
    if even:
        occ[1::2] = 0.0
    if renumber:      # the reference assigns POSITION_IN_STACK after it has already stored the table (:755-759): off by default
        out[:, C_POS] = np.arange(1, M + 1)
    return out
