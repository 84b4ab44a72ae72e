"""Score-based particle selection between refinement and reconstruction, numpy only.

Restates `shape_phase_residuals` (src/pyp/analysis/scores.py:300-761, called with `scores=True` by
`call_shape_phase_residuals`, :764-825): particles whose SCORE falls below a per-(orientation, defocus)-group threshold, or
outside the score / defocus / azimuth / frame / tilt windows, get OCCUPANCY 0, which is how `reconstruct3d`
(`ppm_insert_batch`) is told to skip them.  PINNED: tests/golden/gen_golden_r03.py runs the reference's own function on
240 single-particle rows and 336 tomography rows (thresholds 0 / fraction / 1, groups, every window, odd / even, pooled
particle indices); tests/test_golden_r03.py reproduces its OCCUPANCY column exactly and POSITION_IN_STACK as it leaves it.
The bimodal automatic threshold (`threshold == 0`, `optimal_threshold` of src/pyp/analysis/statistics.py:10-150) is pinned
on its own by tests/golden/golden_r02.json.  Not restated: match-stack sorting, the consistency filter (both off in
`call_shape_phase_residuals`) and the plots.
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


def _gauss(x, x0, sigma):
    return np.exp(-((x - x0) ** 2.0) / (2.0 * sigma ** 2.0))


def optimal_threshold(samples, criteria="optimal"):
    """Score threshold between the two modes of a bimodal sample (statistics.py:10-102): two-component Gaussian mixture
    (scikit-learn, the reference's own dependency, same settings), the crossing of the two weighted components between the
    means; when the components do not cross there (or their sum exceeds both peaks) one Gaussian is fitted instead and the
    threshold is its mean - 3 sigma.  Constant samples give 1."""
    samples = np.asarray(samples, dtype=np.float64)
    if np.var(samples) == 0:
        return 1
    from sklearn.mixture import GaussianMixture
    trapz = getattr(np, "trapezoid", None) or np.trapz
    gmm = GaussianMixture(n_components=2, covariance_type="full", tol=1e-6, reg_covar=1e-6).fit(X=samples.reshape(-1, 1))
    x = np.linspace(samples.min(), samples.max(), 5000)
    comp = []
    total = np.zeros_like(x, dtype=np.float32)
    for m, c, w in zip(gmm.means_.ravel(), gmm.covariances_.ravel(), gmm.weights_.ravel()):
        g = _gauss(x, m, np.sqrt(c))
        comp.append(g / trapz(g, x) * w)
        total += (g / trapz(g, x) * w).astype(np.float32)
    m1, m2 = gmm.means_[0], gmm.means_[1]
    a, b = int(np.argmin(np.fabs(x - m2))), int(np.argmin(np.fabs(x - m1)))
    if a > b:
        a, b = b, a
    minimum = a + int(np.argmin(total[a:b]))
    g1, g2 = comp[0], comp[1]
    opt = int(np.argmin(np.fabs(g1[a:b] - g2[a:b])))
    same_side = (g1[a + opt - 1] - g2[a + opt - 1]) * (g1[a + opt + 1] - g2[a + opt + 1]) > 0
    if same_side or (total[a + opt] > g1.max() and total[a + opt] > g2.max()):
        one = GaussianMixture(n_components=1, covariance_type="full", tol=1e-6, reg_covar=1e-6).fit(X=samples.reshape(-1, 1))
        return float((one.means_[0] - 3.0 * np.sqrt(one.covariances_[0]))[0][0])
    if "opt" in criteria:
        return float(x[a + opt])
    if "min" in criteria:
        return float(minimum)
    return float(np.mean(gmm.means_))


def tilt_angles_from_table(rows, table):
    """Per-row tilt angle from the `<job>_rNN.json` side-car (`{film: {TIND: angle}}`, particle_cspt.py:434-456) the way
    scores.py:340-377 does it: the data set counts as tomography when any angle of the FIRST film of the table is non-zero;
    otherwise every row gets 0.  Rows whose film or TIND is missing from the table get NaN (pandas' map), which no window keeps."""
    rows = np.asarray(rows, dtype=np.float64)
    first = next(iter(table.values())) if table else {}
    if not any(abs(float(v)) > 0 for v in first.values()):
        return np.zeros(rows.shape[0])
    film = rows[:, COL["IMAGE_IS_ACTIVE"]].astype(np.int64)
    tind = rows[:, C_TIND].astype(np.int64)
    out = np.full(rows.shape[0], np.nan)
    for f in np.unique(film):
        ang = {int(k): float(v) for k, v in table.get(str(int(f)), {}).items()}
        m = film == f
        out[m] = [ang.get(int(t), np.nan) for t in tind[m]]
    return out


def _mean_by_particle(scores, pind):
    """pandas' groupby("pind")["score"].mean(): means in ascending particle-index order, and the indices."""
    ids, inv = np.unique(pind, return_inverse=True)
    return np.bincount(inv, weights=scores) / np.bincount(inv), ids


def select_particles(rows, threshold, angles=1, defocuses=1, mindefocus=0.0, maxdefocus=1.0e9, firstframe=0, lastframe=-1,
                     mintilt=-90.0, maxtilt=90.0, minazh=0.0, maxazh=180.0, minscore=0.0, maxscore=1.0, odd=False, even=False,
                     renumber=False, tilt_angles=None):
    """Return a copy of the float64 [M, 32] table with OCCUPANCY zeroed for the rejected particles.

    threshold in (0, 1]: fraction of each group kept by score (scores.py:480-506); threshold > 1 never matches the
    reference's `cluster.ndim == 2` test and therefore removes nothing (:514-525), reproduced as such; threshold == 0:
    1.075 x the bimodal `optimal_threshold` of the group's scores, used when the group has more than 20 of them (:437-462).
    tilt_angles: per-row tilt angle (the `<job>_rNN.json` table of particle_cspt.py:434-456 looked up by film and TIND);
    any non-zero value selects the tomography rules: thresholds come from per-particle mean scores over |tilt| <= 12 and
    a particle whose mean over |tilt| < 10 falls below its group's threshold loses all its rows of the group (:572-607).
    """
    from scipy.ndimage import gaussian_filter
    out = np.array(rows, dtype=np.float64, copy=True)
    M = out.shape[0]
    if M == 0:
        return out
    tilt = np.zeros(M) if tilt_angles is None else np.asarray(tilt_angles, dtype=np.float64).reshape(M)
    is_tomo = bool(np.any(np.abs(tilt) > 0))
    pind = out[:, COL["PIND"]]
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
            low = cluster & (np.abs(tilt) <= 12)
            if threshold == 0:
                use = _mean_by_particle(sc[low], pind[low])[0] if is_tomo else prs
                if use.size > 20:
                    thr[g, f] = 1.075 * optimal_threshold(use, "optimal")
            elif threshold <= 1:
                if is_tomo:
                    ms = _mean_by_particle(sc[low], pind[low])[0]
                    if ms.size:
                        thr[g, f] = np.sort(ms)[int((ms.shape[0] - 1) * (1 - threshold))]
                else:
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
            if is_tomo and thr[g, f] > 0:
                near = grp & (np.abs(tilt) < 10)
                ms, ids = _mean_by_particle(sc[near], pind[near])
                drop = ids[~(ms >= thr[g, f])] if threshold != 1 else ids[:0]
                occ[grp & np.isin(pind, drop)] = 0.0
                with np.errstate(invalid="ignore"):
                    occ[grp & ((sc < lo[g, f]) | (sc > hi[g, f]))] = 0.0
                continue
            with np.errstate(invalid="ignore"):
                bad = (sc < thr[g, f]) | (sc < lo[g, f]) | (sc > hi[g, f])
            occ[grp & bad] = 0.0
    occ[(out[:, C_DF1] < mindefocus) | (out[:, C_DF1] > maxdefocus)] = 0.0                       # :640-645
    if maxazh < 180 or minazh > 0:                                                                 # :649-658
        az = np.mod(out[:, C_THETA], 180.0)
        occ[(az < minazh) | (az > maxazh)] = 0.0
    if lastframe > -1:                                                                             # :663-669
        occ[(out[:, C_TIND] < firstframe) | (out[:, C_TIND] > lastframe)] = 0.0
    occ[(tilt < mintilt) | (tilt > maxtilt)] = 0.0                                               # :672-677 (tilt angle 0 for SPA)
    if odd:
        occ[::2] = 0.0
    if even:
        occ[1::2] = 0.0
    if renumber:      # what the reference does when the output name ends in `_used.cistem` (:755-759; set_data keeps a view, so the
        # assignment after it still lands in the table that is written — seen in the generated fixtures)
        out[:, C_POS] = np.arange(1, M + 1)
    return out
