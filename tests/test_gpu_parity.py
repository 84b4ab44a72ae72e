"""Parity of the HIP path (through the C ABI, libpypmatch.so) with the CPU oracle on identical
seeded inputs.  Tolerances: BASELINE.json states 0.1 deg / 0.5 px for poses; scores and maps are
compared to float32 round-off (stated per test).  Run on the GPU box:  pytest -m gpu"""
import numpy as np
import pytest

from pyp_amd import synth
from pyp_amd.abi import FinalCfg, ReconCfg, RefineCfg

pytestmark = pytest.mark.gpu

ANG_TOL_DEG, SHIFT_TOL_PX = 0.1, 0.5       # BASELINE.json north_star


@pytest.fixture(scope="module")
def H():
    from pyp_amd import host
    return host


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


def dataset(n, m, px, snr):
    vol, stack, rows = synth.make_dataset(n, m, pixel=px, snr=snr)
    return vol, stack.numpy(), rows


def cfg_for(n, px, **kw):
    base = dict(box=n, pixel_size=px, mask_radius=0.4 * n * px, res_high=px * n / (0.375 * n), res_search=px * n / (0.16 * n),
                search_range_x=6 * px, search_range_y=6 * px, res_signed_cc=30.0)
    base.update(kw)
    return RefineCfg.make(**base)


@pytest.fixture(scope="module")
def d64(H, O):
    vol, imgs, rows = dataset(64, 24, 2.0, 0.1)
    return vol, imgs, rows, H.Reference(vol, 32), O.Reference(vol, 32)


def test_score_at_given_poses_matches_oracle(d64, H, O):
    vol, imgs, rows, g, o = d64
    for kw in (dict(), dict(res_signed_cc=0.0), dict(res_low=60.0), dict(invert=1), dict(normalize=0)):
        c = cfg_for(64, 2.0, global_search=0, local_refine=0, **kw)
        want = O.score_batch(o, c, imgs, rows)
        got = g.refine(c, imgs, rows)[:, 14] / 100.0
        assert np.abs(want - got).max() < 2e-5, kw                       # float32 round-off of a normalised sum


def test_focus_mask_matches_oracle(d64, H, O):
    """Answers 29-32 + 44 (class_focusmask): the disc around the projected focus sphere replaces the centred mask.  A sphere at
    the box centre with the mask radius IS the centred mask for a row without shift (bit-identical rows); an off-centre sphere gives other scores,
    the same on both sides, and its disc follows the row's pose and shift."""
    vol, imgs, rows, g, o = d64
    base = cfg_for(64, 2.0, global_search=0, local_refine=0)
    rows0 = rows.copy(); rows0[:, 4:6] = 0.0                       # the disc follows the row's shift: no shift, no offset
    assert np.array_equal(g.refine(base, imgs, rows0),
                          g.refine(cfg_for(64, 2.0, global_search=0, local_refine=0, focus=(0.0, 0.0, 0.0, 0.4 * 64 * 2.0)), imgs, rows0))
    plain = g.refine(base, imgs, rows)
    off = cfg_for(64, 2.0, global_search=0, local_refine=0, focus=(14.0, -9.0, 11.0, 26.0))          # Angstrom from the box centre
    want = O.score_batch(o, off, imgs, rows)
    got = g.refine(off, imgs, rows)[:, 14] / 100.0
    assert np.abs(want - got).max() < 2e-5
    assert np.abs(got - plain[:, 14] / 100.0).max() > 1e-3                                            # a different region is scored
    start = synth.perturb_rows(rows, 2.0, 1.0, 2.0)
    c = cfg_for(64, 2.0, global_search=0, focus=(14.0, -9.0, 11.0, 40.0))
    want, _ = O.refine_batch(o, c, imgs, start)
    got = g.refine(c, imgs, start)
    assert synth.angular_error_deg(want, got).max() < ANG_TOL_DEG and synth.shift_error_px(want, got, 2.0).max() < SHIFT_TOL_PX
    cg = cfg_for(64, 2.0, focus=(14.0, -9.0, 11.0, 40.0))                                               # global search under the focus disc
    want, _ = O.refine_batch(o, cg, imgs[:6], rows[:6])
    got = g.refine(cg, imgs[:6], rows[:6])
    assert synth.angular_error_deg(want, got).max() < ANG_TOL_DEG and synth.shift_error_px(want, got, 2.0).max() < SHIFT_TOL_PX


def test_matching_projections_match_oracle_and_an_independent_projector(d64, H, O):
    """Answers 8 / 43 (refine_fmatch): reference projected at the row's pose x CTF at the row's shift.  HIP = oracle to float
    round-off; and both overlay the noise-free image an independent real-space projector (pyp_amd.synth, torch) renders for
    the same rows — sign, shift direction, CTF and scale conventions at once."""
    vol, imgs, rows, g, o = d64
    c = cfg_for(64, 2.0, res_high=2.0 * 2.0 * 64 / 62.0)                 # band limit just inside Nyquist
    want = O.match_projections(o, c, rows[:8])
    got = g.match_projections(c, rows[:8])
    assert got.shape == (8, 64, 64) and np.linalg.norm(got - want) / np.linalg.norm(want) < 1e-4
    clean = synth.render_rows(vol, rows[:8], 2.0, snr=0, normalize=False).numpy()      # the same rows, no noise, density scale kept
    for a, b in zip(got, clean):
        a0, b0 = a - a.mean(), b - b.mean()
        cc = float((a0 * b0).sum() / np.sqrt((a0 * a0).sum() * (b0 * b0).sum()))
        assert cc > 0.98, cc
        assert 0.9 < float((a0 * b0).sum() / (b0 * b0).sum()) < 1.1      # same scale
    neg = g.match_projections(cfg_for(64, 2.0, res_high=2.0 * 2.0 * 64 / 62.0, invert=1), rows[:2])
    assert np.allclose(neg, -got[:2], atol=1e-6)


def test_beam_tilt_matches_oracle(H, O):
    """BEAM_TILT_X / Y columns: k_prep removes the phase term exp(i phi) from every particle spectrum; scores, refinement and
    insertion then see the corrected image.  HIP = oracle, and the rows' tilt restores the untilted scores."""
    n, px = 64, 1.5
    vol = synth.phantom(n)
    _, _, rows = synth.make_dataset(n, 8, pixel=px, snr=0, vol=vol)
    C = synth.cistem.COL
    tilted = rows.copy(); tilted[:, C["BEAM_TILT_X"]] = 1.5; tilted[:, C["BEAM_TILT_Y"]] = -1.0
    imgs_t = synth.render_rows(vol, tilted, px, snr=0).numpy()
    g, o = H.Reference(vol, n / 2), O.Reference(vol, n / 2)
    c = RefineCfg.make(box=n, pixel_size=px, mask_radius=0.4 * n * px, res_high=2.2 * px, global_search=0, local_refine=0)
    want = O.score_batch(o, c, imgs_t, tilted)
    got = g.refine(c, imgs_t, tilted)[:, 14] / 100.0
    assert np.abs(want - got).max() < 2e-5
    ignored = g.refine(c, imgs_t, rows)[:, 14] / 100.0
    assert (got - ignored).min() > 0.02                                  # the term matters at 1.5 A / pixel and 1.8 mrad
    start = synth.perturb_rows(tilted, 2.0, 1.0, 2.0)
    cl = RefineCfg.make(box=n, pixel_size=px, mask_radius=0.4 * n * px, res_high=2.2 * px, global_search=0, res_signed_cc=30.0)
    w2, _ = O.refine_batch(o, cl, imgs_t, start)
    g2 = g.refine(cl, imgs_t, start)
    assert synth.angular_error_deg(w2, g2).max() < ANG_TOL_DEG and synth.shift_error_px(w2, g2, px).max() < SHIFT_TOL_PX
    # insertion receives the corrected spectra too
    rc = ReconCfg(box=n, pixel_size=px, res_limit=2 * px, normalize=1, split_by_pind=0, mask_radius=0.4 * n * px)
    acc_o = np.zeros(O.accum_floats(n), dtype=np.float32); cnt = np.zeros(2, dtype=np.int64)
    O.insert_batch(acc_o, cnt, rc, "C1", imgs_t, tilted)
    acc = H.Accumulator(n, px, "C1")
    acc.insert(rc, imgs_t, tilted)
    got_acc = acc.download()
    acc.close()
    assert np.linalg.norm(got_acc - acc_o) / np.linalg.norm(acc_o) < 1e-4


def test_local_refinement_matches_oracle(d64, H, O):
    vol, imgs, rows, g, o = d64
    start = synth.perturb_rows(rows, 2.0, 1.0, 2.0)
    c = cfg_for(64, 2.0, global_search=0)
    want, _ = O.refine_batch(o, c, imgs, start)
    got = g.refine(c, imgs, start)
    assert synth.angular_error_deg(want, got).max() < ANG_TOL_DEG
    assert synth.shift_error_px(want, got, 2.0).max() < SHIFT_TOL_PX
    assert np.abs(want[:, 14] - got[:, 14]).max() < 0.01                 # SCORE is 100 x cc
    untouched = [0] + list(range(6, 12)) + list(range(15, 32))
    assert np.array_equal(got[:, untouched], start[:, untouched])


def test_global_grid_search_matches_oracle_exactly(d64, H, O):
    vol, imgs, rows, g, o = d64
    c = cfg_for(64, 2.0, local_refine=0, iters_hit=-1)                   # test hook: the hits stay on the grid
    want, _ = O.refine_batch(o, c, imgs, rows)
    got = g.refine(c, imgs, rows)
    assert synth.angular_error_deg(want, got).max() < 1e-4               # same grid point
    assert np.array_equal(np.round(want[:, 4:6] / 2.0), np.round(got[:, 4:6] / 2.0))   # same integer shift
    assert np.abs(want[:, 14] - got[:, 14]).max() < 0.01


def test_default_mode_global_yes_local_no_refines_the_top_hits(d64, H, O):
    """PYP's default call is global = yes, local = no with 20 hits to refine (frealign.py:3866-3871, :3953): the hits are
    refined (sub-grid poses), the best is kept, nothing continues at the full band."""
    vol, imgs, rows, g, o = d64
    c = cfg_for(64, 2.0, local_refine=0)
    want, counts = O.refine_batch(o, c, imgs, rows)
    got = g.refine(c, imgs, rows)
    lc = g.last_counts()
    assert lc["n_global"] == counts[0] and lc["n_local"] == counts[1] == 20 * 2 * 12 + 1
    assert synth.angular_error_deg(want, got).max() < ANG_TOL_DEG and synth.shift_error_px(want, got, 2.0).max() < SHIFT_TOL_PX
    assert np.abs(want[:, 14] - got[:, 14]).max() < 0.02
    grid = g.refine(cfg_for(64, 2.0, local_refine=0, iters_hit=-1), imgs, rows)
    # sub-grid: closer to the truth than the raw 15 degree grid point, and the hit count matters
    assert np.median(synth.angular_error_deg(got, rows)) < 0.5 * np.median(synth.angular_error_deg(grid, rows))
    one = g.refine(cfg_for(64, 2.0, local_refine=0, top_hits=1), imgs, rows)
    assert g.last_counts()["n_local"] == 2 * 12 + 1 and one.shape == got.shape        # answer 26 reaches the kernel


@pytest.mark.parametrize("step", [10.0, 7.5])
def test_fine_angular_steps_select_top_hits_correctly(H, O, step):
    """n_orient = 14 832 (10 deg) and 34 944 (7.5 deg): the top-K pass needs more LDS than the search tables (it once read
    out of bounds there); the chosen poses must agree with the oracle."""
    vol, imgs, rows = dataset(32, 4, 3.0, 0.2)
    g, o = H.Reference(vol, 16), O.Reference(vol, 16)
    c = cfg_for(32, 3.0, angular_step=step, local_refine=0, iters_hit=-1, top_hits=20)
    want, counts = O.refine_batch(o, c, imgs, rows)
    got = g.refine(c, imgs, rows)
    assert g.last_counts()["n_global"] == counts[0] > 14000
    assert synth.angular_error_deg(want, got).max() < 1e-4
    c2 = cfg_for(32, 3.0, angular_step=step)
    want, _ = O.refine_batch(o, c2, imgs, rows)
    got = g.refine(c2, imgs, rows)
    assert synth.angular_error_deg(want, got).max() < ANG_TOL_DEG and synth.shift_error_px(want, got, 3.0).max() < SHIFT_TOL_PX


@pytest.mark.parametrize("kw", [dict(angular_step=24.0), dict(angular_step=40.0),                       # odd psi counts: no psi / psi + 180 pairing
                                dict(search_range_x=10.0, search_range_y=10.0), dict(search_range_x=0.0, search_range_y=0.0),   # shift windows of 5 and 8 steps
                                dict(search_range_x=4.0, search_range_y=14.0), dict(search_range_x=2.0, search_range_y=2.0),
                                dict(angular_step=24.0, search_range_x=0.0, search_range_y=0.0)])          # odd psi count AND the widest window
def test_search_grid_variants_match_oracle(H, O, kw):
    """Code paths of the grid search the default configuration never takes: an odd number of in-plane angles (every slice
    stored, no conjugate pairing), shift windows wider than 3 steps (512-thread kernel, per-shift wave sums), anisotropic and
    one-step windows."""
    vol, imgs, rows = dataset(64, 8, 2.0, 0.1)
    g, o = H.Reference(vol, 32), O.Reference(vol, 32)
    raw = cfg_for(64, 2.0, local_refine=0, iters_hit=-1, **kw)
    want, counts = O.refine_batch(o, raw, imgs, rows)
    got = g.refine(raw, imgs, rows)
    assert g.last_counts()["n_global"] == counts[0]
    assert synth.angular_error_deg(want, got).max() < 1e-4, kw                          # same grid point ...
    assert np.array_equal(np.round(want[:, 4:6] / 2.0), np.round(got[:, 4:6] / 2.0)), kw   # ... and the same integer shift
    full = cfg_for(64, 2.0, **kw)
    want, _ = O.refine_batch(o, full, imgs, rows)
    got = g.refine(full, imgs, rows)
    assert synth.angular_error_deg(want, got).max() < ANG_TOL_DEG and synth.shift_error_px(want, got, 2.0).max() < SHIFT_TOL_PX, kw


def test_grid_search_at_a_band_between_32_and_64_pixels(H, O):
    """Box 128, search band 45 px: the slice-norm product (k_slice_norms) then covers a partly filled second half of the bank rows,
    k_global a W table of 92 rows; five particles leave a short last block.  Same grid point and integer shift as the oracle."""
    vol, imgs, rows = dataset(128, 5, 1.5, 0.1)
    g, o = H.Reference(vol, 64), O.Reference(vol, 64)
    c = RefineCfg.make(box=128, pixel_size=1.5, mask_radius=0.4 * 128 * 1.5, res_high=1.5 * 128 / 45.0, res_search=1.5 * 128 / 45.0,
                       search_range_x=6.0, search_range_y=6.0, res_signed_cc=30.0, local_refine=0, iters_hit=-1, angular_step=20.0)
    want, counts = O.refine_batch(o, c, imgs, rows)
    got = g.refine(c, imgs, rows)
    assert g.last_counts()["n_global"] == counts[0]
    assert synth.angular_error_deg(want, got).max() < 1e-4
    step = 128.0 / 128.0       # search grid Ns = 128 for B_s = 44: one pixel per step
    assert np.array_equal(np.round(want[:, 4:6] / 1.5 / step), np.round(got[:, 4:6] / 1.5 / step))
    assert np.abs(want[:, 14] - got[:, 14]).max() < 0.01


def test_preprocessing_options_at_256_match_oracle(H, O):
    """Box 256 takes the scratch-free pre-processing path (half spectrum held in registers, statistics folded into the staging pass
    when no mask is applied): its options against the oracle — scores under invert / no normalisation / a focus mask / a beam tilt,
    and insertion (no mask: folded statistics) with inverted contrast."""
    n, px = 256, 1.2
    vol, imgs, rows = dataset(n, 4, px, 0.2)
    g, o = H.Reference(vol, 64), O.Reference(vol, 64)
    C = synth.cistem.COL
    tilted = rows.copy(); tilted[:, C["BEAM_TILT_X"]] = 0.8; tilted[:, C["BEAM_TILT_Y"]] = -0.5
    for kw, rr in ((dict(invert=1), rows), (dict(normalize=0), rows), (dict(focus=(20.0, -15.0, 10.0, 60.0)), rows), (dict(), tilted)):
        c = RefineCfg.make(box=n, pixel_size=px, mask_radius=0.35 * n * px, res_high=px * n / 48.0, global_search=0, local_refine=0, res_signed_cc=30.0, **kw)
        want = O.score_batch(o, c, imgs, rr)
        got = g.refine(c, imgs, rr)[:, 14] / 100.0
        assert np.abs(want - got).max() < 2e-4, kw          # 7 000 samples summed in float32 (both pre-processing paths give the same 1e-4)
    rc = ReconCfg(box=n, pixel_size=px, res_limit=px * n / 60.0, normalize=1, invert=1, split_by_pind=0, mask_radius=0.35 * n * px)
    acc_o = np.zeros(O.accum_floats(n), dtype=np.float32); cnt = np.zeros(2, dtype=np.int64)
    O.insert_batch(acc_o, cnt, rc, "C1", imgs, tilted)
    acc = H.Accumulator(n, px, "C1")
    acc.insert(rc, imgs, tilted)
    got_acc = acc.download()
    acc.close()
    assert np.linalg.norm(got_acc - acc_o) / np.linalg.norm(acc_o) < 1e-4


def test_capped_search_band_and_shift_window_are_reported(H):
    """A search band above 64 Fourier pixels is capped by the grid search, not refused, and ppm_refine_note (the refine3d log) says
    so; a shift range of 0 (PYP's default, "0.0 = mask radius", config/pyp_config.toml:5338-5343) or beyond 8 search-grid steps is
    HONOURED (tiles of the window) and only a range beyond what the search grid can hold without aliasing is reported."""
    vol, imgs, rows = dataset(256, 2, 1.0, 0.2)
    g = H.Reference(vol, 128)
    base = dict(box=256, pixel_size=1.0, mask_radius=82.0, res_high=3.0, res_search=3.0, angular_step=40.0, iters_hit=-1, local_refine=0)
    g.refine(RefineCfg.make(search_range_x=6.0, search_range_y=6.0, **base), imgs, rows)
    assert "band lowered from 85.3 to 64.0" in g.note() and "shift window" not in g.note()
    g.refine(RefineCfg.make(search_range_x=0.0, search_range_y=0.0, **base), imgs, rows)         # +-82 px = 41 steps of 2 px: 5 x 5 tiles
    assert "shift window" not in g.note()
    g.refine(RefineCfg.make(search_range_x=200.0, search_range_y=6.0, **base), imgs, rows)       # 100 steps > Ns / 2 - 1 = 63
    assert "shift window of the grid search: +-126 x +-6 pixels" in g.note() and "asked: 200 pixels" in g.note()
    g.refine(RefineCfg.make(search_range_x=6.0, search_range_y=6.0, res_high=6.0, res_search=6.0, **{k: v for k, v in base.items() if not k.startswith("res_")}), imgs, rows)
    assert g.note() == ""


def test_wide_shift_window_is_searched_in_tiles_like_the_oracle(H, O):
    """Particles up to +-11 px off centre, window +-24 px = 12 search-grid steps (2 x 2 tiles of the kernel's 17 x 17 window): the
    grid search alone (hits left on the grid) gives the oracle's orientation and integer shift for every particle; with PYP's
    range 0 the window is the mask radius (13 steps here)."""
    n, px = 64, 2.0
    vol, stack, rows = synth.make_dataset(n, 10, pixel=px, snr=0.3, shift_sigma=6.0, shift_clip=11.0)
    imgs = stack.numpy()
    assert np.abs(rows[:, 4:6]).max() / px > 8.5
    g, o = H.Reference(vol, n / 2), O.Reference(vol, n / 2)
    for sr in ((24.0, 24.0), (0.0, 30.0), (0.0, 0.0)):
        c = cfg_for(n, px, search_range_x=sr[0] * px if sr[0] else 0.0, search_range_y=sr[1] * px if sr[1] else 0.0, iters_hit=-1, local_refine=0)
        want, cw = O.refine_batch(o, c, imgs, rows)
        got = g.refine(c, imgs, rows)
        assert synth.angular_error_deg(want, got).max() < 1e-3 and synth.shift_error_px(want, got, px).max() < 1e-3, sr
        assert np.abs(want[:, 14] - got[:, 14]).max() < 0.02
    # and refined from there: on the truth
    c = cfg_for(n, px, search_range_x=0.0, search_range_y=0.0)
    want, _ = O.refine_batch(o, c, imgs, rows)
    got = g.refine(c, imgs, rows)
    assert synth.angular_error_deg(want, got).max() < ANG_TOL_DEG and synth.shift_error_px(want, got, px).max() < SHIFT_TOL_PX
    assert np.median(synth.shift_error_px(got, rows, px)) < 0.5 and np.median(synth.angular_error_deg(got, rows)) < 2.0
    # the old +-8-step window could not have found the largest shifts
    narrow = g.refine(cfg_for(n, px, search_range_x=16.0, search_range_y=16.0, iters_hit=-1, local_refine=0), imgs, rows)
    far = np.abs(rows[:, 4:6]).max(axis=1) / px > 8.2              # beyond the narrow window of +-8 px
    assert far.any() and synth.shift_error_px(narrow[far], rows[far], px).min() > 0.5 > np.median(synth.shift_error_px(got[far], rows[far], px))


def test_particle_pairs_of_the_grid_search_do_not_couple(H):
    """k_global works on two particles per block (they share the streamed slice rows): a particle's result must not depend on
    its neighbour, on being the odd one of a short last block, or on where the chunk boundaries fall."""
    vol, imgs, rows = dataset(64, 9, 2.0, 0.1)
    g = H.Reference(vol, 32)
    for c in (cfg_for(64, 2.0), cfg_for(64, 2.0, search_range_x=0.0, search_range_y=0.0)):      # 7 x 7 and 17 x 17 shift windows
        full = g.refine(c, imgs, rows)
        for sel in ([0], [8], [0, 1, 2], [1, 0], [3, 7, 5, 2, 8]):
            assert np.array_equal(g.refine(c, imgs[sel], rows[sel]), full[sel]), sel


def test_two_live_references_with_different_search_grids(H, O):
    """Row twiddles of the grid search belong to the reference handle: alternating calls on two references with different
    boxes / bands must not see each other's tables."""
    va, ia, ra = dataset(64, 6, 2.0, 0.1)
    vb, ib, rb = dataset(48, 6, 3.0, 0.1)
    ga, gb = H.Reference(va, 32), H.Reference(vb, 24)
    ca, cb = cfg_for(64, 2.0), cfg_for(48, 3.0, angular_step=20.0)
    a1 = ga.refine(ca, ia, ra)
    b1 = gb.refine(cb, ib, rb)
    a2 = ga.refine(ca, ia, ra)
    b2 = gb.refine(cb, ib, rb)
    assert np.array_equal(a1, a2) and np.array_equal(b1, b2)
    wa, _ = O.refine_batch(O.Reference(va, 32), ca, ia, ra)
    assert synth.angular_error_deg(wa, a2).max() < ANG_TOL_DEG


@pytest.mark.parametrize("n,px,m,step", [(64, 2.0, 24, 15.0), (128, 1.5, 8, 15.0), (64, 2.0, 6, 20.0), (32, 3.0, 6, 30.0),
                                          (96, 1.5, 6, 15.0), (80, 2.0, 6, 15.0), (48, 3.0, 6, 20.0),    # 2^5 3, 2^4 5, 2^4 3: mixed-radix FFT
                                          (90, 2.0, 6, 20.0),              # 2 3^2 5: not a multiple of 4 (4-byte row fetches)
                                          (56, 2.0, 6, 20.0), (112, 1.5, 4, 20.0), (98, 2.0, 4, 24.0)])   # 2^3 7, 2^4 7, 2 7^2: radix-7 stages (boxes 224 / 336 / 448)
def test_full_refinement_matches_oracle(H, O, n, px, m, step):
    vol, imgs, rows = dataset(n, m, px, 0.1)
    g, o = H.Reference(vol, n / 2), O.Reference(vol, n / 2)
    c = cfg_for(n, px, angular_step=step)
    want, counts = O.refine_batch(o, c, imgs, rows)
    got = g.refine(c, imgs, rows)
    lc = g.last_counts()
    assert lc["n_global"] == counts[0] and lc["n_local"] == counts[1]
    ang, shf = synth.angular_error_deg(want, got), synth.shift_error_px(want, got, px)
    assert ang.max() < ANG_TOL_DEG and shf.max() < SHIFT_TOL_PX


@pytest.mark.parametrize("n,px,m,pad", [(64, 2.0, 12, 2), (48, 3.0, 6, 4), (128, 1.5, 6, 2)])
def test_padded_reference_matches_oracle(H, O, n, px, m, pad):
    """"padding factor" (refine_iblow): the reference transform sampled pad times finer, oracle and kernels alike."""
    vol, imgs, rows = dataset(n, m, px, 0.1)
    g, o = H.Reference(vol, n / 2, pad=pad), O.Reference(vol, n / 2, pad=pad)
    c = cfg_for(n, px, angular_step=20.0)
    want, counts = O.refine_batch(o, c, imgs, rows)
    got = g.refine(c, imgs, rows)
    ang, shf = synth.angular_error_deg(want, got), synth.shift_error_px(want, got, px)
    assert ang.max() < ANG_TOL_DEG and shf.max() < SHIFT_TOL_PX
    assert np.abs(want[:, 14] - got[:, 14]).max() < 0.02
    # scores at the true poses: padding removes most of the interpolation loss on noise-free images
    _, clean, truth = synth.make_dataset(n, m, pixel=px, snr=1e6)
    s1 = H.Reference(vol, n / 2).refine(cfg_for(n, px, global_search=0, local_refine=0), clean.numpy(), truth)[:, 14]
    s2 = g.refine(cfg_for(n, px, global_search=0, local_refine=0), clean.numpy(), truth)[:, 14]
    assert s2.mean() > s1.mean()


def test_defocus_refinement_matches_oracle_and_recovers_offsets(H, O):
    """answers 33 / 34 / 45: defocus offsets scored at the final pose.  Rows start 200 A off their true defocus."""
    n, px, m = 96, 1.5, 12
    vol, stack, truth = synth.make_dataset(n, m, pixel=px, snr=0.5)
    imgs = stack.numpy()
    rows = truth.copy(); rows[:, 6] += 200.0; rows[:, 7] += 200.0
    c = cfg_for(n, px, global_search=0, refine_defocus=1, defocus_range=400.0, defocus_step=50.0, res_high=2.2 * px)
    o, g = O.Reference(vol, n / 2), H.Reference(vol, n / 2)
    want, cw = O.refine_batch(o, c, imgs, rows)
    got = g.refine(c, imgs, rows)
    lc = g.last_counts()
    assert (lc["n_local"], lc["samples_local"]) == (cw[1], cw[2])
    same = np.abs(want[:, 6] - got[:, 6]) < 1e-6
    assert same.mean() >= 0.9 and np.abs(want[:, 6] - got[:, 6]).max() <= 50.0 + 1e-6         # at most one step apart, rarely
    assert np.allclose(got[:, 7] - rows[:, 7], got[:, 6] - rows[:, 6])
    assert np.abs(want[:, 14] - got[:, 14]).max() < 0.05
    err = np.abs(got[:, 6] - truth[:, 6])
    assert np.median(err) <= 50.0 and (err <= 100.0).mean() >= 0.8                           # back to the truth within the grid
    off = g.refine(cfg_for(n, px, global_search=0, res_high=2.2 * px), imgs, rows)
    assert np.array_equal(off[:, 6], rows[:, 6]) and (got[:, 14] >= off[:, 14] - 1e-3).all()


def test_ring_weighted_reference_equals_host_side_filter(H):
    """"use statistics": the radial weight applied on the device while the cube is cut out = the same filter applied to the
    volume with numpy beforehand."""
    n, px, m = 64, 2.0, 10
    vol, stack, rows = synth.make_dataset(n, m, pixel=px, snr=0.3)
    imgs = stack.numpy()
    w = np.clip(1.0 - np.arange(n // 2 + 1) / 40.0, 0.2, 1.0).astype(np.float32)
    f = np.fft.rfftn(vol)
    kz, ky, kx = np.fft.fftfreq(n) * n, np.fft.fftfreq(n) * n, np.arange(n // 2 + 1)
    k = np.sqrt(kz[:, None, None] ** 2 + ky[None, :, None] ** 2 + kx[None, None, :] ** 2)
    volw = np.fft.irfftn(f * np.interp(k, np.arange(n // 2 + 1), w), s=vol.shape, axes=(0, 1, 2)).astype(np.float32)
    c = cfg_for(n, px, global_search=0, local_refine=0)
    a = H.Reference(vol, n / 2, ring_weight=w).refine(c, imgs, rows)[:, 14]
    b = H.Reference(volw, n / 2).refine(c, imgs, rows)[:, 14]
    plain = H.Reference(vol, n / 2).refine(c, imgs, rows)[:, 14]
    # (not bit-identical: the host filter acts before the sinc^2 pre-compensation, the device weight after it)
    assert np.abs(a - b).max() < 0.06 and np.abs(a - plain).max() > 10 * np.abs(a - b).max()


def test_non_finite_inputs_do_not_fault(H):
    """NaN / Inf pixels, a NaN angle and an absurd shift: no device fault, the healthy particles are unaffected."""
    n, px, m = 64, 2.0, 8
    vol, stack, rows = synth.make_dataset(n, m, pixel=px, snr=0.2)
    imgs = stack.numpy().copy()
    good = H.Reference(vol, n / 2).refine(cfg_for(n, px, angular_step=30.0), imgs, rows)
    imgs[1, 10, 10] = np.nan; imgs[2] = np.inf
    bad = rows.copy(); bad[3, 1] = np.nan; bad[4, 4] = 1e30
    out = H.Reference(vol, n / 2).refine(cfg_for(n, px, angular_step=30.0), imgs, bad)
    ok = [0, 5, 6, 7]
    assert synth.angular_error_deg(out[ok], good[ok]).max() < ANG_TOL_DEG
    acc = H.Accumulator(n, px, "C1")
    acc.insert(ReconCfg(box=n, pixel_size=px, res_limit=2 * px, normalize=1, split_by_pind=0, mask_radius=0.4 * n * px), imgs, bad)
    assert acc.counts()[0] + acc.counts()[1] == m


def test_padding_limits_are_loud(H):
    vol = np.zeros((64, 64, 64), np.float32)
    with pytest.raises(Exception):
        H.Reference(vol, 32, pad=3)
    with pytest.raises(Exception):
        H.Reference(np.zeros((384, 384, 384), np.float32), 32, pad=2)


def test_wide_band_grid_search_is_capped_at_64_pixels(H, O):
    """res_search beyond 64 Fourier pixels: the grid search runs at 64 px, the refinement at the full band (72 px)."""
    n, px = 160, 1.0
    vol, imgs, rows = dataset(n, 4, px, 0.1)
    g, o = H.Reference(vol, n / 2), O.Reference(vol, n / 2)
    c = RefineCfg.make(box=n, pixel_size=px, mask_radius=0.4 * n * px, res_high=px * n / 72.0, res_search=px * n / 72.0,
                       angular_step=30.0, search_range_x=6.0, search_range_y=6.0, res_signed_cc=30.0)
    assert O.band_dims(c)["Ns"] == 128
    want, _ = O.refine_batch(o, c, imgs, rows)
    got = g.refine(c, imgs, rows)
    assert synth.angular_error_deg(want, got).max() < ANG_TOL_DEG and synth.shift_error_px(want, got, px).max() < SHIFT_TOL_PX


def test_odd_psi_count_and_wide_shift_window(d64, H, O):
    vol, imgs, rows, g, o = d64
    c = cfg_for(64, 2.0, angular_step=24.0, search_range_x=0.0, search_range_y=10.0)    # n_psi = 15 (no conjugate pairing), RSx = 8
    want, _ = O.refine_batch(o, c, imgs[:6], rows[:6])
    got = g.refine(c, imgs[:6], rows[:6])
    assert synth.angular_error_deg(want, got).max() < ANG_TOL_DEG
    assert synth.shift_error_px(want, got, 2.0).max() < SHIFT_TOL_PX


def test_separate_search_mask_and_frozen_parameters(d64, H, O):
    vol, imgs, rows, g, o = d64
    c = cfg_for(64, 2.0, search_mask_radius=1.5 * 0.32 * 64 * 2.0, refine_x=0, refine_y=0)
    want, _ = O.refine_batch(o, c, imgs[:6], rows[:6])
    got = g.refine(c, imgs[:6], rows[:6])
    assert synth.angular_error_deg(want, got).max() < ANG_TOL_DEG
    assert np.allclose(want[:, 4:6], got[:, 4:6], atol=1e-6)


@pytest.mark.parametrize("flags", [dict(refine_phi=0), dict(refine_theta=0), dict(refine_psi=0, refine_x=0), dict(refine_theta=0, refine_phi=0)])
def test_partial_refine_flags_match_oracle(d64, H, O, flags):
    """theta-only / phi-only use Euler-angle steps instead of the image-frame tilts; frozen parameters stay put."""
    vol, imgs, rows, g, o = d64
    start = synth.perturb_rows(rows, 1.5, 0.7, 2.0, seed=11)
    c = cfg_for(64, 2.0, global_search=0, **flags)
    want, _ = O.refine_batch(o, c, imgs[:8], start[:8])
    got = g.refine(c, imgs[:8], start[:8])
    assert synth.angular_error_deg(want, got).max() < ANG_TOL_DEG
    assert synth.shift_error_px(want, got, 2.0).max() < SHIFT_TOL_PX
    if not c.refine_x:
        assert np.allclose(got[:, 4], start[:8, 4], atol=1e-6)


@pytest.mark.parametrize("sym", ["C3", "D2", "O"])
def test_symmetry_restricted_grid_matches_oracle(d64, H, O, sym):
    vol, imgs, rows, g, o = d64
    c = cfg_for(64, 2.0, symmetry=sym)
    want, cw = O.refine_batch(o, c, imgs[:8], rows[:8])
    got = g.refine(c, imgs[:8], rows[:8])
    assert g.last_counts()["n_global"] == cw[0] < 4416
    assert synth.angular_error_deg(want, got).max() < ANG_TOL_DEG


def test_frequency_marching_can_be_switched_off(d64, H, O):
    vol, imgs, rows, g, o = d64
    c = cfg_for(64, 2.0, band_factor=-1.0)
    want, cw = O.refine_batch(o, c, imgs[:6], rows[:6])
    got = g.refine(c, imgs[:6], rows[:6])
    assert synth.angular_error_deg(want, got).max() < ANG_TOL_DEG
    assert g.last_counts()["samples_local"] == cw[2]


def test_device_resident_stack_equals_host_stack(d64, H):
    import torch
    vol, imgs, rows, g, o = d64
    c = cfg_for(64, 2.0)
    a = g.refine(c, imgs[:8], rows[:8])
    b = g.refine(c, torch.as_tensor(imgs[:8]).cuda(), rows[:8])
    # ring sums are accumulated per wave and combined in a fixed order: runs are bit-identical
    assert np.array_equal(a, b)
    assert np.array_equal(a, g.refine(c, imgs[:8], rows[:8]))


def test_tap_addresses_from_lds_tables_equal_the_arithmetic_path(d64, H, monkeypatch):
    """k_local reads its tap addresses from LDS tables (ppm_dev.h, cube_tab_fill); PPM_LOCAL_TABLES=0 computes them per gather.
    Same taps, same weights up to the last bit of a fraction: the refined rows agree far inside the parity tolerance."""
    vol, imgs, rows, g, o = d64
    c = cfg_for(64, 2.0)
    tab = g.refine(c, imgs[:8], rows[:8])
    monkeypatch.setenv("PPM_LOCAL_TABLES", "0")
    ari = g.refine(c, imgs[:8], rows[:8])
    monkeypatch.delenv("PPM_LOCAL_TABLES")
    assert np.allclose(tab[:, 1:6], ari[:, 1:6], atol=2e-3)             # angles (deg) and shifts (A)
    assert np.allclose(tab, ari, rtol=1e-4, atol=2e-3)


def test_chunked_batches_and_empty_input(d64, H, monkeypatch):
    """Ragged chunking (n not a multiple of the chunk) gives the same rows as one chunk; zero particles is a no-op."""
    vol, imgs, rows, g, o = d64
    c = cfg_for(64, 2.0)
    whole = g.refine(c, imgs[:11], rows[:11])
    monkeypatch.setenv("PPM_CHUNK", "4")
    parts = g.refine(c, imgs[:11], rows[:11])
    monkeypatch.delenv("PPM_CHUNK")
    assert np.array_equal(whole, parts)                                 # a particle's result does not depend on its chunk
    cd = cfg_for(64, 2.0, refine_defocus=1, defocus_range=300.0, defocus_step=100.0)
    whole = g.refine(cd, imgs[:11], rows[:11])
    monkeypatch.setenv("PPM_CHUNK", "4")
    parts = g.refine(cd, imgs[:11], rows[:11])
    monkeypatch.delenv("PPM_CHUNK")
    assert np.array_equal(whole, parts)
    assert g.refine(c, np.zeros((0, 64, 64), np.float32), np.zeros((0, 32))).shape == (0, 32)


def test_errors_are_loud(d64, H):
    from pyp_amd import lib
    vol, imgs, rows, g, o = d64
    with pytest.raises(lib.PpmError, match="ERROR"):
        g.refine(cfg_for(128, 2.0), np.zeros((1, 128, 128), np.float32), rows[:1])          # box mismatch
    with pytest.raises(lib.PpmError, match="ERROR"):
        g.refine(RefineCfg.make(box=64, pixel_size=2.0, mask_radius=40, res_high=-1.0), imgs[:1], rows[:1])
    with pytest.raises(lib.PpmError, match="ERROR"):
        H.Accumulator(64, 2.0, "Q5")
    with pytest.raises(ValueError):
        H.Reference(np.zeros((32, 32, 16), np.float32))


@pytest.mark.parametrize("sym", ["C1", "D2", "O"])
def test_insertion_and_finalise_match_oracle(H, O, sym):
    n, px, m = 64, 2.0, 40
    vol, imgs, rows = dataset(n, m, px, 0.2)
    rows[:, 14] = np.linspace(5, 35, m)
    rows[3, 11] = 0.0                                                   # one rejected particle
    rc = ReconCfg(box=n, pixel_size=px, res_limit=2 * px, score_weight_bfactor=2.0, score_average=20.0, score_threshold=0.0,
                  normalize=1, invert=0, split_by_pind=1, mask_radius=0.4 * n * px)
    acc = np.zeros(O.accum_floats(n), dtype=np.float32)
    counts = np.zeros(2, dtype=np.int64)
    O.insert_batch(acc, counts, rc, sym, imgs, rows)
    ga = H.Accumulator(n, px, sym)
    ga.insert(rc, imgs[:25], rows[:25])
    ga.insert(rc, imgs[25:], rows[25:])                                 # ragged second batch
    assert ga.counts() == list(counts)
    got = ga.download()
    assert np.linalg.norm(got - acc) / np.linalg.norm(acc) < 1e-4       # float atomics: order-dependent round-off
    fc = FinalCfg(molecular_mass_kda=300.0, inner_radius=0.0, outer_radius=0.45 * n * px, mask_falloff=0.0)
    w1, w2, wf, ws = O.finalize(acc, n, px, fc)
    g1, g2, gf, gs = ga.finalize(fc)
    for a, b in ((w1, g1), (w2, g2), (wf, gf)):
        assert np.linalg.norm(a - b) / np.linalg.norm(a) < 1e-4
    assert np.abs(ws[:, 3:5] - gs[:, 3:5]).max() < 1e-4                  # FSC, part-FSC


def test_dose_weighted_insertion_matches_oracle(H, O):
    """Data-driven dose weighting (the five-line reconstruct3d answer, frealign.py:1731-1753): per-exposure attenuation by TIND."""
    n, px, m = 64, 2.0, 36
    vol, imgs, rows = dataset(n, m, px, 0.2)
    rows[:, 27] = np.arange(m) % 6
    rows[:, 14] = 30.0 - 4.0 * rows[:, 27]
    from pyp_amd import dose
    q = dose.normalised(dose.compute_global_weights(rows))
    rc = ReconCfg(box=n, pixel_size=px, res_limit=2 * px, normalize=1, invert=0, split_by_pind=0, mask_radius=0.4 * n * px)
    rc.set_dose_weights(q, 4.0, 0.75)
    acc = np.zeros(O.accum_floats(n), dtype=np.float32)
    counts = np.zeros(2, dtype=np.int64)
    O.insert_batch(acc, counts, rc, "C1", imgs, rows)
    ga = H.Accumulator(n, px, "C1")
    ga.insert(rc, imgs, rows)
    got = ga.download()
    assert ga.counts() == list(counts) and np.linalg.norm(got - acc) / np.linalg.norm(acc) < 1e-4
    plain = H.Accumulator(n, px, "C1")
    plain.insert(ReconCfg(box=n, pixel_size=px, res_limit=2 * px, normalize=1, invert=0, split_by_pind=0, mask_radius=0.4 * n * px), imgs, rows)
    assert np.linalg.norm(plain.download() - acc) / np.linalg.norm(acc) > 0.05        # the weighting is not a no-op


@pytest.mark.parametrize("n", [96, 112])
def test_insertion_non_power_of_two_box(H, O, n):
    px, m = 1.5, 16
    vol, imgs, rows = dataset(n, m, px, 0.2)
    rc = ReconCfg(box=n, pixel_size=px, res_limit=2 * px, normalize=1, split_by_pind=0, mask_radius=0.4 * n * px)
    acc = np.zeros(O.accum_floats(n), dtype=np.float32)
    counts = np.zeros(2, dtype=np.int64)
    O.insert_batch(acc, counts, rc, "C2", imgs, rows)
    ga = H.Accumulator(n, px, "C2")
    ga.insert(rc, imgs, rows)
    assert np.linalg.norm(ga.download() - acc) / np.linalg.norm(acc) < 1e-4
    fc = FinalCfg(molecular_mass_kda=200.0, inner_radius=0.0, outer_radius=0.45 * n * px, mask_falloff=0.0)
    w = O.finalize(acc, n, px, fc)
    g = ga.finalize(fc)
    for a, b in zip(w[:3], g[:3]):
        assert np.linalg.norm(a - b) / np.linalg.norm(a) < 1e-4


@pytest.mark.parametrize("n,px,m,sym,minp", [(256, 1.0, 24, "C1", "4"), (160, 1.5, 12, "C2", None), (128, 2.0, 600, "C1", "64"),
                                             (50, 2.5, 10, "C1", None), (150, 1.5, 8, "C3", "2")])
def test_insertion_large_boxes_bricks_and_slices(H, O, monkeypatch, n, px, m, sym, minp):
    """Boxes >= 128 use 16^3-voxel bricks, 1024-thread blocks and (with more particles than PPM_BRICK_MINP per slice)
    several particle slices per brick whose partial bricks are summed in global memory; 256 also takes the 16 x 16
    register FFT of the pre-processing kernel."""
    if minp:
        monkeypatch.setenv("PPM_BRICK_MINP", minp)
    vol, imgs, rows = dataset(n, min(m, 24), px, 0.2)
    if m > imgs.shape[0]:                                               # many cheap particles: repeat with shifted halves
        rep = (m + imgs.shape[0] - 1) // imgs.shape[0]
        imgs = np.concatenate([imgs] * rep)[:m]; rows = np.concatenate([rows] * rep)[:m].copy()
        rows[:, 0] = np.arange(1, m + 1); rows[:, 26] = np.arange(m)
    rc = ReconCfg(box=n, pixel_size=px, res_limit=2 * px, normalize=1, split_by_pind=0, mask_radius=0.4 * n * px)
    acc = np.zeros(O.accum_floats(n), dtype=np.float32)
    counts = np.zeros(2, dtype=np.int64)
    O.insert_batch(acc, counts, rc, sym, imgs, rows)
    ga = H.Accumulator(n, px, sym)
    ga.insert(rc, imgs, rows)
    assert ga.counts() == list(counts)
    assert np.linalg.norm(ga.download() - acc) / np.linalg.norm(acc) < 1e-4


def test_insertion_in_several_chunks_from_host_and_device(H, O, monkeypatch):
    """PPM_CHUNK forces 5 chunks (double-buffered host uploads, a ragged last chunk, brick items rebuilt for it)."""
    import torch
    monkeypatch.setenv("PPM_CHUNK", "160")
    monkeypatch.setenv("PPM_BRICK_MINP", "32")
    n, px, m = 128, 2.0, 700
    vol, imgs, rows = dataset(n, 20, px, 0.2)
    imgs = np.concatenate([imgs] * 35)[:m]; rows = np.concatenate([rows] * 35)[:m].copy()
    rows[:, 0] = np.arange(1, m + 1)
    rc = ReconCfg(box=n, pixel_size=px, res_limit=2 * px, normalize=1, split_by_pind=0, mask_radius=0.4 * n * px)
    acc = np.zeros(O.accum_floats(n), dtype=np.float32)
    counts = np.zeros(2, dtype=np.int64)
    O.insert_batch(acc, counts, rc, "C1", imgs, rows)
    for stack in (imgs, torch.from_numpy(imgs).cuda()):
        ga = H.Accumulator(n, px, "C1")
        ga.insert(rc, stack, rows)
        assert ga.counts() == list(counts)
        assert np.linalg.norm(ga.download() - acc) / np.linalg.norm(acc) < 1e-4


def test_external_accumulator_tensor_and_reduce(H, O):
    """The accumulator can live in a caller-allocated torch tensor (what RCCL reduces in place)."""
    import torch
    from pyp_amd import dist as pdist
    n, px = 32, 3.0
    vol, imgs, rows = dataset(n, 12, px, 0.2)
    rc = ReconCfg(box=n, pixel_size=px, res_limit=2 * px, normalize=1, mask_radius=0.4 * n * px)
    own = H.Accumulator(n, px)
    own.insert(rc, imgs, rows)
    t = torch.zeros(own.nfloats, dtype=torch.float32, device="cuda")
    ext = H.Accumulator(n, px, ext_tensor=t)
    ext.insert(rc, imgs, rows)
    a, b = own.download(), t.cpu().numpy()
    assert np.linalg.norm(a - b) / np.linalg.norm(a) < 1e-5
    t2, c2 = pdist.reduce_accumulators(t, ext.counts())           # single process: identity
    assert c2 == ext.counts() and t2.data_ptr() == t.data_ptr()
    with pytest.raises(ValueError):
        H.Accumulator(n, px, ext_tensor=torch.zeros(7, device="cuda"))


def test_accum_reduce_through_the_c_abi_on_one_rank(H, O):
    """ppm_comm_unique_id / ppm_comm_create / ppm_accum_reduce (include/ppm.h): librccl is opened on first use and
    a one-rank communicator sums the accumulator with itself = identity, as all-reduce and as a reduce to root 0; counters
    travel along.  More than one rank needs one GPU per rank (RCCL refuses two ranks on a device): world-size-2 logic is
    covered with gloo in tests/test_dist_cpu.py."""
    n, px = 32, 3.0
    vol, imgs, rows = dataset(n, 12, px, 0.2)
    rc = ReconCfg(box=n, pixel_size=px, res_limit=2 * px, normalize=1, mask_radius=0.4 * n * px)
    acc = H.Accumulator(n, px)
    acc.insert(rc, imgs, rows)
    before, counts = acc.download(), acc.counts()
    uid = H.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    comm = H.make_comm(1, 0, uid)
    try:
        assert H.comm_count(comm) == 1     # what RCCL itself says about the communicator (ppm_comm_count = ncclCommCount)
        acc.reduce(comm)                   # all-reduce
        assert np.array_equal(acc.download(), before) and acc.counts() == counts and sum(counts) == 12
        acc.reduce(comm, root=0)
        assert np.array_equal(acc.download(), before) and acc.counts() == counts
    finally:
        H.destroy_comm(comm)
    with pytest.raises(ValueError):
        H.make_comm(1, 0, b"short")


def test_largest_box_512_properties(H):
    """N = 512 (the largest supported box): no oracle at this size; the true pose must outscore perturbed poses and a
    local refinement from a perturbed start must move towards it."""
    n, px, m = 512, 1.0, 4
    vol = synth.phantom(n, n_blobs=6, n_atoms=4000)
    _, stack, rows = synth.make_dataset(n, m, pixel=px, snr=0.1, vol=vol, device="cuda", batch=2)
    g = H.Reference(vol, 128)
    c = RefineCfg.make(box=n, pixel_size=px, mask_radius=0.32 * n * px, res_high=px * n / 100.0, global_search=0, local_refine=0,
                       res_signed_cc=30.0)
    s_true = g.refine(c, stack, rows)[:, 14]
    pert = synth.perturb_rows(rows, 1.0, 1.0, px, seed=5)
    s_pert = g.refine(c, stack, pert)[:, 14]
    assert (s_true > s_pert).all() and (s_true > 5).all()
    c.local_refine = 1
    out = g.refine(c, stack, pert)
    assert np.median(synth.angular_error_deg(out, rows)) < np.median(synth.angular_error_deg(pert, rows))
    assert (out[:, 14] >= s_pert - 1e-3).all()
    # the full band of the largest box: the per-wave ring tables alone take 54 KB of LDS, so k_local must fall back from its LDS address
    # tables to arithmetic (a launch may ask for 64 KB) - and still rank the true pose first
    gf = H.Reference(vol, 256)
    cf = RefineCfg.make(box=n, pixel_size=px, mask_radius=0.32 * n * px, res_high=px * n / 255.0, global_search=0, local_refine=0, res_signed_cc=30.0)
    assert (gf.refine(cf, stack, rows)[:, 14] > gf.refine(cf, stack, pert)[:, 14]).all()


def test_accumulator_sum_is_linear(H, O):
    """merge = sum of dumps: inserting two halves of a stack separately and adding equals inserting all (local_merge3d)."""
    n, px = 32, 3.0
    vol, imgs, rows = dataset(n, 20, px, 0.2)
    rc = ReconCfg(box=n, pixel_size=px, res_limit=2 * px, normalize=1, mask_radius=0.4 * n * px)
    a, b, c = H.Accumulator(n, px), H.Accumulator(n, px), H.Accumulator(n, px)
    a.insert(rc, imgs[:10], rows[:10]); b.insert(rc, imgs[10:], rows[10:]); c.insert(rc, imgs, rows)
    a.add(b.download())
    x, y = a.download(), c.download()
    assert np.linalg.norm(x - y) / np.linalg.norm(y) < 1e-5


def test_baseline_size_256_matches_oracle(H, O):
    """BASELINE.json configs[1] geometry (256^2 box, 15 deg, band 64 px, 20 hits) on 32 particles against the oracle itself (OpenMP over
    the particles on the GPU box's cores; bench.py's parity_vs_oracle does the same on 256 particles of the timed stack)."""
    n, px, m = 256, 1.0, 32
    vol, stack, rows = synth.make_dataset(n, m, pixel=px, snr=0.05)
    imgs = stack.numpy()
    c = RefineCfg.make(box=n, pixel_size=px, mask_radius=0.32 * n * px, res_high=4.0, res_search=4.0, search_range_x=6.0,
                       search_range_y=6.0, res_signed_cc=30.0)
    want, cw = O.refine_batch(O.Reference(vol, n / 2), c, imgs, rows)
    g = H.Reference(vol, n / 2)
    got = g.refine(c, imgs, rows)
    lc = g.last_counts()
    assert (lc["n_global"], lc["n_local"], lc["samples_local"]) == (cw[0], cw[1], cw[2]) and cw[0] == 4416
    assert synth.angular_error_deg(want, got).max() < ANG_TOL_DEG and synth.shift_error_px(want, got, px).max() < SHIFT_TOL_PX
    assert np.abs(want[:, 14] - got[:, 14]).max() < 0.02


def test_full_size_properties_256(H):
    """BASELINE.json size (256^2, band 64 px): size-independent properties instead of the (slow) oracle."""
    n, px, m = 256, 1.0, 48
    vol, stack, rows = synth.make_dataset(n, m, pixel=px, snr=0.05)
    imgs = stack.numpy()
    g = H.Reference(vol, n / 2)
    c = RefineCfg.make(box=n, pixel_size=px, mask_radius=0.32 * n * px, res_high=4.0, res_search=4.0, search_range_x=6.0,
                       search_range_y=6.0, res_signed_cc=30.0)
    out = g.refine(c, imgs, rows)
    ang = synth.angular_error_deg(out, rows)
    assert np.median(ang) < 1.5 and (ang < 5).mean() > 0.9                # recovers the true poses from scratch
    # idempotence: a local refinement started at the refined poses stays there
    c2 = RefineCfg.make(box=n, pixel_size=px, mask_radius=0.32 * n * px, res_high=4.0, global_search=0, res_signed_cc=30.0,
                        local_angle_step=0.2, local_shift_step=0.2)
    again = g.refine(c2, imgs, out)
    assert np.median(synth.angular_error_deg(again, out)) < 0.1
    assert (again[:, 14] >= out[:, 14] - 1e-3).all()
    # in-plane rotation covariance: rotating the image by 90 deg (exact on the grid) adds 90 deg to psi
    rot = np.ascontiguousarray(np.rot90(imgs[:8], k=1, axes=(1, 2)))
    # our rot90 of an even box moves the centre by one pixel; compare poses up to that shift
    o2 = g.refine(c, rot, rows[:8])
    d = synth.angular_error_deg(o2, out[:8])
    assert np.median(np.abs(d - 90.0)) < 2.0


def test_priors_restrain_the_search_like_the_oracle(d64, H, O):
    """Answer 7 "use priors" (frealign.py:3841-3844, :3927; include/ppm.h ppm_refine_cfg.use_priors): GPU = oracle with the
    restraint switched on; a flat prior changes nothing; a tight prior around the start keeps the poses there."""
    vol, imgs, rows, g, o = d64
    n, px = 64, 2.0
    start = rows.copy()
    rng = np.random.default_rng(9)
    start[:, 1:4] += rng.normal(0, 1.5, (len(rows), 3)); start[:, 4:6] += rng.normal(0, 1.0, (len(rows), 2)) * px
    base = dict(box=n, pixel_size=px, mask_radius=0.4 * n * px, res_high=px * n / 24, res_signed_cc=30.0, global_search=0)
    free = g.refine(RefineCfg.make(**base), imgs, start)
    mean = [180.0, 90.0, 180.0, 0.0, 0.0]
    flat = g.refine(RefineCfg.make(priors=(mean, [1e12] * 5), **base), imgs, start)
    assert synth.angular_error_deg(free, flat).max() < 1e-3 and synth.shift_error_px(free, flat, px).max() < 1e-3
    # per-particle tight priors are not what the file carries (one mean for the data set): take one particle and restrain it to its start
    for j in (0, 3):
        pri = ([start[j, 1], start[j, 2], start[j, 3], start[j, 4], start[j, 5]], [1e-3, 1e-3, 1e-3, 1e-3, 1e-3])
        cfg = RefineCfg.make(priors=pri, **base)
        want, _ = O.refine_batch(o, cfg, imgs[j:j + 1], start[j:j + 1])
        got = g.refine(cfg, imgs[j:j + 1], start[j:j + 1])
        assert synth.angular_error_deg(want, got).max() < 0.1 and synth.shift_error_px(want, got, px).max() < 0.5
        assert synth.angular_error_deg(got, start[j:j + 1]).max() < 0.25 * synth.angular_error_deg(free[j:j + 1], start[j:j + 1]).max()
        assert abs(want[0, 14] - got[0, 14]) < 0.05                   # SCORE is the data term alone, at the restrained pose
    # a realistic data-set prior (broad angles, shifts within a few Angstrom): GPU = oracle on all particles
    pri = ([180.0, 90.0, 180.0, 0.0, 0.0], [1.0e4, 2.5e3, 1.0e4, 4.0, 4.0])
    cfg = RefineCfg.make(priors=pri, **base)
    want, _ = O.refine_batch(o, cfg, imgs, start)
    got = g.refine(cfg, imgs, start)
    assert synth.angular_error_deg(want, got).max() < 0.1 and synth.shift_error_px(want, got, px).max() < 0.5
    assert np.abs(want[:, 14] - got[:, 14]).max() < 0.05


def test_two_handles_driven_from_two_threads_equal_serial_runs(H, O):
    """include/ppm.h "Thread-compatible per handle": two references (two classes) refined concurrently from two threads, and an
    insertion running next to them, give bit for bit what the same calls give one after the other."""
    import threading
    n, px = 64, 2.0
    vol, imgs, rows = dataset(n, 64, px, 0.2)
    vol2 = np.ascontiguousarray(vol[::-1, :, :])                     # a second, different reference
    cfg = cfg_for(n, px)
    rc = ReconCfg(box=n, pixel_size=px, res_limit=2 * px, normalize=1, mask_radius=0.4 * n * px)
    a, b = H.Reference(vol, n / 2), H.Reference(vol2, n / 2)
    acc = H.Accumulator(n, px)
    want_a, want_b = a.refine(cfg, imgs, rows), b.refine(cfg, imgs, rows)
    acc.insert(rc, imgs, rows)
    want_acc = acc.download()
    acc2 = H.Accumulator(n, px)
    out, errs = {}, []

    def run(key, fn):
        try:
            for _ in range(3):
                out[key] = fn()
        except Exception as e:          # noqa: BLE001
            errs.append(e)
    def ins():
        acc2.set_counts(0, 0)
        t = H.Accumulator(n, px)
        t.insert(rc, imgs, rows)
        r = t.download(); t.close()
        return r
    ts = [threading.Thread(target=run, args=("a", lambda: a.refine(cfg, imgs, rows))),
          threading.Thread(target=run, args=("b", lambda: b.refine(cfg, imgs, rows))),
          threading.Thread(target=run, args=("acc", ins))]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert not errs, errs
    assert np.array_equal(out["a"], want_a) and np.array_equal(out["b"], want_b)
    assert np.linalg.norm(out["acc"] - want_acc) / np.linalg.norm(want_acc) < 1e-6       # global float atomics of the halo write-back
    assert not np.array_equal(want_a[:, 1:4], want_b[:, 1:4])


@pytest.mark.parametrize("kw", [dict(global_search=0), dict(global_search=1, local_refine=0),
                                dict(global_search=0, refine_defocus=1, defocus_range=200.0, defocus_step=50.0)])
def test_classification_limit_moves_logp_and_sigma_like_the_oracle(H, O, kw):
    """Answer 22 (class_rhcls, frealign.py:3945): LOGP / SIGMA over res_low .. res_classification at the final pose (and the final
    CTF when the defocus is refined), SCORE and the pose as without it; evaluation counts include the extra sweep."""
    n, px, m = 64, 2.0, 10
    vol, imgs, rows = dataset(n, m, px, 0.2)
    rows = rows.copy(); rows[:, 6] += 100.0; rows[:, 7] += 100.0
    o, g = O.Reference(vol, n / 2), H.Reference(vol, n / 2)
    c0 = cfg_for(n, px, **kw)
    c1 = cfg_for(n, px, res_classification=px * n / 12, **kw)
    w0, _ = O.refine_batch(o, c0, imgs, rows)
    w1, cw = O.refine_batch(o, c1, imgs, rows)
    g0, g1 = g.refine(c0, imgs, rows), g.refine(c1, imgs, rows)
    lc = g.last_counts()
    assert (lc["n_local"], lc["samples_local"]) == (cw[1], cw[2])
    keep = [c for c in range(32) if c not in (12, 13)]
    assert np.array_equal(g0[:, keep], g1[:, keep])                      # only LOGP (12) and SIGMA (13) move
    assert not np.allclose(g0[:, 12], g1[:, 12])
    assert synth.angular_error_deg(w1, g1).max() < ANG_TOL_DEG and synth.shift_error_px(w1, g1, px).max() < SHIFT_TOL_PX
    assert np.abs(w1[:, 13] - g1[:, 13]).max() < 2e-4 and np.abs(w1[:, 12] - g1[:, 12]).max() < 2e-3 * np.abs(w1[:, 12]).max()
    assert np.abs(w0[:, 13] - g0[:, 13]).max() < 2e-4
    # a limit at or beyond res_high is the full band
    g2 = g.refine(cfg_for(n, px, res_classification=1.0, **kw), imgs, rows)
    assert np.array_equal(g2, g0)
