"""Global search over WIDE shift windows (k_gfft, pyp_amd/csrc/ppm_gfft.h): PYP's default call sends search range X / Y = 0 = the
mask radius (frealign.py:3954-3957, config/pyp_config.toml:5338-5350).  The HIP path (through the C ABI) against the oracle's
zero-filled inverse transform (oracle/ppm_oracle.c ccf_peak, mode 0) on identical seeded inputs: the same grid orientation and the
same integer shift for every particle, and against the library's own tiled register-window kernel.  Run on the GPU box: pytest -m gpu"""
import numpy as np
import pytest

from pyp_amd import synth
from pyp_amd.abi import RefineCfg

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


def dataset(n, m, px, snr, shift_sigma=None):
    vol, stack, rows = synth.make_dataset(n, m, pixel=px, snr=snr)
    return vol, stack.numpy(), rows


def raw_cfg(n, px, band_px, search_px, **kw):
    """Grid search only: the hits stay at their grid points (test hook iters_hit = -1), no local refinement."""
    base = dict(box=n, pixel_size=px, mask_radius=0.4 * n * px, res_high=px * n / band_px, res_search=px * n / search_px,
                search_range_x=0.0, search_range_y=0.0, res_signed_cc=30.0, local_refine=0, iters_hit=-1)
    base.update(kw)
    return RefineCfg.make(**base)


def same_grid_point_and_shift(want, got, px, step):
    assert synth.angular_error_deg(want, got).max() < 1e-4
    assert np.array_equal(np.round(want[:, 4:6] / px / step), np.round(got[:, 4:6] / px / step))
    assert np.abs(want[:, 14] - got[:, 14]).max() < 0.01                 # SCORE is 100 x cc


@pytest.mark.parametrize("kw", [dict(), dict(angular_step=24.0), dict(search_range_x=16.0, search_range_y=26.0),
                                dict(search_range_x=30.0, search_range_y=14.0, angular_step=20.0), dict(symmetry="D2")])
def test_window_of_the_mask_radius_at_64_matches_oracle_mode_0(H, O, kw):
    """Box 64, search band 10 px: search grid of 32 points (2-pixel steps), window +-13 steps for range 0 = mask radius; odd in-plane
    counts (no psi / psi + 180 pairing), anisotropic windows, a symmetric grid."""
    n, px = 64, 2.0
    vol, imgs, rows = dataset(n, 10, px, 0.1)
    c = raw_cfg(n, px, 24.0, 10.24, **kw)
    d = O.band_dims(c)
    assert d["Ns"] == 32 and max(d["RSx"], d["RSy"]) >= 7
    want, counts = O.refine_batch(O.Reference(vol, n / 2), c, imgs, rows, ccf_mode=0)
    g = H.Reference(vol, n / 2)
    got = g.refine(c, imgs, rows)
    assert g.last_counts()["n_global"] == counts[0]
    same_grid_point_and_shift(want, got, px, d["step"])


def test_particles_far_off_centre_are_found_inside_the_wide_window(H, O):
    """What the wide window is for: particles displaced by up to 10 pixels (5 search-grid steps) are found by the grid search; a
    window of +-2 steps cannot hold them."""
    n, px = 64, 2.0
    vol, stack, rows = synth.make_dataset(n, 12, pixel=px, snr=0.2)
    rng = np.random.default_rng(5)
    big = rows.copy()
    big[:, 4:6] = rng.integers(-5, 6, (len(rows), 2)) * 2.0 * px           # whole search-grid steps
    dx = np.round((big[:, 4:6] - rows[:, 4:6]) / px).astype(int)
    big[:, 4:6] = rows[:, 4:6] + dx * px                                    # the images move by whole pixels
    imgs = np.stack([np.roll(np.roll(stack.numpy()[i], dx[i, 0], axis=1), dx[i, 1], axis=0) for i in range(len(rows))])
    c = raw_cfg(n, px, 24.0, 10.24)
    g = H.Reference(vol, n / 2)
    got = g.refine(c, imgs, big)
    want, _ = O.refine_batch(O.Reference(vol, n / 2), c, imgs, big, ccf_mode=0)
    same_grid_point_and_shift(want, got, px, 2.0)
    err = np.abs(got[:, 4:6] - big[:, 4:6]).max(axis=1) / px
    assert np.median(err) <= 2.0                                            # within one search-grid step of the truth
    narrow = g.refine(raw_cfg(n, px, 24.0, 10.24, search_range_x=2 * 2.0 * px, search_range_y=2 * 2.0 * px), imgs, big)
    far = np.abs(big[:, 4:6]).max(axis=1) / px >= 8
    assert far.any() and (np.abs(narrow[far, 4:6] - big[far, 4:6]).max(axis=1) / px > 2.0).all()


def test_transform_and_tiled_register_windows_agree(H, O, monkeypatch):
    """The two kernels of the grid search on the same inputs: the full-window transform (forced for a narrow window too) and the
    tiled register windows (forced for a wide one) pick the same orientation and shift."""
    n, px = 64, 2.0
    vol, imgs, rows = dataset(n, 16, px, 0.1)
    g = H.Reference(vol, n / 2)
    for kw in (dict(search_range_x=6 * px, search_range_y=6 * px), dict(), dict(search_range_x=4.0, search_range_y=20.0)):
        c = raw_cfg(n, px, 24.0, 10.24, **kw)
        monkeypatch.setenv("PPM_GLOBAL_PATH", "tiles")
        a = g.refine(c, imgs, rows)
        monkeypatch.setenv("PPM_GLOBAL_PATH", "fft")
        b = g.refine(c, imgs, rows)
        monkeypatch.delenv("PPM_GLOBAL_PATH")
        same_grid_point_and_shift(a, b, px, 2.0)
    # and the whole default call (hits refined, best continued) through either kernel
    c = RefineCfg.make(box=n, pixel_size=px, mask_radius=0.4 * n * px, res_high=px * n / 24.0, res_search=px * n / 10.24, res_signed_cc=30.0)
    monkeypatch.setenv("PPM_GLOBAL_PATH", "tiles")
    a = g.refine(c, imgs, rows)
    monkeypatch.delenv("PPM_GLOBAL_PATH")
    b = g.refine(c, imgs, rows)
    assert synth.angular_error_deg(a, b).max() < 1e-3 and synth.shift_error_px(a, b, px).max() < 1e-3


@pytest.mark.parametrize("n,px,band,search,ns", [(32, 3.0, 12.0, 6.0, 16), (128, 1.5, 40.0, 25.6, 64), (128, 1.5, 48.0, 45.0, 128)])
def test_every_search_grid_size_matches_oracle_mode_0(H, O, n, px, band, search, ns):
    """Search grids of 16, 64 and 128 points (8, 2 and 1 slices per pass of the kernel; the 128-point grid at a band that leaves the
    upper rows of the bank empty), window = the mask radius."""
    vol, imgs, rows = dataset(n, 5, px, 0.15)
    c = raw_cfg(n, px, band, search, angular_step=20.0)
    d = O.band_dims(c)
    assert d["Ns"] == ns
    want, counts = O.refine_batch(O.Reference(vol, n / 2), c, imgs, rows, ccf_mode=0)
    g = H.Reference(vol, n / 2)
    got = g.refine(c, imgs, rows)
    assert g.last_counts()["n_global"] == counts[0]
    same_grid_point_and_shift(want, got, px, d["step"])


def test_pyp_default_search_at_256_matches_oracle_mode_0(H, O):
    """BASELINE.json configs[1] geometry as PYP calls it: 256^2, 15 degrees, band 64 px, search range 0 = mask radius (82 pixels = 41
    steps of the 128-point grid: 83 x 83 shifts per orientation).  Grid points and shifts against the oracle's zero-filled inverse
    transform, then the whole default call (20 hits refined, best continued at the full band)."""
    n, px, m = 256, 1.0, 16
    vol, stack, rows = synth.make_dataset(n, m, pixel=px, snr=0.05)
    imgs = stack.numpy()
    base = dict(box=n, pixel_size=px, mask_radius=0.32 * n * px, res_high=4.0, res_search=4.0, search_range_x=0.0, search_range_y=0.0, res_signed_cc=30.0)
    raw = RefineCfg.make(local_refine=0, iters_hit=-1, **base)
    d = O.band_dims(raw)
    assert (d["Ns"], d["RSx"], d["RSy"], d["n_orient"]) == (128, 41, 41, 4416)
    o, g = O.Reference(vol, n / 2), H.Reference(vol, n / 2)
    want, counts = O.refine_batch(o, raw, imgs, rows, ccf_mode=0)
    got = g.refine(raw, imgs, rows)
    assert g.last_counts()["n_global"] == counts[0] == 4416
    same_grid_point_and_shift(want, got, px, d["step"])
    full = RefineCfg.make(**base)
    want, _ = O.refine_batch(o, full, imgs, rows, ccf_mode=0)
    got = g.refine(full, imgs, rows)
    assert synth.angular_error_deg(want, got).max() < ANG_TOL_DEG and synth.shift_error_px(want, got, px).max() < SHIFT_TOL_PX
    assert np.abs(want[:, 14] - got[:, 14]).max() < 0.02


def test_rows_of_the_window_in_several_chunks(H, O, monkeypatch):
    """Windows whose rows do not fit the LDS next to the particle's table are transformed in row chunks (128-point grid, more than
    +-44 steps); forced here on a small case: the chunked run equals the one-chunk run."""
    n, px = 64, 2.0
    vol, imgs, rows = dataset(n, 6, px, 0.1)
    g = H.Reference(vol, n / 2)
    c = raw_cfg(n, px, 24.0, 10.24)
    a = g.refine(c, imgs, rows)
    monkeypatch.setenv("PPM_GFFT_ROWS", "10")
    b = g.refine(c, imgs, rows)
    same_grid_point_and_shift(a, b, px, 2.0)
    assert np.array_equal(a[:, 14], b[:, 14])


def test_both_search_kernels_agree_at_256_on_256_particles(H, monkeypatch):
    """PYP's default window (+-41 steps of the 128-point grid) at 256^2 / 15 deg / band 64 on 256 particles through the two independent
    kernels: k_gfft (one pruned 2-D transform per orientation) and k_global (25 overlapping register tiles of +-8 steps, merged): the same
    grid orientation and integer shift for every particle, scores to rounding."""
    n, px, m = 256, 1.0, 256
    vol, stack, rows = synth.make_dataset(n, m, pixel=px, snr=0.05, device="cuda", unique=64)
    c = RefineCfg.make(box=n, pixel_size=px, mask_radius=0.32 * n * px, res_high=4.0, res_search=4.0, search_range_x=0.0, search_range_y=0.0,
                       res_signed_cc=30.0, local_refine=0, iters_hit=-1)
    g = H.Reference(vol, n / 2)
    monkeypatch.setenv("PPM_GLOBAL_PATH", "tiles")
    a = g.refine(c, stack, rows)
    monkeypatch.setenv("PPM_GLOBAL_PATH", "fft")
    b = g.refine(c, stack, rows)
    monkeypatch.delenv("PPM_GLOBAL_PATH")
    g.close()
    same_grid_point_and_shift(a, b, px, 2.0)
