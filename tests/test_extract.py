"""Particle extraction: the numpy restatement is pinned by fixtures produced by the reference's own
normalize_image / extract_background (tests/golden/gen_golden.py); the HIP kernel is compared with the restatement."""
import json
import os

import numpy as np
import pytest

from oracle import extract_oracle as xo


def test_normalisation_restatement_matches_reference_outputs(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "golden.json")))["normalize_image"]
    z = np.load(os.path.join(golden_dir, "normalize_image.npz"))
    for i, (boxsize, radius, pixel, binning, bmean, bstd) in enumerate(g):
        img = z[f"in{i}"]
        m, s = xo.background(img.copy(), radius, pixel * binning)
        assert m == pytest.approx(bmean, rel=1e-13) and s == pytest.approx(bstd, rel=1e-13)
        assert np.allclose(xo.normalize_image(img.copy(), radius, pixel, int(binning)), z[f"out{i}"], rtol=1e-13, atol=1e-13)


def test_window_edge_rules():
    img = np.arange(40 * 50, dtype=np.float64).reshape(40, 50)
    w = xo.window(img, 25, 20, 16, 1)                       # fully inside: rows 12..27, cols 17..32
    assert np.array_equal(w, img[12:28, 17:33])
    w = xo.window(img, 3, 2, 16, 1)                         # hangs over the top-left corner
    assert np.array_equal(w[6:, 5:], img[0:10, 0:11]) and np.allclose(w[:6, :], img[0:10, 0:11].mean())
    w = xo.window(img, 25, 32, 16, 1)                       # touches the far row edge exactly: last row is filled (reference quirk)
    assert np.array_equal(w[:15, :], img[24:39, 17:33]) and np.allclose(w[15, :], img[24:39, 17:33].mean())
    w = xo.window(img, 50.0, 20.0, 16, 2)                   # coordinate binning halves the coordinates
    assert np.array_equal(w, img[2:18, 17:33])


@pytest.mark.gpu
def test_hip_extraction_matches_restatement():
    import torch
    from pyp_amd import host
    rng = np.random.default_rng(5)
    rows, cols, box = 300, 420, 64
    mic = rng.normal(10.0, 3.0, (rows, cols)).astype(np.float32)
    mic[:100, :120] = 7.0                                     # a dead (constant) corner -> an "empty" box
    coords = np.array([[200.0, 150.0], [10.0, 8.0], [415.0, 295.0], [388.0, 150.0], [200.5, 268.0], [50.0, 50.0],
                       [-500.0, 150.0], [100.7, 99.2]])
    want, empty = xo.extract(mic.astype(np.float64), coords, box, radius=40.0, pixelsize=2.0, coordinate_binning=1)
    got = host.extract_boxes(mic, coords, box, radius_A=40.0, pixel_size=2.0)
    assert got.shape == want.shape
    for i in range(len(coords)):
        if empty[i]:                                          # reference: unseeded noise; ours: deterministic unit noise
            assert abs(got[i].mean()) < 0.2 and 0.7 < got[i].std() < 1.3, i
        else:
            assert np.abs(got[i] - want[i]).max() < 2e-5, i
    assert empty[5] and empty[6] and not empty[0]
    # resident path: device micrograph in, device stack out; no normalisation
    out = torch.empty((len(coords), box, box), dtype=torch.float32, device="cuda")
    host.extract_boxes(torch.as_tensor(mic).cuda(), coords, box, 40.0, 2.0, normalize=False, fix_empty=False, out=out)
    want2, _ = xo.extract(mic.astype(np.float64), coords, box, 40.0, 2.0, normalize=False)
    assert np.abs(out.cpu().numpy() - want2).max() < 1e-5
    # coordinate binning
    got3 = host.extract_boxes(mic, coords[:1] * 2, box, 40.0, 1.0, coordinate_binning=2)
    want3, _ = xo.extract(mic.astype(np.float64), coords[:1] * 2, box, 40.0, 1.0, coordinate_binning=2)
    assert np.abs(got3 - want3).max() < 2e-5
