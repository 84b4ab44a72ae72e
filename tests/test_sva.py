"""Sub-tomogram alignment (3DAVG row): CPU oracle on synthetic sub-tomograms with a missing wedge, and the HIP path
(ppm_sva_align) against it.  Tolerances: BASELINE.json's 0.1 deg / 0.5 px between GPU and oracle."""
import numpy as np
import pytest

from pyp_amd import synth
from pyp_amd.abi import SvaCfg


@pytest.fixture(scope="module")
def subtomos():
    from oracle import oracle
    n = 32
    vol, vols, poses, wedges = synth.make_subtomograms(n, 8, snr=0.5)
    return n, vol, vols.numpy(), poses, wedges, oracle, oracle.Reference(vol, n / 2)


def cfg_for(n, **kw):
    base = dict(window=(12, 12, 12), window_sigma=2.0, highpass=(0.03, 0.01), lowpass=(0.30, 0.04), tol_angle=10.0, tol_shift=4.0)
    base.update(kw)
    return SvaCfg.make(n, **base)


def test_oracle_recovers_perturbed_alignments(subtomos):
    n, vol, vols, poses, wedges, O, ref = subtomos
    start = synth.perturb_poses(poses, 3.0, 1.0)
    out, sc, nev = O.sva_align(ref, cfg_for(n), vols, wedges, start)
    assert synth.pose_angle_error(out, poses).max() < 1.2 and synth.pose_angle_error(out, poses).mean() < 0.3 * synth.pose_angle_error(start, poses).mean()
    assert np.linalg.norm(out[:, 9:] - poses[:, 9:], axis=1).max() < 0.15
    _, at_truth, _ = O.sva_align(ref, cfg_for(n, tol_angle=0.0, tol_shift=0.0), vols, wedges, poses)
    assert np.abs(sc - at_truth).max() < 0.01 and sc.min() > 0.8
    # the missing wedge matters: scoring the empty wedge as data lowers the score
    _, no_wedge, _ = O.sva_align(ref, cfg_for(n, tol_angle=0.0, tol_shift=0.0, use_missing_wedge=0), vols, wedges, poses)
    assert (no_wedge < at_truth - 0.02).all()
    # bounds are respected
    tight, _, _ = O.sva_align(ref, cfg_for(n, tol_angle=0.5, tol_shift=0.25), vols, wedges, start)
    assert np.abs(tight[:, 9:] - start[:, 9:]).max() <= 0.25 + 1e-9 and synth.pose_angle_error(tight, start).max() <= 0.5 * np.sqrt(3) + 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("n,kw", [(32, {}), (32, dict(use_missing_wedge=0)), (32, dict(tol_angle=0.0)), (32, dict(tol_shift=0.0, highpass=(0.0, 0.0))),
                                  (48, dict(window=(18, 18, 14), lowpass=(0.22, 0.03)))])
def test_gpu_alignment_matches_oracle(n, kw):
    from oracle import oracle as O
    from pyp_amd import host
    vol, vols, poses, wedges = synth.make_subtomograms(n, 6, snr=0.5, wedge=(-54.0, 60.0))
    wedges[3] = (-40.0, 45.0)                                   # a sub-volume from a series with a narrower tilt range
    c = cfg_for(n, **kw)
    start = synth.perturb_poses(poses, 3.0, 1.0)
    want, wsc, _ = O.sva_align(O.Reference(vol, n / 2), c, vols.numpy(), wedges, start)
    g = host.Reference(vol, n / 2)
    got, gsc = g.sva_align(c, vols.numpy(), wedges, start)
    assert synth.pose_angle_error(want, got).max() < 0.1 and np.abs(want[:, 9:] - got[:, 9:]).max() < 0.5, kw
    assert np.abs(wsc - gsc).max() < 2e-3
    import torch
    got2, gsc2 = g.sva_align(c, vols.cuda(), wedges, start)                   # resident volumes, same bits
    assert np.array_equal(got, got2) and np.array_equal(gsc, gsc2)
    if kw.get("tol_angle", 1) and kw.get("tol_shift", 1):
        assert synth.pose_angle_error(got, poses).mean() < 0.4 * synth.pose_angle_error(start, poses).mean()


@pytest.mark.gpu
def test_gpu_alignment_errors_are_loud():
    from pyp_amd import host, lib
    vol, vols, poses, wedges = synth.make_subtomograms(32, 2, snr=0.5)
    g = host.Reference(vol, 8)                                   # prepared for a band narrower than the low-pass limit
    with pytest.raises(lib.PpmError, match="ERROR"):
        g.sva_align(cfg_for(32), vols.numpy(), wedges, poses)
    with pytest.raises(lib.PpmError, match="ERROR"):
        host.Reference(vol, 16).sva_align(cfg_for(48), np.zeros((2, 48, 48, 48), np.float32), wedges, poses)
