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


def test_oracle_global_search_finds_alignments_from_random_starts(subtomos):
    """ppm_sva_cfg.search_mode 1 (the protocol's alignment_mode 0, iteration_002_mode_3.xml:29-38): from rotations anywhere on SO(3)
    the amplitude-ranked grid + 25 refined candidates land on the true alignment; the refinement mode alone cannot; mode 2 moves
    shifts only."""
    n, vol, vols, poses, wedges, O, ref = subtomos
    rng = np.random.default_rng(0)
    start = poses[:3].copy()
    for v in range(3):
        R = synth.euler_matrix(rng.uniform(0, 360), np.degrees(np.arccos(rng.uniform(-1, 1))), rng.uniform(0, 360))
        start[v, :9] = (poses[v, :9].reshape(3, 3) @ R).ravel()
        start[v, 9:] += rng.normal(0, 1.5, 3)
    assert synth.pose_angle_error(start, poses[:3]).min() > 60
    out, sc, _ = O.sva_align(ref, cfg_for(n, search_mode=1, global_step=20.0), vols[:3], wedges[:3], start)
    assert synth.pose_angle_error(out, poses[:3]).max() < 1.2 and np.linalg.norm(out[:, 9:] - poses[:3, 9:], axis=1).max() < 0.2 and sc.min() > 0.8
    loc, lsc, _ = O.sva_align(ref, cfg_for(n), vols[:3], wedges[:3], start)
    assert synth.pose_angle_error(loc, poses[:3]).min() > 50 and lsc.max() < 0.7
    tr, _, _ = O.sva_align(ref, cfg_for(n, search_mode=2), vols[:3], wedges[:3], start)
    assert np.array_equal(tr[:, :9], start[:, :9]) and not np.array_equal(tr[:, 9:], start[:, 9:])


@pytest.mark.gpu
def test_gpu_global_search_matches_oracle():
    """The grid scores (k_sva_global), the candidates and their refinement against the oracle: same poses within BASELINE's tolerance."""
    from oracle import oracle as O
    from pyp_amd import host
    n = 32
    vol, vols, poses, wedges = synth.make_subtomograms(n, 6, snr=0.5, wedge=(-54.0, 60.0))
    rng = np.random.default_rng(4)
    start = poses.copy()
    for v in range(len(start)):
        R = synth.euler_matrix(rng.uniform(0, 360), np.degrees(np.arccos(rng.uniform(-1, 1))), rng.uniform(0, 360))
        start[v, :9] = (poses[v, :9].reshape(3, 3) @ R).ravel()
        start[v, 9:] += rng.normal(0, 1.5, 3)
    for kw in (dict(search_mode=1, global_step=20.0), dict(search_mode=1, global_step=30.0, n_candidates=8, tol_shift=0.0), dict(search_mode=2)):
        c = cfg_for(n, **kw)
        s0 = start if kw.get("tol_shift", 1) else np.concatenate([start[:, :9], poses[:, 9:]], axis=1)
        want, wsc, _ = O.sva_align(O.Reference(vol, n / 2), c, vols.numpy(), wedges, s0)
        g = host.Reference(vol, n / 2)
        got, gsc = g.sva_align(c, vols.numpy(), wedges, s0)
        assert synth.pose_angle_error(want, got).max() < 0.1 and np.abs(want[:, 9:] - got[:, 9:]).max() < 0.5, kw
        assert np.abs(wsc - gsc).max() < 2e-3
        if kw["search_mode"] == 1:         # from anywhere on SO(3) and shifts a few pixels off: (nearly) all starts reach the true alignment
            err = synth.pose_angle_error(got, poses)
            assert (err < 1.5).sum() >= len(err) - 1 and err.max() < 8.0 and gsc.min() > 0.8, (kw, err)
        got2, gsc2 = g.sva_align(c, vols.cuda(), wedges, s0)
        assert np.array_equal(got, got2) and np.array_equal(gsc, gsc2)


@pytest.mark.gpu
def test_host_volumes_in_several_chunks_equal_resident_volumes(monkeypatch):
    """Host volumes are uploaded chunk by chunk, the next chunk by a helper thread while the current one is searched
    (PPM_SVA_CHUNK forces chunks of 3 here: 11 sub-volumes = 4 chunks, the transforms of a chunk in one batch): same bits as
    one call on resident volumes."""
    from pyp_amd import host
    n = 32
    vol, vols, poses, wedges = synth.make_subtomograms(n, 11, snr=0.5)
    start = synth.perturb_poses(poses, 3.0, 1.0)
    g = host.Reference(vol, n / 2)
    resident, rsc = g.sva_align(cfg_for(n), vols.cuda(), wedges, start)
    monkeypatch.setenv("PPM_SVA_CHUNK", "3")
    chunked, csc = g.sva_align(cfg_for(n), vols.numpy(), wedges, start)
    assert np.array_equal(resident, chunked) and np.array_equal(rsc, csc)


@pytest.mark.gpu
def test_gpu_alignment_errors_are_loud():
    from pyp_amd import host, lib
    vol, vols, poses, wedges = synth.make_subtomograms(32, 2, snr=0.5)
    g = host.Reference(vol, 8)                                   # prepared for a band narrower than the low-pass limit
    with pytest.raises(lib.PpmError, match="ERROR"):
        g.sva_align(cfg_for(32), vols.numpy(), wedges, poses)
    with pytest.raises(lib.PpmError, match="ERROR"):
        host.Reference(vol, 16).sva_align(cfg_for(48), np.zeros((2, 48, 48, 48), np.float32), wedges, poses)


def test_table_line_to_pose_matches_reference_spa_euler_angles():
    """A 3DAVG line (normal, 4 x 4 matrix) -> the particle pose PYP derives from it (spa_euler_angles at tilt 0, geometry/core.py:238-470;
    golden produced by that function), and the inverse used when the refined pose is written back."""
    import json
    import os
    from pyp_amd import sva
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_r02.json")))["sva_matrix_to_particle"]
    for c in gold:
        N, p = sva.line_to_pose(c["normal"], c["matrix"])
        want = np.array(c["particle"])
        assert np.abs(N - synth.euler_matrix(-want[0], -want[1], -want[2])).max() < 1e-9
        assert np.abs(p - want[3:]).max() < 1e-9
        got = sva.particle_from_pose(N, p)
        assert np.abs(synth.euler_matrix(-got[0], -got[1], -got[2]) - N).max() < 1e-9 and np.allclose(got[3:], want[3:])
        back = sva.pose_to_matrix(N, p, c["normal"])
        assert np.abs(back - np.array(c["matrix"])).max() < 1e-9


def test_volumes_table_roundtrip_and_protocol_fields(tmp_path):
    from pyp_amd import sva
    tab = np.zeros((3, 32))
    tab[:, 0] = [1, 2, 3]; tab[:, 1] = -60; tab[:, 2] = 60; tab[:, 3:6] = [[10, 20, 30]] * 3
    for k in range(3):
        tab[k, 12:28] = np.eye(4).ravel()
    names = ["TS_01_spk0000.rec", "TS_01_spk0001.rec", "sub/TS_02_vir0001_spk0003.mrc"]
    sva.write_volumes(str(tmp_path / "d_volumes.txt"), tab, names)
    t2, n2 = sva.read_volumes(str(tmp_path / "d_volumes.txt"))
    assert np.allclose(t2, tab) and n2 == names
    assert open(tmp_path / "d_volumes.txt").readline().startswith("number\tlwedge\tuwedge\tposX")
    xml = tmp_path / "iteration_002_mode_3.xml"
    xml.write_text("""<?xml version="1.0"?><config><general><mode>3</mode><metric><use_missing_wedge>1</use_missing_wedge></metric></general>
      <refine><refine_image_window_x>20</refine_image_window_x></refine>
      <mra><mra_image_window_x>32</mra_image_window_x><mra_image_window_y>32</mra_image_window_y><mra_image_window_z>28</mra_image_window_z>
      <mra_image_window_sigma>4</mra_image_window_sigma><mra_high_pass_cutoff>.05</mra_high_pass_cutoff><mra_high_pass_decay>.01</mra_high_pass_decay>
      <mra_low_pass_cutoff>0.125</mra_low_pass_cutoff><mra_low_pass_decay>.05</mra_low_pass_decay>
      <mra_out_of_plane_search_range>15</mra_out_of_plane_search_range><mra_shifts_tolerance>10.0</mra_shifts_tolerance></mra></config>""")
    c = sva.cfg_from_xml(str(xml), 96)
    assert (c.box, list(c.window), c.window_sigma) == (96, [32.0, 32.0, 28.0], 4.0)
    assert abs(c.lowpass_cutoff - 0.125) < 1e-7 and abs(c.highpass_decay - 0.01) < 1e-7 and c.tol_angle == 15.0 and c.tol_shift == 10.0 and c.use_missing_wedge == 1
    assert c.search_mode == 0 and c.n_candidates == 25                      # no alignment_mode in the protocol: refinement
    xml.write_text(xml.read_text().replace("<metric>", "<metric><alignment_mode>0</alignment_mode><number_of_candidate_peaks_to_search>12</number_of_candidate_peaks_to_search>"))
    c = sva.cfg_from_xml(str(xml), 96)
    assert c.search_mode == 1 and c.n_candidates == 12                      # the protocol's 0 = global search


@pytest.mark.gpu
def test_sva_align_executable_refines_a_table(tmp_path):
    import os
    import subprocess
    import sys
    from pyp_amd import sva
    from pyp_amd.formats import mrc
    n = 32
    vol, vols, poses, wedges = synth.make_subtomograms(n, 5, snr=0.5)
    start = synth.perturb_poses(poses, 3.0, 1.0)
    tab = np.zeros((5, 32)); names = []
    for k in range(5):
        tab[k, 0], tab[k, 1], tab[k, 2] = k + 1, wedges[k, 0], wedges[k, 1]
        tab[k, 9:12] = [10.0 * k, 0.0, -20.0 * k]
        tab[k, 12:28] = sva.pose_to_matrix(start[k, :9], start[k, 9:], tab[k, 9:12])
        names.append(f"TS_01_spk{k:04d}.rec")
        mrc.write(vols[k].numpy(), str(tmp_path / names[-1]))
    sva.write_volumes(str(tmp_path / "d_volumes.txt"), tab, names)
    mrc.write(vol, str(tmp_path / "ref.mrc"))
    (tmp_path / "p.xml").write_text("""<config><general><mode>2</mode><metric><use_missing_wedge>1</use_missing_wedge></metric></general>
      <refine><refine_image_window_x>12</refine_image_window_x><refine_image_window_y>12</refine_image_window_y><refine_image_window_z>12</refine_image_window_z>
      <refine_image_window_sigma>2</refine_image_window_sigma><refine_high_pass_cutoff>.03</refine_high_pass_cutoff><refine_high_pass_decay>.01</refine_high_pass_decay>
      <refine_low_pass_cutoff>0.30</refine_low_pass_cutoff><refine_low_pass_decay>.04</refine_low_pass_decay>
      <refine_out_of_plane_search_range>10</refine_out_of_plane_search_range><refine_shifts_tolerance>4.0</refine_shifts_tolerance></refine></config>""")
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bin", "sva_align")
    r = subprocess.run([sys.executable, exe, "p.xml", "d_volumes.txt", "ref.mrc", "out_volumes.txt"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0 and "SVA: Normal termination" in r.stdout, r.stdout + r.stderr
    out, n2 = sva.read_volumes(str(tmp_path / "out_volumes.txt"))
    got = np.array([np.concatenate([sva.line_to_pose(out[k, 9:12], out[k, 12:28])[0].ravel(), sva.line_to_pose(out[k, 9:12], out[k, 12:28])[1]]) for k in range(5)])
    assert synth.pose_angle_error(got, poses).mean() < 0.4 * synth.pose_angle_error(start, poses).mean()
    assert np.linalg.norm(got[:, 9:] - poses[:, 9:], axis=1).max() < 0.2 and (out[:, 31] > 0.8).all() and np.array_equal(out[:, 9:12], tab[:, 9:12])
    r = subprocess.run([sys.executable, exe, "p.xml", "missing.txt", "ref.mrc", "o.txt"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode != 0 and "ERROR" in r.stdout and not (tmp_path / "o.txt").exists()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [32, 48, 80, 96, 112])
def test_gpu_two_step_transforms_equal_the_staged_ones(n, monkeypatch):
    """Boxes that are multiples of 16 go through k_sva_x16 / k_sva_yz16 (N = 16 M, M = 2, 3, 5, 6, 7 here; other layout of the work arrays); PPM_SVA_GENERIC_FFT=1 forces the staged line transforms every other box takes.
    Same scores and poses to rounding."""
    from pyp_amd import host
    vol, vols, poses, wedges = synth.make_subtomograms(n, 5, snr=0.5, wedge=(-50.0, 62.0))
    start = synth.perturb_poses(poses, 3.0, 1.5)
    for kw in (dict(), dict(lowpass=(0.45, 0.05)), dict(tol_angle=0.0, tol_shift=0.0)):          # pruned band; band beyond N/2 - 1 (no pruning); scores only
        c = cfg_for(n, **kw)
        g = host.Reference(vol, n / 2)
        monkeypatch.delenv("PPM_SVA_GENERIC_FFT", raising=False)
        fast, fsc = g.sva_align(c, vols.numpy(), wedges, start)
        monkeypatch.setenv("PPM_SVA_GENERIC_FFT", "1")
        slow, ssc = g.sva_align(c, vols.numpy(), wedges, start)
        monkeypatch.delenv("PPM_SVA_GENERIC_FFT", raising=False)
        assert np.abs(fsc - ssc).max() < 2e-5, (n, kw)
        assert synth.pose_angle_error(fast, slow).max() < 0.02 and np.abs(fast[:, 9:] - slow[:, 9:]).max() < 0.02, (n, kw)
        # the z pass emits the band's samples itself (round 5); PPM_SVA_FOLD=0 writes the array back and lets k_sva_gather16 pick them: the same numbers
        monkeypatch.setenv("PPM_SVA_FOLD", "0")
        two, tsc = g.sva_align(c, vols.numpy(), wedges, start)
        monkeypatch.delenv("PPM_SVA_FOLD", raising=False)
        assert np.array_equal(two, fast) and np.array_equal(tsc, fsc), (n, kw)
        g.close()


@pytest.mark.gpu
def test_gpu_normalisation_through_the_transform_holds_for_a_large_density_offset():
    """The two-step x pass transforms the RAW windowed sub-volume and the normalisation is applied at the samples,
    (F - mean F_window) / sigma.  Sub-volumes whose mean is 40 standard deviations away from zero (unnormalised tomogram densities)
    must still give the oracle's scores and poses, and the same as their normalised copies to rounding."""
    from oracle import oracle as O
    from pyp_amd import host
    n = 48
    vol, vols, poses, wedges = synth.make_subtomograms(n, 5, snr=0.5, wedge=(-55.0, 58.0))
    raw = (vols.numpy() * 7.0 + 280.0).astype(np.float32)
    start = synth.perturb_poses(poses, 3.0, 1.0)
    c = cfg_for(n)
    want, wsc, _ = O.sva_align(O.Reference(vol, n / 2), c, raw, wedges, start)
    g = host.Reference(vol, n / 2)
    got, gsc = g.sva_align(c, raw, wedges, start)
    assert synth.pose_angle_error(want, got).max() < 0.1 and np.abs(want[:, 9:] - got[:, 9:]).max() < 0.5 and np.abs(wsc - gsc).max() < 2e-3
    ref_, rsc = g.sva_align(c, vols.numpy(), wedges, start)
    assert np.abs(rsc - gsc).max() < 5e-4 and synth.pose_angle_error(ref_, got).max() < 0.05
    g.close()
