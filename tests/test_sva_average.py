"""Sub-tomogram AVERAGE (BASELINE config 5, "Subtomogram averaging (3davg)"; the step of a 3DAVG iteration that writes
`<dataset>_iteration_%03d_refined_selected_average_0.mrc`, src/pyp/refine/tomo_avg/sub_tomo_avg.py:79-94, src/pyp_main.py:3076-3100):
the oracle's restatement against synthetic truth (CPU), and the HIP path (ppm_sva_insert through the C ABI) against the oracle (GPU)."""
import numpy as np
import pytest

from pyp_amd import synth
from pyp_amd.abi import FinalCfg, SvaCfg


def cc(a, b, mask=None):
    a, b = (a, b) if mask is None else (a[mask], b[mask])
    a, b = a - a.mean(), b - b.mean()
    return float((a * b).sum() / np.sqrt((a * a).sum() * (b * b).sum()))


def ball(n, r):
    k = np.arange(n) - n // 2
    z, y, x = np.meshgrid(k, k, k, indexing="ij")
    return (x * x + y * y + z * z) < r * r


def oracle_average(O, n, vols, wedges, poses, use_wedge=1, index=None):
    acc, cnt = np.zeros(O.accum_floats(n), np.float32), np.zeros(2, np.int64)
    O.sva_insert(acc, cnt, SvaCfg.make(n, use_missing_wedge=use_wedge), vols, wedges, poses, index)
    h1, h2, fl, stats = O.finalize(acc, n, 1.0, FinalCfg(molecular_mass_kda=0.0, inner_radius=0.0, outer_radius=0.0, mask_falloff=0.0))
    return acc, cnt, h1, h2, fl, stats


def test_oracle_average_recovers_the_reference_and_fills_the_wedge():
    """With the true poses the average of 24 noisy sub-tomograms (missing wedge +-60 deg, random orientations) correlates with the
    phantom far better than any single sub-volume brought into the reference frame, and misaligned poses give a worse map."""
    from oracle import oracle as O
    n, nv = 32, 24
    vol, vols, poses, wedges = synth.make_subtomograms(n, nv, snr=0.5, seed=3)
    vols = vols.numpy()
    acc, cnt, h1, h2, fl, stats = oracle_average(O, n, vols, wedges, poses)
    m = ball(n, 0.4 * n)
    c_avg = cc(fl, vol, m)
    assert list(cnt) == [12, 12] and c_avg > 0.9
    _, _, _, _, one, _ = oracle_average(O, n, np.concatenate([vols[:1], vols[:1]]), wedges[:2], np.vstack([poses[:1], poses[:1]]))
    assert c_avg > cc(one, vol, m) + 0.15
    bad = synth.perturb_poses(poses, 12.0, 2.0)
    _, _, _, _, flb, _ = oracle_average(O, n, vols, wedges, bad)
    assert cc(flb, vol, m) < c_avg - 0.05
    # the half maps come from disjoint sub-volumes and agree at low resolution (FSC column of the table)
    assert stats[1, 3] > 0.9 and cc(h1, h2, m) > 0.7
    # weights: every voxel of the band counts the sub-volumes whose wedge covers it
    w = acc.reshape(2, n, n, n // 2 + 1, 3)[..., 2]
    assert w.max() <= 12 and w.sum() > 0 and np.all(w == np.round(w))


def test_oracle_average_without_wedge_weights_equals_the_plain_mean():
    """use_missing_wedge = 0 and identity poses: the average is the plain mean of the normalised volumes (band-limited at
    box/2 - 1 pixels, so compared on a smooth input)."""
    from oracle import oracle as O
    n = 32
    vol = synth.phantom(n)
    rng = np.random.default_rng(0)
    vols = np.stack([vol * s for s in (1.0, 2.0, 0.5, 3.0)]).astype(np.float32)            # scale drops out through the normalisation
    poses = np.tile(np.concatenate([np.eye(3).ravel(), np.zeros(3)]), (4, 1))
    poses[:, 9:] = 0.0
    wedges = np.tile(np.float32([-60, 60]), (4, 1))
    _, cnt, h1, h2, fl, _ = oracle_average(O, n, vols, wedges, poses, use_wedge=0)
    want = (vol - vol.mean()) / vol.std()
    m = ball(n, 0.42 * n)
    # identity poses sample the transforms ON the grid: nothing is interpolated, so the gridding correction of the finalisation
    # (division by sinc^2 per axis, right for rotated sub-volumes) is taken out again before the comparison
    t = (np.arange(n) - n // 2) / n
    s1 = np.sinc(t) ** 2
    g3 = s1[:, None, None] * s1[None, :, None] * s1[None, None, :]
    fl = fl * g3
    assert cc(fl, want, m) > 0.995 and abs((fl[m] * want[m]).sum() / (want[m] ** 2).sum() - 1.0) < 0.02          # the mean, at the mean's scale
    # integer shifts are undone exactly.  Convention (ppm_sva_align, synth.make_subtomograms): F_v(k) = Ref(N k) e^{+2 pi i k.p / n}, i.e.
    # the sub-volume shows the reference displaced by -p: content rolled by d = (2, -1, 3) pixels in (x, y, z) has p = -d
    sh = np.roll(vol, (3, -1, 2), axis=(0, 1, 2))[None].astype(np.float32)       # axes z, y, x
    p1 = poses[:1].copy(); p1[0, 9:] = (-2.0, 1.0, -3.0)
    two = np.concatenate([sh, sh])
    _, _, _, _, fs, _ = oracle_average(O, n, two, wedges[:2], np.vstack([p1, p1]), use_wedge=0)
    fs = fs * g3
    assert cc(fs, want, m) > 0.995 and np.abs(fs - fl)[m].max() < 0.05 * np.abs(fl)[m].max()


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("n,nv,generic", [(48, 20, False), (40, 9, False), (48, 6, True)])
def test_gpu_average_matches_oracle(n, nv, generic, monkeypatch):
    """ppm_sva_insert + ppm_finalize against orc_sva_insert + orc_finalize on the same sub-volumes, wedges and (perturbed) poses:
    accumulators to float32 round-off, maps and FSC table to the tolerances of the reconstruction tests.  48 = 16 x 3 takes the
    two-step transforms, 40 (and PPM_SVA_GENERIC_FFT) the staged ones; several batches when nv > 32 is covered at size."""
    from oracle import oracle as O
    from pyp_amd import host as H
    if generic:
        monkeypatch.setenv("PPM_SVA_GENERIC_FFT", "1")
    vol, vols, poses, wedges = synth.make_subtomograms(n, nv, snr=0.5, seed=5)
    wedges[::3] = (-50.0, 64.0)
    poses = synth.perturb_poses(poses, 1.0, 0.5)
    index = np.arange(nv) * 3 + 1
    acc_o, cnt_o, h1o, h2o, flo, st_o = oracle_average(O, n, vols.numpy(), wedges, poses, index=index)
    acc = H.Accumulator(n, 1.0, "C1")
    cfg = SvaCfg.make(n, use_missing_wedge=1)
    acc.sva_insert(cfg, vols.numpy()[: nv // 2], wedges[: nv // 2], poses[: nv // 2], index[: nv // 2])      # two calls: sums accumulate
    acc.sva_insert(cfg, vols.numpy()[nv // 2:], wedges[nv // 2:], poses[nv // 2:], index[nv // 2:])
    g = acc.download()
    assert acc.counts() == [int(cnt_o[0]), int(cnt_o[1])]
    go, gg = acc_o.reshape(-1, 3), g.reshape(-1, 3)
    assert np.abs(gg[:, 2] - go[:, 2]).sum() <= 1e-4 * go[:, 2].sum()            # a wedge edge can flip a voxel between float and double
    ok = gg[:, 2] == go[:, 2]
    assert np.linalg.norm((gg - go)[ok, :2]) < 2e-5 * np.linalg.norm(go[:, :2])
    h1, h2, fl, st = acc.finalize(FinalCfg(molecular_mass_kda=0.0, inner_radius=0.0, outer_radius=0.0, mask_falloff=0.0))
    acc.close()
    for a, b in ((h1, h1o), (h2, h2o), (fl, flo)):
        assert np.abs(a - b).max() < 2e-3 * np.abs(b).max()
    assert np.abs(st[:, 3] - st_o[:, 3]).max() < 2e-3
    assert cc(fl, vol, ball(n, 0.4 * n)) > 0.85


@pytest.mark.gpu
def test_gpu_average_from_device_volumes_and_without_wedge(monkeypatch):
    """Resident sub-volumes (a CUDA tensor) give the same accumulator as host volumes; use_missing_wedge = 0 weighs every voxel
    with the number of sub-volumes."""
    import torch
    from pyp_amd import host as H
    n, nv = 48, 8
    vol, vols, poses, wedges = synth.make_subtomograms(n, nv, snr=1.0, seed=9)
    a1, a2 = H.Accumulator(n, 1.0, "C1"), H.Accumulator(n, 1.0, "C1")
    cfg = SvaCfg.make(n, use_missing_wedge=0)
    a1.sva_insert(cfg, vols.numpy(), wedges, poses)
    a2.sva_insert(cfg, vols.cuda(), wedges, poses)
    x, y = a1.download(), a2.download()
    a1.close(); a2.close()
    assert np.array_equal(x, y)
    w = x.reshape(2, n, n, n // 2 + 1, 3)[..., 2]
    assert set(np.unique(w)) == {0.0, 4.0}


@pytest.mark.gpu
def test_two_iterations_align_average_align_improve_the_score():
    """A 3DAVG iteration as the reference runs it (align all volumes to the reference, average, next iteration aligns to the
    average; src/pyp_main.py:3007-3107): starting from a blurred reference and perturbed poses, the average of iteration 1 is a
    better reference than the one it was aligned to - the mean alignment score and the pose accuracy of iteration 2 improve."""
    from scipy.ndimage import gaussian_filter
    from pyp_amd import host as H
    n, nv = 48, 48
    vol, vols, poses, wedges = synth.make_subtomograms(n, nv, snr=0.3, seed=21)
    start = synth.perturb_poses(poses, 4.0, 1.5)
    cfg = SvaCfg.make(n, window=(0.36 * n,) * 3, window_sigma=3.0, highpass=(0.03, 0.01), lowpass=(0.3, 0.05), tol_angle=12.0, tol_shift=5.0)
    ref0 = gaussian_filter(vol, 2.5).astype(np.float32)              # a poor starting reference (what a global average looks like)
    r0 = H.Reference(ref0, n / 2)
    p1, s1 = r0.sva_align(cfg, vols.numpy(), wedges, start)
    r0.close()
    acc = H.Accumulator(n, 1.0, "C1")
    acc.sva_insert(cfg, vols.numpy(), wedges, p1)
    _, _, avg1, _ = acc.finalize(FinalCfg(molecular_mass_kda=0.0, inner_radius=0.0, outer_radius=0.0, mask_falloff=0.0))
    acc.close()
    m = ball(n, 0.4 * n)
    assert cc(avg1, vol, m) > cc(ref0, vol, m)
    r1 = H.Reference(avg1, n / 2)
    p2, s2 = r1.sva_align(cfg, vols.numpy(), wedges, p1)
    r1.close()
    assert s2.mean() > s1.mean()
    e1, e2 = synth.pose_angle_error(p1, poses), synth.pose_angle_error(p2, poses)
    assert np.median(e2) <= np.median(e1) + 0.05 and np.median(e2) < np.median(synth.pose_angle_error(start, poses))


@pytest.mark.gpu
def test_empty_and_single_inputs_are_no_ops_not_faults():
    """Zero sub-volumes: alignment returns empty arrays, insertion leaves the accumulator untouched; one sub-volume lands in the half its
    index parity names (the reference's tests feed ragged tables: the edge of the range must not reach a kernel with a zero grid)."""
    from pyp_amd import host as H
    n = 32
    vol, vols, poses, wedges = synth.make_subtomograms(n, 2, snr=0.5)
    cfg = SvaCfg.make(n, use_missing_wedge=1, tol_angle=5.0, tol_shift=2.0)
    out, sc = H.Reference(vol, n / 2).sva_align(cfg, vols.numpy()[:0], wedges[:0], poses[:0])
    assert out.shape == (0, 12) and sc.shape == (0,)
    acc = H.Accumulator(n, 1.0, "C1")
    acc.sva_insert(cfg, vols.numpy()[:0], wedges[:0], poses[:0], np.zeros(0, np.int64))
    assert acc.counts() == [0, 0] and not acc.download().any()
    acc.sva_insert(cfg, vols.numpy()[:1], wedges[:1], poses[:1], np.array([5]))
    assert acc.counts() == [0, 1]
    acc.close()


def test_filtered_map_applies_the_protocol_window_and_band_pass():
    """`<average>_filtered.mrc`: the map as the metric sees it - zero outside the (hard) window, and a plane wave inside the pass band
    survives while one beyond the low-pass is removed."""
    from pyp_amd import sva
    n = 32
    cfg = SvaCfg.make(n, window=(8, 8, 6), window_sigma=0.0, highpass=(0.05, 0.01), lowpass=(0.2, 0.02))
    k = np.arange(n) - n // 2
    z, y, x = np.meshgrid(k, k, k, indexing="ij")
    f = sva.filtered_map(np.ones((n, n, n), np.float32) + np.cos(2 * np.pi * 4 * x / n).astype(np.float32), SvaCfg.make(n, highpass=(0.05, 0.01), lowpass=(0.2, 0.02)))
    assert abs(np.abs(np.fft.fftn(f))[0, 0, 4] / (n ** 3 / 2) - 1.0) < 1e-3                # 4 / 32 = 0.125 cycles per pixel: inside the band
    g = sva.filtered_map(np.cos(2 * np.pi * 12 * x / n).astype(np.float32), SvaCfg.make(n, highpass=(0.05, 0.01), lowpass=(0.2, 0.02)))
    assert np.abs(g).max() < 1e-3                                                          # 0.375 cycles per pixel: removed
    w = sva.band_weights(cfg, n)
    assert w[0, 0, 0] < 1e-4 and abs(w[0, 0, 4] - 1.0) < 1e-12 and w.shape == (n, n, n)
    inside = (np.abs(x) <= 8) & (np.abs(y) <= 8) & (np.abs(z) <= 6)
    h = sva.filtered_map(np.random.default_rng(0).normal(size=(n, n, n)).astype(np.float32), SvaCfg.make(n, window=(8, 8, 6), window_sigma=0.0, highpass=(0, 0), lowpass=(0, 0)))
    assert np.abs(h[~inside]).max() < 1e-5 and np.abs(h[inside]).max() > 0.5


@pytest.mark.gpu
def test_sva_align_executable_writes_the_average(tmp_path):
    """bin/sva_align with a fifth argument: one pass over the table aligns every sub-volume and averages it at its refined pose;
    the five average files appear (the names PYP expects are `<prefix>.mrc` and `<prefix>_filtered.mrc`, src/pyp_main.py:3076-3100),
    the average resembles the phantom more than the blurred reference it was aligned to, a failure leaves no output."""
    import os
    import subprocess
    import sys
    from scipy.ndimage import gaussian_filter
    from pyp_amd import sva
    from pyp_amd.formats import mrc
    n, nv = 32, 16
    vol, vols, poses, wedges = synth.make_subtomograms(n, nv, snr=0.5, seed=8)
    start = synth.perturb_poses(poses, 3.0, 1.0)
    tab = np.zeros((nv, 32)); names = []
    for k in range(nv):
        tab[k, 0], tab[k, 1], tab[k, 2] = k + 1, wedges[k, 0], wedges[k, 1]
        tab[k, 12:28] = sva.pose_to_matrix(start[k, :9], start[k, 9:], tab[k, 9:12])
        names.append(f"TS_01_spk{k:04d}.rec")
        mrc.write(vols[k].numpy(), str(tmp_path / names[-1]))
    sva.write_volumes(str(tmp_path / "d_volumes.txt"), tab, names)
    ref0 = gaussian_filter(vol, 1.5).astype(np.float32)
    mrc.write(ref0, str(tmp_path / "ref.mrc"))
    (tmp_path / "p.xml").write_text("""<config><general><mode>3</mode><metric><use_missing_wedge>1</use_missing_wedge><alignment_mode>1</alignment_mode></metric></general>
      <mra><mra_image_window_x>12</mra_image_window_x><mra_image_window_y>12</mra_image_window_y><mra_image_window_z>12</mra_image_window_z>
      <mra_image_window_sigma>2</mra_image_window_sigma><mra_high_pass_cutoff>.03</mra_high_pass_cutoff><mra_high_pass_decay>.01</mra_high_pass_decay>
      <mra_low_pass_cutoff>0.30</mra_low_pass_cutoff><mra_low_pass_decay>.04</mra_low_pass_decay>
      <mra_out_of_plane_search_range>10</mra_out_of_plane_search_range><mra_shifts_tolerance>4.0</mra_shifts_tolerance></mra></config>""")
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bin", "sva_align")
    prefix = "d_iteration_002_refined_selected_average_0"
    r = subprocess.run([sys.executable, exe, "p.xml", "d_volumes.txt", "ref.mrc", "d_iteration_002_alignments_to_reference_0.txt", prefix], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0 and "SVA: Normal termination" in r.stdout and "Averaged 8 + 8 sub-volumes" in r.stdout, r.stdout + r.stderr
    for suffix in (".mrc", "_filtered.mrc", "_half1.mrc", "_half2.mrc", "_statistics.txt"):
        assert (tmp_path / (prefix + suffix)).exists(), suffix
    avg = mrc.read(str(tmp_path / (prefix + ".mrc")))
    m = ball(n, 0.4 * n)
    assert avg.shape == (n, n, n) and cc(avg, vol, m) > 0.85 and cc(avg, vol, m) > cc(ref0, vol, m) - 0.02
    st = np.loadtxt(str(tmp_path / (prefix + "_statistics.txt")), comments="C")
    assert st.shape == (n // 2 - 1, 7) and st[0, 3] > 0.9
    flt = mrc.read(str(tmp_path / (prefix + "_filtered.mrc")))
    assert np.abs(flt[0, 0, :]).max() < 0.05 * np.abs(flt).max()                          # the window has removed the box corners
    r = subprocess.run([sys.executable, exe, "p.xml", "missing.txt", "ref.mrc", "o.txt", "o_avg"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode != 0 and "ERROR" in r.stdout and not (tmp_path / "o.txt").exists() and not (tmp_path / "o_avg.mrc").exists()


@pytest.mark.gpu
def test_fused_align_and_average_equals_the_two_calls(monkeypatch):
    """ppm_sva_align_average (one pass: every chunk aligned, then added to the average while it is in device memory) gives the poses of
    ppm_sva_align and the accumulator of ppm_sva_insert at those poses - from host volumes in several chunks and from resident ones."""
    from pyp_amd import host as H
    n, nv = 32, 10
    vol, vols, poses, wedges = synth.make_subtomograms(n, nv, snr=0.5, seed=4)
    start = synth.perturb_poses(poses, 3.0, 1.0)
    cfg = SvaCfg.make(n, window=(12, 12, 12), window_sigma=2.0, highpass=(0.03, 0.01), lowpass=(0.3, 0.04), tol_angle=10.0, tol_shift=4.0)
    index = np.arange(nv) + 7
    ref = H.Reference(vol, n / 2)
    p0, s0 = ref.sva_align(cfg, vols.numpy(), wedges, start)
    a0 = H.Accumulator(n, 1.0, "C1")
    a0.sva_insert(cfg, vols.numpy(), wedges, p0, index)
    want, wc = a0.download(), a0.counts()
    a0.close()
    monkeypatch.setenv("PPM_SVA_CHUNK", "4")                     # host volumes: chunks of 4, the next one uploaded while this one is searched
    for src in (vols.numpy(), vols.cuda()):
        a1 = H.Accumulator(n, 1.0, "C1")
        p1, s1 = ref.sva_align(cfg, src, wedges, start, accumulator=a1, index=index)
        got, gc = a1.download(), a1.counts()
        a1.close()
        assert np.array_equal(p1, p0) and np.array_equal(s1, s0) and gc == wc
        assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max()            # batches of the gather differ between the two call shapes: float32 sums in another order
    ref.close()
