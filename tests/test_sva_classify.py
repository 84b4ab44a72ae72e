"""Classification of aligned sub-volumes (3DAVG protocol mode 1, src/pyp/refine/tomo_avg/sub_tomo_avg.py:435-449): the multi-reference
driver of pyp_amd/sva.py on the CPU oracle (two structures mixed, recovered from poor class references), and the same driver on the GPU
against it."""
import numpy as np
import pytest

from pyp_amd import sva, synth
from pyp_amd.abi import FinalCfg, SvaCfg


class OracleBackend:
    """The interface of sva.GpuBackend on the CPU oracle (test infrastructure)."""

    def scores(self, reference, cfg, volumes, wedges, poses):
        from oracle import oracle as O
        ref = O.Reference(reference, cfg.box / 2)
        return O.sva_align(ref, cfg, volumes, wedges, poses)[1]

    def average(self, cfg, chunks, members):
        from oracle import oracle as O
        n = cfg.box
        acc, cnt = np.zeros(O.accum_floats(n), np.float32), np.zeros(2, np.int64)
        for lo, hi, vols, wedges, poses, index in chunks:
            sel = np.where(members[lo:hi])[0]
            if len(sel):
                O.sva_insert(acc, cnt, cfg, np.ascontiguousarray(vols[sel]), wedges[sel], poses[sel], index[sel])
        if cnt.sum() == 0:
            return None, None, None, [0, 0]
        he, ho, avg, _ = O.finalize(acc, n, 1.0, FinalCfg(molecular_mass_kda=0.0, inner_radius=0.0, outer_radius=0.0, mask_falloff=0.0))
        return avg, (he if cnt[0] else None), (ho if cnt[1] else None), [int(cnt[0]), int(cnt[1])]


def two_structures(n, per_class, seed=3):
    """Sub-tomograms of two different structures (the phantom, and the phantom with a lobe removed), interleaved."""
    vol_a = synth.phantom(n)
    k = np.arange(n) - n // 2
    z, y, x = np.meshgrid(k, k, k, indexing="ij")
    vol_b = vol_a.copy()
    vol_b[(x > 1) & (y > -2)] *= 0.15
    _, va, pa, wa = synth.make_subtomograms(n, per_class, snr=1.0, vol=vol_a, seed=seed)
    _, vb, pb, wb = synth.make_subtomograms(n, per_class, snr=1.0, vol=vol_b, seed=seed + 10)
    vols = np.concatenate([va.numpy(), vb.numpy()]); poses = np.vstack([pa, pb]); wedges = np.vstack([wa, wb])
    truth = np.repeat([0, 1], per_class)
    perm = np.random.default_rng(seed).permutation(len(truth))
    return vol_a, vol_b, vols[perm], wedges[perm], poses[perm], truth[perm]


def chunks_of(vols, wedges, poses, chunk=5):
    index = np.arange(len(vols), dtype=np.int64)

    def gen():
        for lo in range(0, len(vols), chunk):
            hi = min(lo + chunk, len(vols))
            yield lo, hi, vols[lo:hi], wedges[lo:hi], poses[lo:hi], index[lo:hi]
    return gen


def blurred(v, sigma):
    from scipy.ndimage import gaussian_filter
    return gaussian_filter(v, sigma).astype(np.float32)


def test_multi_reference_classification_separates_two_structures_on_the_oracle():
    """From two poor class references (70 / 30 mixtures of the two structures, blurred) the driver sorts 24 noisy sub-tomograms with
    missing wedges into the two structures and returns class averages that resemble them; with one reference everything is one class."""
    n = 24
    va, vb, vols, wedges, poses, truth = two_structures(n, 12)
    cfg = SvaCfg.make(n, window=(9, 9, 9), window_sigma=2.0, highpass=(0.03, 0.01), lowpass=(0.3, 0.05), tol_angle=5.0, tol_shift=2.0)
    refs = [blurred(0.7 * va + 0.3 * vb, 1.0), blurred(0.3 * va + 0.7 * vb, 1.0)]
    classes, sc, avgs, its = sva.classify(chunks_of(vols, wedges, poses), len(vols), cfg, refs, OracleBackend(), iterations=4)
    assert np.array_equal(classes, truth) and sc.shape == (24, 2) and 1 <= its <= 4 and all(a is not None for a in avgs)
    k = np.arange(n) - n // 2
    z, y, x = np.meshgrid(k, k, k, indexing="ij")
    m = (x * x + y * y + z * z) < (0.4 * n) ** 2
    cc = lambda a, b: float(np.corrcoef(a[m], b[m])[0, 1])
    assert cc(avgs[0], va) > cc(avgs[0], vb) and cc(avgs[1], vb) > cc(avgs[1], va)
    assert cc(avgs[0], va) > cc(refs[0], va) - 0.02                       # the average of the class is at least as good as the reference it started from
    one, _, a1, _ = sva.classify(chunks_of(vols, wedges, poses), len(vols), cfg, refs[:1], OracleBackend(), iterations=2)
    assert (one == 0).all() and a1[0] is not None
    with pytest.raises(ValueError):
        sva.classify(chunks_of(vols, wedges, poses), len(vols), cfg, [], OracleBackend())
    # the settings that score: the search is switched off, the metric kept
    s = sva.score_cfg(cfg)
    assert s.tol_angle == 0 and s.tol_shift == 0 and abs(s.lowpass_cutoff - cfg.lowpass_cutoff) < 1e-9 and cfg.tol_angle == 5.0


def test_classification_without_references_from_random_starts_on_the_oracle():
    """classify_unsupervised: five random first assignments (neighbouring even / odd indices share a class, so every class has both
    half-averages), each iterated with the cross-validated driver; the partition with the best mean cross-validated score of a
    sub-volume against its own class is the true one (up to the naming of the classes), and restarts that ended in mixed classes score lower."""
    n = 24
    va, vb, vols, wedges, poses, truth = two_structures(n, 12)
    cfg = SvaCfg.make(n, window=(9, 9, 9), window_sigma=2.0, highpass=(0.03, 0.01), lowpass=(0.3, 0.05), tol_angle=5.0, tol_shift=2.0)
    classes, sc, avgs, its, obj, objs = sva.classify_unsupervised(chunks_of(vols, wedges, poses), len(vols), cfg, 2, OracleBackend(), restarts=5, iterations=10, seed=0)
    assert np.array_equal(classes, truth) or np.array_equal(classes, 1 - truth)
    assert len(objs) == 5 and obj == max(objs) and min(objs) < obj - 0.003 and all(a is not None for a in avgs)
    # the first assignment: pairs of neighbouring indices stay together, the classes are equally large
    start = sva.random_assignment(np.arange(24), 3, np.random.default_rng(1))
    assert (start[0::2] == start[1::2]).all() and sorted(np.bincount(start)) == [8, 8, 8]
    with pytest.raises(ValueError, match="ERROR"):
        sva.classify_unsupervised(chunks_of(vols[:6], wedges[:6], poses[:6]), 6, cfg, 2, OracleBackend())        # two classes need eight sub-volumes
    with pytest.raises(ValueError, match="ERROR"):
        sva.classify(chunks_of(vols, wedges, poses), len(vols), cfg, None, OracleBackend(), start=np.r_[np.zeros(23, int), 1])     # class 1 has no halves


@pytest.mark.gpu
def test_gpu_classification_without_references_equals_the_oracle_run():
    """The random-start driver through the HIP path: the same restarts (same seed) end in the same partition with the same objective."""
    n = 32
    va, vb, vols, wedges, poses, truth = two_structures(n, 12, seed=5)
    cfg = SvaCfg.make(n, window=(12, 12, 12), window_sigma=2.0, highpass=(0.03, 0.01), lowpass=(0.3, 0.05), tol_angle=5.0, tol_shift=2.0)
    co, _, _, _, oo, objs_o = sva.classify_unsupervised(chunks_of(vols, wedges, poses, 7), len(vols), cfg, 2, OracleBackend(), restarts=4, iterations=8, seed=2)
    cg, _, ag, _, og, objs_g = sva.classify_unsupervised(chunks_of(vols, wedges, poses, 7), len(vols), cfg, 2, sva.GpuBackend(0), restarts=4, iterations=8, seed=2)
    assert np.array_equal(co, cg) and abs(oo - og) < 2e-3 and np.abs(np.array(objs_o) - np.array(objs_g)).max() < 2e-3
    assert np.array_equal(cg, truth) or np.array_equal(cg, 1 - truth)
    assert all(a is not None for a in ag)


@pytest.mark.gpu
def test_gpu_classification_equals_the_oracle_run():
    """The same driver, references and data through the HIP path: the same classes, scores to the alignment tests' tolerance."""
    n = 32
    va, vb, vols, wedges, poses, truth = two_structures(n, 12, seed=5)
    cfg = SvaCfg.make(n, window=(12, 12, 12), window_sigma=2.0, highpass=(0.03, 0.01), lowpass=(0.3, 0.05), tol_angle=5.0, tol_shift=2.0)
    refs = [blurred(0.7 * va + 0.3 * vb, 1.0), blurred(0.3 * va + 0.7 * vb, 1.0)]
    co, so, ao, io = sva.classify(chunks_of(vols, wedges, poses, 7), len(vols), cfg, refs, OracleBackend(), iterations=3)
    cg, sg, ag, ig = sva.classify(chunks_of(vols, wedges, poses, 7), len(vols), cfg, refs, sva.GpuBackend(0), iterations=3)
    assert np.array_equal(co, cg) and io == ig and np.abs(so - sg).max() < 2e-3
    assert np.array_equal(cg, truth)
    for a, b in zip(ao, ag):
        assert np.abs(a - b).max() < 5e-3 * np.abs(a).max()


@pytest.mark.gpu
def test_sva_align_executable_classifies_under_protocol_mode_1(tmp_path):
    import os
    import subprocess
    import sys
    from pyp_amd.formats import mrc
    n = 32
    va, vb, vols, wedges, poses, truth = two_structures(n, 12, seed=7)
    tab = np.zeros((len(vols), 32)); names = []
    for k in range(len(vols)):
        tab[k, 0], tab[k, 1], tab[k, 2] = k + 1, wedges[k, 0], wedges[k, 1]
        tab[k, 12:28] = sva.pose_to_matrix(poses[k, :9], poses[k, 9:], tab[k, 9:12])
        names.append(f"TS_01_spk{k:04d}.rec")
        mrc.write(vols[k], str(tmp_path / names[-1]))
    sva.write_volumes(str(tmp_path / "d_volumes.txt"), tab, names)
    mrc.write(blurred(0.7 * va + 0.3 * vb, 1.0), str(tmp_path / "c0.mrc"))
    mrc.write(blurred(0.3 * va + 0.7 * vb, 1.0), str(tmp_path / "c1.mrc"))
    (tmp_path / "p.xml").write_text("""<config><general><mode>1</mode><metric><use_missing_wedge>1</use_missing_wedge></metric></general>
      <class><class_number_of_classes>2</class_number_of_classes><class_image_window_x>12</class_image_window_x><class_image_window_y>12</class_image_window_y>
      <class_image_window_z>12</class_image_window_z><class_image_window_sigma>2</class_image_window_sigma><class_high_pass_cutoff>.03</class_high_pass_cutoff>
      <class_high_pass_decay>.01</class_high_pass_decay><class_low_pass_cutoff>0.30</class_low_pass_cutoff><class_low_pass_decay>.05</class_low_pass_decay></class></config>""")
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bin", "sva_align")
    prefix = "d_iteration_002_level_2_average"
    r = subprocess.run([sys.executable, exe, "p.xml", "d_volumes.txt", "c0.mrc,c1.mrc", "out_volumes.txt", prefix], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0 and "SVA: Normal termination" in r.stdout and "Classified 24 sub-volumes into 2 classes" in r.stdout, r.stdout + r.stderr
    assert (tmp_path / (prefix + "_000.mrc")).exists() and (tmp_path / (prefix + "_001.mrc")).exists()
    cl = np.loadtxt(str(tmp_path / (prefix + "_classes.txt")), skiprows=1)
    assert cl.shape == (24, 4) and np.array_equal(cl[:, 1].astype(int), truth)
    out, _ = sva.read_volumes(str(tmp_path / "out_volumes.txt"))
    assert np.allclose(out[:, 12:28], tab[:, 12:28], atol=1e-5)                 # classification leaves the poses alone
    r = subprocess.run([sys.executable, exe, "p.xml", "d_volumes.txt", "c0.mrc,c1.mrc", "o2.txt"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode != 0 and "ERROR" in r.stdout and "average_prefix" in r.stdout
    # ONE reference: the classes are found without references (class_number_of_classes = 2 of the protocol, random first assignments)
    r = subprocess.run([sys.executable, exe, "p.xml", "d_volumes.txt", "c0.mrc", "o3.txt", "u_avg"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0 and "no references: 2 classes" in r.stdout and "SVA: Normal termination" in r.stdout, r.stdout + r.stderr
    cu = np.loadtxt(str(tmp_path / "u_avg_classes.txt"), skiprows=1)[:, 1].astype(int)
    assert np.array_equal(cu, truth) or np.array_equal(cu, 1 - truth)
    assert (tmp_path / "u_avg_000.mrc").exists() and (tmp_path / "u_avg_001.mrc").exists()
