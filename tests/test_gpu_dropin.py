"""End-to-end run of the drop-in executables on the GPU, driven exactly like PYP drives the binaries:
shell here-docs of positional answers, particle ranges fanned out, range files merged, dumps merged
(src/pyp/refine/frealign/frealign.py:3014-3193, :1622-1835, :1838-1903, :1910-2175)."""
import os
import subprocess

import numpy as np
import pytest

from pyp_amd import synth
from pyp_amd.formats import cistem, mrc, parfile

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "bin")
N, PX, M = 64, 2.0, 60


def run(prog, script, cwd, log):
    cmd = f"{BIN}/{prog} << eot >> {log} 2>&1\n{script}eot\n"
    return subprocess.run(cmd, shell=True, cwd=cwd).returncode


def refine_script(first, last, global_search, inp="p_r01.cistem", out=None):
    rng = "%07d_%07d" % (first, last)
    out = out or f"p_r01_{rng}.cistem"
    yn = lambda b: "yes" if b else "no"
    lines = ["p_stack.mrc", inp, "null", "p_r01.mrc", "statistics_r01.txt", "no", "no", f"p_r01_match.mrc_{rng}", out,
             f"p_r01_{rng}_changes.cistem", "C1", first, last, 1, PX, 300, 0, 0.4 * N * PX, 0, PX * N / 24, 30.0, 8.0,
             0.4 * N * PX, PX * N / 10, 15.0, 20, 12.0, 12.0, 0, 0, 0, 0, 500, 50.0, 1, yn(global_search), "yes",
             "yes", "yes", "yes", "yes", "yes", "no", "no", "no", "yes", "no", "no", "no", "no"]
    return "\n".join(str(x) for x in lines) + "\n"


@pytest.fixture(scope="module")
def project(tmp_path_factory):
    d = tmp_path_factory.mktemp("proj")
    vol, stack, rows = synth.make_dataset(N, M, pixel=PX, snr=0.2)
    mrc.write(stack.numpy(), str(d / "p_stack.mrc"), pixel_size=PX)
    mrc.write(vol, str(d / "p_r01.mrc"), pixel_size=PX)
    start = cistem.default_rows(M, PX, 300.0, 2.7, 0.07)
    for c in ("DEFOCUS_1", "DEFOCUS_2", "DEFOCUS_ANGLE"):
        start[:, cistem.COL[c]] = rows[:, cistem.COL[c]]
    cistem.write_parameters(str(d / "p_r01.cistem"), start)
    return d, vol, stack.numpy(), rows, start


def test_refine3d_ranges_then_merge(project):
    d, vol, imgs, truth, start = project
    for first, last in ((1, 31), (32, 60)):                 # ranges like local_run.create_split_commands
        assert run("refine3d", refine_script(first, last, True), d, "refine.log") == 0
    assert "Refine3D: Normal termination" in open(d / "refine.log").read()
    files = sorted(str(p) for p in d.glob("p_r01_0*_0*.cistem") if "changes" not in p.name)
    assert len(files) == 2
    merged = cistem.merge_parameters(files)
    assert merged.shape == (M, 32) and list(merged[:, 0]) == list(range(1, M + 1))
    cistem.write_parameters(str(d / "p_r01_refined.cistem"), merged)
    # same numbers as the library called directly on the whole stack
    from pyp_amd import host
    from pyp_amd.abi import RefineCfg
    cfg = RefineCfg.make(box=N, pixel_size=PX, molecular_mass_kda=300, mask_radius=0.4 * N * PX, res_high=PX * N / 24, res_signed_cc=30.0,
                         search_mask_radius=0.4 * N * PX, res_search=PX * N / 10, search_range_x=12.0, search_range_y=12.0)
    direct = host.Reference(vol, N / 2).refine(cfg, imgs, start)
    on_disk = direct.copy()
    for j, (_, _, code) in enumerate(cistem.COLUMNS):
        on_disk[:, j] = direct[:, j].astype(np.float32) if code == cistem.FLOAT else direct[:, j]
    assert synth.angular_error_deg(merged, on_disk).max() < 0.05
    ang = synth.angular_error_deg(merged, truth)
    assert np.median(ang) < 3.0
    ch = cistem.read_parameters(str(d / "p_r01_0000001_0000031_changes.cistem"))
    assert ch.shape == (31, 32) and np.allclose(ch[:, 14], merged[:31, 14] - 0.5, atol=1e-3)


def test_refine3d_par_surface_local(project):
    d, vol, imgs, truth, start = project
    pert = synth.perturb_rows(truth, 2.0, 1.0, PX)
    parfile.write(str(d / "q_r01_02.par"), parfile.cistem_to_par(pert, parfile.NEW), version=parfile.NEW)
    lines = ["p_stack.mrc", "q_r01_02.par", "p_r01.mrc", "statistics_r01.txt", "no", "q_match.mrc_0000001_0000020",
             "q_r01_02.par_0000001_0000020", "/dev/null", "C1", 1, 20, PX, 300.0, 2.7, 0.07, 300.0, 0.4 * N * PX, 0, PX * N / 24, 30.0, 8,
             0.4 * N * PX, PX * N / 24, 200, 20, 0, 0, 0, 0, 0, 0, 500.0, 50.0, 1, "no", "yes", "yes", "yes", "yes", "yes", "yes",
             "no", "no", "no", "no"]
    assert run("refine3d", "\n".join(str(x) for x in lines) + "\n", d, "refine_par.log") == 0
    data, version, ext, pro, epi = parfile.read(str(d / "q_r01_02.par_0000001_0000020"))
    assert data.shape == (20, 16) and version == parfile.NEW
    back = parfile.par_to_cistem(data, version, PX, 300.0, 2.7, 0.07)
    assert np.median(synth.angular_error_deg(back, truth[:20])) < np.median(synth.angular_error_deg(pert[:20], truth[:20]))


def test_reconstruct_merge_pipeline(project):
    d, vol, imgs, truth, start = project
    used = truth.copy()
    used[:, cistem.COL["SCORE"]] = 20.0
    used[5, cistem.COL["OCCUPANCY"]] = 0.0
    cistem.write_parameters(str(d / "p_r01_used.cistem"), used)

    def rec_script(first, last, count):
        lines = ["p_stack.mrc", "p_r01_used.cistem", "null", "p_r01.mrc", "p_r01_map1.mrc", "p_r01_map2.mrc", "output.mrc",
                 f"p_r01_n{first}.res", "C1", first, last, PX, 300, 0, PX * N / 2, 2 * PX, 0, 2.0, "no", 0, -1, "no", 0, 1, 1,
                 "yes", "yes", "no", "no", "no", "yes", "yes", "no", "no", "no", "yes",
                 f"{d}/p_r01_map1_n{count}.mrc", f"{d}/p_r01_map2_n{count}.mrc", 1]
        return "\n".join(str(x) for x in lines) + "\n"
    for count, (first, last) in enumerate(((1, 20), (21, 40), (41, 60)), start=1):
        assert run("reconstruct3d", rec_script(first, last, count), d, "rec.log") == 0
    log = open(d / "rec.log").read()
    assert log.count("Reconstruct3D: Normal termination") == 3 and "ERROR" not in log
    lm = "\n".join([f"{d}/m_map1_n1.mrc", f"{d}/m_map2_n1.mrc", f"{d}/p_r01_map1_n.mrc", f"{d}/p_r01_map2_n.mrc", "3"]) + "\n"
    assert run("local_merge3d", lm, d, "lmerge.log") == 0
    mg = "\n".join(["p_half1.mrc", "p_half2.mrc", "p.mrc", "p_statistics.txt", "300", "0", str(0.45 * N * PX),
                    f"{d}/m_map1_n.mrc", f"{d}/m_map2_n.mrc", "1"]) + "\n"
    assert run("merge3d", mg, d, "merge.log") == 0
    A = open(d / "merge.log").read()
    assert "Merge3D: Normal termination" in A
    from io import StringIO
    Afsc = A[A.find("Rec_SSNR") + 9: A.find("Merge3D: Normal termination") - 3]
    rows = len(Afsc.split("\n"))
    tab = np.genfromtxt(StringIO(Afsc), delimiter=[5, 8, 10, 10, 10, 10, 10]).reshape((rows, 7))
    assert rows == N // 2 - 1 and np.allclose(tab[:, 1], np.round(N * PX / tab[:, 0], 2)) and (tab[:4, 3] > 0.8).all()
    st = np.loadtxt(str(d / "p_statistics.txt"), comments=["C"])
    assert st.shape == (N // 2 - 1, 7)
    m = mrc.read(str(d / "p.mrc"))
    h = mrc.read_header(str(d / "p.mrc"))
    assert m.shape == (N, N, N) and abs(h["pixel_size"] - PX) < 1e-5

    def cc(a, b):
        a, b = a - a.mean(), b - b.mean()
        return float((a * b).sum() / np.sqrt((a * a).sum() * (b * b).sum()))
    assert cc(m, vol) > 0.85
    assert cc(mrc.read(str(d / "p_half1.mrc")), mrc.read(str(d / "p_half2.mrc"))) > 0.8
    # next iteration the way PYP runs it by default (refine_fssnr true): the merged map as the reference, "use statistics" yes
    mrc.write(m, str(d / "p_r02.mrc"), pixel_size=PX)
    s = refine_script(1, 30, True, out="p_r02_0000001_0000030.cistem").split("\n")
    s[3], s[4], s[5] = "p_r02.mrc", "p_statistics.txt", "yes"
    assert run("refine3d", "\n".join(s), d, "refine2.log") == 0 and "Normal termination" in open(d / "refine2.log").read()
    r2 = cistem.read_parameters(str(d / "p_r02_0000001_0000030.cistem"))
    assert np.median(synth.angular_error_deg(r2, truth[:30])) < 3.0


def test_unsupported_options_fail_loudly(project):
    d, vol, imgs, truth, start = project
    s = refine_script(1, 10, True, out="bad_out.cistem").split("\n")
    s[6] = "yes"                                      # use priors
    assert run("refine3d", "\n".join(s), d, "bad.log") != 0
    assert "ERROR" in open(d / "bad.log").read() and not (d / "bad_out.cistem").exists()
    s = refine_script(1, 10, True, out="def_out.cistem").split("\n")
    s[44] = "yes"                                     # refine defocus: supported, columns 6 / 7 move on the 50 A grid
    assert run("refine3d", "\n".join(s), d, "def.log") == 0 and (d / "def_out.cistem").exists()
    got = cistem.read_parameters(str(d / "def_out.cistem"))
    delta = got[:, 6] - start[:10, 6]
    assert np.allclose(delta, np.round(delta / 50.0) * 50.0, atol=1e-2) and np.allclose(got[:, 7] - start[:10, 7], delta, atol=1e-2)


@pytest.mark.gpu
def test_two_class_round_assigns_particles_to_their_map():
    """SURVEY 8f-4 (classification occupancies): particles projected from map A and from map B are refined against both
    references on the GPU; the LOGP-based occupancy update (pinned on the CPU side by the reference's golden) must send
    each particle to its own class."""
    import numpy as np
    from pyp_amd import classify, host, synth
    from pyp_amd.abi import RefineCfg
    n, px, m = 64, 2.0, 24
    volA = synth.phantom(n)
    volB = synth.phantom(n, seed=4242)
    _, sA, rA = synth.make_dataset(n, m, pixel=px, snr=0.2, vol=volA)
    _, sB, rB = synth.make_dataset(n, m, pixel=px, snr=0.2, vol=volB, seed_poses=99, seed_noise=98)
    stack = np.concatenate([sA.numpy(), sB.numpy()])
    rows = np.concatenate([rA, rB]); rows[:, 0] = np.arange(1, 2 * m + 1); rows[:, 11] = 50.0
    start = synth.perturb_rows(rows, angle_sigma=2.0, shift_sigma_px=0.5, pixel=px)
    cfg = RefineCfg.make(box=n, pixel_size=px, mask_radius=0.4 * n * px, res_high=px * n / 24.0, global_search=0, res_signed_cc=30.0)
    refs = [host.Reference(volA, n / 2), host.Reference(volB, n / 2)]
    out = classify.refine_classes(refs, cfg, stack, [start, start])
    occ = np.stack([t[:, 11] for t in out])
    assert np.allclose(occ.sum(axis=0), 100.0, atol=1e-6)
    assert (occ[0, :m] > 90).mean() > 0.9 and (occ[1, m:] > 90).mean() > 0.9
    assert np.array_equal(out[0][:, 13], out[1][:, 13])                    # one SIGMA per particle


@pytest.mark.gpu
def test_in_memory_iterations_improve_the_map():
    """pyp_amd.pipeline.iteration twice from perturbed poses and a low-pass reference: poses approach the truth, the second
    map correlates better with the phantom than the first, FSC at mid resolution goes up."""
    import numpy as np
    from pyp_amd import pipeline, synth
    from pyp_amd.abi import RefineCfg
    n, px, m = 64, 2.0, 400
    vol, stack, truth = synth.make_dataset(n, m, pixel=px, snr=0.3)
    imgs = stack.numpy()
    start = synth.perturb_rows(truth, angle_sigma=4.0, shift_sigma_px=1.0, pixel=px)
    k = np.fft.fftfreq(n); kk = np.sqrt(k[:, None, None] ** 2 + k[None, :, None] ** 2 + k[None, None, :] ** 2)
    ref0 = np.real(np.fft.ifftn(np.fft.fftn(vol) * (kk < 0.12))).astype(np.float32)       # 16 A low-pass start
    cfg = RefineCfg.make(box=n, pixel_size=px, mask_radius=0.4 * n * px, res_high=px * n / 20.0, global_search=0, res_signed_cc=30.0)
    it1 = pipeline.iteration(ref0, imgs, start, cfg, pixel_size=px, molecular_mass_kda=300.0, keep_fraction=0.9)
    it2 = pipeline.iteration(it1["filtered"], imgs, it1["rows"], cfg, pixel_size=px, molecular_mass_kda=300.0, keep_fraction=0.9)
    e0 = np.median(synth.angular_error_deg(start, truth)); e1 = np.median(synth.angular_error_deg(it1["rows"], truth))
    e2 = np.median(synth.angular_error_deg(it2["rows"], truth))
    assert e1 < 0.6 * e0 and e2 <= e1 + 0.05
    cc = lambda a: float(np.corrcoef(a.ravel(), vol.ravel())[0, 1])
    assert cc(it2["filtered"]) >= cc(it1["filtered"]) - 1e-3 and cc(it1["filtered"]) > cc(ref0) - 0.05
    assert abs((it1["rows"][:, 11] > 0).mean() - 0.9) < 0.02                      # selection kept 90 %
    assert it2["stats"][10, 3] >= it1["stats"][10, 3] - 0.02                      # FSC at shell 11
