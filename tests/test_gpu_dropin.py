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


def test_refine3d_writes_matching_projections(project):
    """Answer 43 "calculate matching projections" = yes (refine_fmatch, frealign.py:3929-3931): answer 8's file holds one section
    per particle of the range, the model at the refined pose; it overlays the particle it belongs to."""
    d, vol, imgs, truth, start = project
    lines = refine_script(5, 16, True, out="m.cistem").splitlines()
    lines[42] = "yes"                                               # answer 43
    assert run("refine3d", "\n".join(lines) + "\n", d, "match.log") == 0, open(d / "match.log").read()[-2000:]
    m = mrc.read(str(d / "p_r01_match.mrc_0000005_0000016"))
    assert m.shape == (12, N, N)
    for a, b in zip(m, imgs[4:16]):
        a0, b0 = a - a.mean(), b - b.mean()
        assert (a0 * b0).sum() / np.sqrt((a0 * a0).sum() * (b0 * b0).sum()) > 0.25      # SNR 0.2: cc with the noisy particle ~ 0.4
    other = (m[0] - m[0].mean()) * (imgs[20] - imgs[20].mean())
    assert abs(other.sum()) / np.sqrt(((m[0] - m[0].mean()) ** 2).sum() * ((imgs[20] - imgs[20].mean()) ** 2).sum()) < 0.2


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
    # the range is streamed from the stack file through two pinned buffers: a 7-image chunk gives the same bytes
    env = dict(os.environ, PPM_IO_CHUNK="7")
    cmd = f"{BIN}/refine3d << eot >> refine_chunk.log 2>&1\n{refine_script(1, 31, True, out='chunked.cistem')}eot\n"
    assert subprocess.run(cmd, shell=True, cwd=d, env=env).returncode == 0
    assert np.array_equal(cistem.read_parameters(str(d / "chunked.cistem")), cistem.read_parameters(files[0]))


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
    s[6] = "yes"                                      # use priors without the statistics file of answer 3 ("null")
    assert run("refine3d", "\n".join(s), d, "bad.log") != 0
    assert "ERROR" in open(d / "bad.log").read() and "statistics" in open(d / "bad.log").read() and not (d / "bad_out.cistem").exists()
    # with `<name>_stat.cistem` (means, variances: src/pyp_main.py:2667-2674) the answer is honoured
    stat = np.vstack([truth.mean(axis=0), truth.var(axis=0)])
    cistem.write_parameters(str(d / "p_r01_stat.cistem"), stat)
    s = refine_script(1, 10, True, out="prior_out.cistem").split("\n")
    s[2], s[6] = "p_r01_stat.cistem", "yes"
    assert run("refine3d", "\n".join(s), d, "prior.log") == 0, open(d / "prior.log").read()[-1500:]
    log = open(d / "prior.log").read()
    assert "priors: mean psi theta phi x y" in log and "Refine3D: Normal termination" in log
    got = cistem.read_parameters(str(d / "prior_out.cistem"))
    assert np.median(synth.angular_error_deg(got, truth[:10])) < 3.0
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


def test_reconstruct3d_with_dose_weighting_writes_the_side_files(project):
    """The five-line dose-weighting answer (frealign.py:1731-1753) is honoured: weights.txt / scores.txt appear next to the
    dumps (particle_cspt.py:808-810 plots them) and the dump differs from the unweighted one."""
    d, vol, imgs, truth, start = project
    used = truth.copy()
    used[:, cistem.COL["TIND"]] = np.arange(M) % 5
    used[:, cistem.COL["SCORE"]] = 25.0 - 4.0 * used[:, cistem.COL["TIND"]]
    cistem.write_parameters(str(d / "dw_r01_used.cistem"), used)
    from pyp_amd.surface import cli as pcli

    def script(dose, tag):
        lines = ["p_stack.mrc", "dw_r01_used.cistem", "null", "p_r01.mrc", "dw_map1.mrc", "dw_map2.mrc", "output.mrc", f"dw_r01_n1.res", "C1", 1, M, PX, 300, 0,
                 PX * N / 2, 2 * PX, 0, 2.0, "no", 0, -1] + dose + [0, 1, 1, "yes", "no", "no", "no", "no", "yes", "no", "no", "no", "no", "yes",
                 f"{d}/dw{tag}_map1_n1.mrc", f"{d}/dw{tag}_map2_n1.mrc", 1]
        return "\n".join(str(x) for x in lines) + "\n"
    for f in ("weights.txt", "scores.txt"):
        if (d / f).exists():
            os.remove(d / f)
    assert run("reconstruct3d", script(["no"], "a"), d, "rec_dw.log") == 0 and not (d / "weights.txt").exists()
    assert run("reconstruct3d", script(["yes", "/scratch/not_provided", "yes", 4, 0.75], "b"), d, "rec_dw.log") == 0
    log = open(d / "rec_dw.log").read()
    assert log.count("Reconstruct3D: Normal termination") == 2 and "ERROR" not in log and "dose weighting: 5 exposures" in log
    A = np.loadtxt(str(d / "weights.txt"))
    assert A.shape[0] == 5 * (N + 1) * (N // 2) and A.max() <= 1.0 + 1e-6 and A.min() > 0
    sc = np.loadtxt(str(d / "scores.txt"))
    assert np.allclose(sc, np.array([25.0, 21.0, 17.0, 13.0, 9.0]) / 25.0, atol=1e-5)
    a = pcli.read_dump(str(d / "dwa_map1_n1.mrc"))[3]; b = pcli.read_dump(str(d / "dwb_map1_n1.mrc"))[3]
    wa, wb = a.reshape(-1, 3)[:, 2].sum(), b.reshape(-1, 3)[:, 2].sum()
    assert 0.3 * wa < wb < 0.95 * wa                                        # weaker exposures weigh less at high resolution


def test_reconstruct3d_likelihood_blurring_and_crop_answers(project):
    """"likelihood blurring" = yes (reconstruct_lblur, frealign.py:1772, :1817) inserts every particle at the in-plane rotations
    around its pose weighted by their likelihood against the reference; "crop" = yes is accepted with a note.  With sharp
    likelihoods (high SNR, true poses) the blurred map stays close to the plain one."""
    d, vol, imgs, truth, start = project
    used = truth.copy()
    used[:, cistem.COL["SCORE"]] = 20.0
    cistem.write_parameters(str(d / "lb_r01_used.cistem"), used)
    from pyp_amd.surface import cli as pcli

    def script(crop, blur, tag):
        lines = ["p_stack.mrc", "lb_r01_used.cistem", "null", "p_r01.mrc", "lb_map1.mrc", "lb_map2.mrc", "output.mrc", "lb_r01_n1.res", "C1", 1, M, PX, 300, 0,
                 PX * N / 2, 2 * PX, 0, 2.0, "no", 0, -1, "no", 0, 1, 1, "yes", "no", "no", "no", crop, "yes", "no", "no", blur, "no", "yes",
                 f"{d}/lb{tag}_map1_n1.mrc", f"{d}/lb{tag}_map2_n1.mrc", 1]
        return "\n".join(str(x) for x in lines) + "\n"
    assert run("reconstruct3d", script("no", "no", "a"), d, "rec_lb.log") == 0
    assert run("reconstruct3d", script("yes", "yes", "b"), d, "rec_lb.log") == 0
    log = open(d / "rec_lb.log").read()
    assert log.count("Reconstruct3D: Normal termination") == 2 and "ERROR" not in log
    assert "crop = yes has no effect" in log and "likelihood blurring: rows 1..60" in log
    ba, pa, ca, a = pcli.read_dump(str(d / "lba_map1_n1.mrc")); bb, pb, cb, b = pcli.read_dump(str(d / "lbb_map1_n1.mrc"))
    assert ca == cb                                                        # particles are counted once, not once per rotation
    wa, wb = a.reshape(-1, 3)[:, 2], b.reshape(-1, 3)[:, 2]
    assert abs(wb.sum() / wa.sum() - 1.0) < 0.05                           # the weights of a particle's rotations sum to one
    cc = np.corrcoef(a.reshape(-1, 3)[:, 0], b.reshape(-1, 3)[:, 0])[0, 1]
    assert 0.7 < cc < 0.9999                                                # close to the plain insertion, but not the same


def test_refine3d_verbatim_default_script_matches_oracle(tmp_path):
    """The script exactly as PYP's default iteration writes it — the string frealign.mrefine_version itself produced
    (tests/golden/golden_r03.json "dropin_box64": global = yes, local = no, 20 hits to refine, D7, 143 particles;
    frealign.py:3866-3871, :3918-3994) — fed through the shell; the
    refined poses are sub-grid and agree with the oracle run on the parsed answers."""
    import io
    from oracle import oracle
    from pyp_amd.surface import cli, prompts
    from test_surface_cpu import REFINE_CISTEM
    n, px, m = 64, 4.32, 143
    work = tmp_path / "swarm"
    work.mkdir()
    vol = synth.phantom_sym(n, oracle.symmetry_ops("D7"))
    _, stack, truth = synth.make_dataset(n, m, pixel=px, snr=0.3, vol=vol, particle_rad_frac=85.0 / (n * px))
    mrc.write(stack.numpy(), str(tmp_path / "t20s_stack.mrc"), pixel_size=px)
    mrc.write(vol, str(work / "t20s_r01_01.mrc"), pixel_size=px)
    start = cistem.default_rows(m, px, 300.0, 2.7, 0.07)
    for c in ("DEFOCUS_1", "DEFOCUS_2", "DEFOCUS_ANGLE"):
        start[:, cistem.COL[c]] = truth[:, cistem.COL[c]]
    cistem.write_parameters(str(work / "t20s_r01_01.cistem"), start)
    assert run("refine3d", REFINE_CISTEM.replace("eot\n", ""), work, "msearch.log") == 0
    log = open(work / "msearch.log").read()
    assert "Refine3D: Normal termination" in log and "ERROR" not in log
    got = cistem.read_parameters(str(work / "t20s_r01_01_0000001_0000143.cistem"))
    assert got.shape == (m, 32)
    d = prompts.parse_refine3d(prompts.read_answers(io.StringIO(REFINE_CISTEM)))
    cfg = cli.refine_cfg_from_answers(d, n)
    assert cfg.global_search == 1 and cfg.local_refine == 0 and cfg.top_hits == 20
    rin = cistem.read_parameters(str(work / "t20s_r01_01.cistem"))
    want, counts = oracle.refine_batch(oracle.Reference(vol, n / 2), cfg, stack.numpy(), rin)
    assert counts[1] == 20 * 2 * 12 + 1                       # 20 hits x 2 compass iterations x 12 scores, one final score
    # D7 reference: a grid point on the edge of the asymmetric unit and its symmetry mate score the same up to round-off,
    # so poses are compared modulo the point group
    d7 = oracle.symmetry_ops("D7")
    assert synth.angular_error_deg(want, got, d7).max() < 0.1 and synth.shift_error_px(want, got, px).max() < 0.5
    assert np.abs(want[:, 14] - got[:, 14]).max() < 0.05
    assert np.median(synth.angular_error_deg(got, truth, d7)) < 2.5
    # sub-grid: most poses sit off the 20 degree grid the search walks
    on_grid = np.isclose(got[:, 2] % 18.0, 0.0, atol=1e-3) | np.isclose(got[:, 2] % 18.0, 18.0, atol=1e-3)
    assert on_grid.mean() < 0.1
    # answer 26 is honoured: refining only the first hit finds worse poses on average
    s1 = REFINE_CISTEM.replace("eot\n", "").split("\n")
    s1[25] = "1"
    s1[8] = "one_hit.cistem"
    assert run("refine3d", "\n".join(s1), work, "msearch1.log") == 0
    one = cistem.read_parameters(str(work / "one_hit.cistem"))
    assert (got[:, 14] > one[:, 14] + 1e-3).sum() > 2 * (got[:, 14] < one[:, 14] - 1e-3).sum() and got[:, 14].mean() > one[:, 14].mean()


def test_config1_local_refinement_1k_128_par_surface_matches_oracle(tmp_path):
    """BASELINE.json configs[0] as SURVEY.md 8(d) states it: 1 000 particles of 128^2, one 128^3 reference, local refinement
    (refine_mode = 1 -> global no / local yes) from poses perturbed by N(0, 2 deg) / N(0, 1 px), rhref 8 A (r = 16 px),
    through the .par surface (answer order of src/pyp/system/wrapper_functions.py:512-561); HIP path vs oracle."""
    import io
    import torch
    from oracle import oracle
    from pyp_amd.surface import cli, prompts
    n, px, m = 128, 1.0, 1000
    vol, stack, truth = synth.make_dataset(n, m, pixel=px, snr=0.05, device="cuda" if torch.cuda.is_available() else "cpu")
    imgs = stack.cpu().numpy()
    mrc.write(imgs, str(tmp_path / "c1_stack.mrc"), pixel_size=px)
    mrc.write(vol, str(tmp_path / "c1_r01_01.mrc"), pixel_size=px)
    pert = synth.perturb_rows(truth, 2.0, 1.0, px)
    parfile.write(str(tmp_path / "c1_r01_02.par"), parfile.cistem_to_par(pert, parfile.NEW), version=parfile.NEW)
    lines = ["c1_stack.mrc", "c1_r01_02.par", "c1_r01_01.mrc", "statistics_r01.txt", "no", "c1_r01_02_match.mrc_0000001_0001000",
             "c1_r01_02.par_0000001_0001000", "/dev/null", "C1", 1, m, px, 300.0, 2.7, 0.07, 400.0, 0.32 * n * px, 100.0, 8.0, 30.0, 8,
             0.48 * n * px, 8.0, 200, 20, 0, 0, 0, 0, 0, 0, 500.0, 50.0, 1, "no", "yes", "yes", "yes", "yes", "yes", "yes",
             "no", "no", "no", "no"]
    script = "\n".join(str(x) for x in lines) + "\n"
    assert run("refine3d", script, tmp_path, "c1.log") == 0
    assert "Refine3D: Normal termination" in open(tmp_path / "c1.log").read()
    out, version, ext, _, _ = parfile.read(str(tmp_path / "c1_r01_02.par_0000001_0001000"))
    assert out.shape == (m, 16) and version == parfile.NEW
    got = parfile.par_to_cistem(out, version, px, 300.0, 2.7, 0.07)
    # the oracle starts from the same (two-decimal) text the executable read
    inp, _, _, _, _ = parfile.read(str(tmp_path / "c1_r01_02.par"))
    rin = parfile.par_to_cistem(inp, parfile.NEW, px, 300.0, 2.7, 0.07)
    d = prompts.parse_refine3d(prompts.read_answers(io.StringIO(script)))
    cfg = cli.refine_cfg_from_answers(d, n)
    assert cfg.global_search == 0 and cfg.local_refine == 1 and abs(n * px / cfg.res_high - 16.0) < 1e-6
    want, counts = oracle.refine_batch(oracle.Reference(vol, n / 2), cfg, imgs, rin)
    assert counts[0] == 0 and counts[1] == 9 * 12 + 1
    ang, shf = synth.angular_error_deg(want, got), synth.shift_error_px(want, got, px)
    assert ang.max() < 0.1 and shf.max() < 0.5, (ang.max(), shf.max())          # north_star tolerance (text rounding 0.005 included)
    assert np.abs(want[:, 14] - got[:, 14]).max() < 0.02
    assert np.median(synth.angular_error_deg(got, truth)) < np.median(synth.angular_error_deg(rin, truth))
    assert np.array_equal(out[:, 6], inp[:, 6]) and np.array_equal(out[:, 7], inp[:, 7])      # MAG and FILM carried over


def test_refine3d_par_surface_keeps_extended_columns(project):
    """A 45-column (extended NEW) input comes back with its MAG and its 29 trailing columns untouched."""
    d, vol, imgs, truth, start = project
    pert = synth.perturb_rows(truth[:10], 2.0, 1.0, PX)
    base = parfile.cistem_to_par(pert, parfile.NEW, mag=12345.0)
    ext = np.arange(10 * 29, dtype=np.float64).reshape(10, 29) / 8.0
    ext[:, 0] = np.arange(10); ext[:, 3] = np.arange(10) % 3
    parfile.write(str(d / "x_r01_02.par"), np.hstack([base, ext]), version=parfile.NEW, extended=True)
    lines = ["p_stack.mrc", "x_r01_02.par", "p_r01.mrc", "statistics_r01.txt", "no", "x_match.mrc_0000001_0000010",
             "x_r01_02.par_0000001_0000010", "/dev/null", "C1", 1, 10, PX, 300.0, 2.7, 0.07, 300.0, 0.4 * N * PX, 0, PX * N / 24, 30.0, 8,
             0.4 * N * PX, PX * N / 24, 200, 20, 0, 0, 0, 0, 0, 0, 500.0, 50.0, 1, "no", "yes", "yes", "yes", "yes", "yes", "yes",
             "no", "no", "no", "no"]
    assert run("refine3d", "\n".join(str(x) for x in lines) + "\n", d, "refine_parx.log") == 0
    data, version, extended, _, _ = parfile.read(str(d / "x_r01_02.par_0000001_0000010"))
    inp, _, _, _, _ = parfile.read(str(d / "x_r01_02.par"))
    assert data.shape == (10, 45) and extended and version == parfile.NEW
    assert np.array_equal(data[:, 16:], inp[:, 16:]) and np.all(data[:, 6] == 12345.0)


def test_bench_two_ranks_share_one_gpu_over_gloo():
    """bench.py --gpus 2 launches two ranks itself; here both use device 0 (PPM_FORCE_DEVICE) and rendezvous over gloo, so the
    sharded refinement and the reduced reconstruction run on the one-GPU box.  Small sizes; the line must say n_gpus = 2."""
    import json
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PPM_FORCE_DEVICE="0", PPM_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--box", "64",
                        "--band", "24", "--particles", "512", "--recon-particles", "1024", "--no-cpu"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["particles_per_gpu"] == 512
    assert d["roofline"]["bound"] == "valu_fp32" and 0 < d["roofline"]["frac"] <= 1
    rec = d["reconstruct"]
    assert rec["n_gpus"] == 2 and rec["value"] > 0 and 0 < rec["roofline"]["frac"] <= 1 and rec["map_cc_vs_truth"] > 0.5


def test_native_reconstruct3d_equals_the_python_implementation(project, monkeypatch):
    """bin/reconstruct3d is the compiled fast path (pyp_amd/csrc/reconstruct3d_main.cpp) and hands everything it does not cover to
    bin/reconstruct3d.py before the GPU is touched: the same range through both (PPM_NATIVE=0 forces the hand-over) must give the same
    dump files — with score weighting, the defocus regression of the scores, a statistics file and PIND splitting switched on —
    and an input the fast path does not take (scattered positions) must still work."""
    d, vol, imgs, truth, start = project
    rows = truth.copy()
    rng = np.random.default_rng(5)
    rows[:, cistem.COL["SCORE"]] = 20.0 + rng.normal(0, 3, len(rows)) + 1e-4 * (rows[:, cistem.COL["DEFOCUS_1"]] - rows[:, cistem.COL["DEFOCUS_1"]].mean())
    rows[7, cistem.COL["OCCUPANCY"]] = 0.0
    rows[:, cistem.COL["PIND"]] = np.arange(len(rows)) // 2
    cistem.write_parameters(str(d / "n_r01_used.cistem"), rows)
    stat = np.vstack([rows.mean(axis=0), rows.var(axis=0)])
    cistem.write_parameters(str(d / "n_r01_stat.cistem"), stat)

    def script(tag, params="n_r01_used.cistem", stats="n_r01_stat.cistem", first=3, last=52):
        lines = ["p_stack.mrc", params, stats, "p_r01.mrc", "n_map1.mrc", "n_map2.mrc", "output.mrc", f"n_{tag}.res", "C2", first, last, PX, 300, 0,
                 PX * N / 2, 2 * PX, 0, 2.0, "yes", 0, -1, "no", 12.0, 1, 1, "yes", "yes", "no", "no", "yes", "yes", "yes", "no", "no", "no", "yes",
                 f"{d}/n_{tag}_map1_n1.mrc", f"{d}/n_{tag}_map2_n1.mrc", 1]
        return "\n".join(str(x) for x in lines) + "\n"
    assert run("reconstruct3d", script("nat"), d, "rec_nat.log") == 0
    monkeypatch.setenv("PPM_NATIVE", "0")
    assert run("reconstruct3d", script("py"), d, "rec_py.log") == 0
    monkeypatch.delenv("PPM_NATIVE")
    ln, lp = open(d / "rec_nat.log").read(), open(d / "rec_py.log").read()
    assert "libpypmatch, native" in ln and "libpypmatch, native" not in lp and "Reconstruct3D: Normal termination" in ln and "Reconstruct3D: Normal termination" in lp
    ins = [ln_ for ln_ in ln.splitlines() if ln_.startswith("Inserted")][0].split(" in ")[0]
    assert ins == [l_ for l_ in lp.splitlines() if l_.startswith("Inserted")][0].split(" in ")[0]
    for k in (1, 2):
        a, b = open(d / f"n_nat_map{k}_n1.mrc", "rb").read(), open(d / f"n_py_map{k}_n1.mrc", "rb").read()
        assert a[:24] == b[:24]
        fa, fb = np.frombuffer(a, "<f4", offset=24), np.frombuffer(b, "<f4", offset=24)
        assert np.abs(fa - fb).max() <= 2e-5 * np.abs(fb).max()          # the score regression's slope is summed in another order
    assert open(d / "n_nat.res").read() == open(d / "n_py.res").read()
    # scattered positions: not the fast path's business, the same executable still serves them
    sc = rows[[3, 9, 10, 30, 31, 44]].copy()
    cistem.write_parameters(str(d / "n_r01_scattered.cistem"), sc)
    assert run("reconstruct3d", script("sc", params="n_r01_scattered.cistem", stats="null", first=1, last=60), d, "rec_sc.log") == 0
    lsc = open(d / "rec_sc.log").read()
    assert "Reconstruct3D: Normal termination" in lsc and "Inserted 6 of 6" in lsc
    # a missing stack ends in the Python implementation's ERROR line and no output
    bad = script("bad").replace("p_stack.mrc", "no_such_stack.mrc")
    assert run("reconstruct3d", bad, d, "rec_bad.log") != 0
    assert "ERROR" in open(d / "rec_bad.log").read() and not (d / "n_bad_map1_n1.mrc").exists()


@pytest.mark.parametrize("env", [{"PPM_IO_CHUNK_MB": "1", "PPM_IO_THREADS": "3"}, {"PPM_IO_CHUNK_MB": "1", "PPM_IO_READER": "python", "PPM_NATIVE": "0"},
                                 {"PPM_IO_CHUNK_MB": "4096"}])
def test_executables_give_the_same_files_whatever_the_pipeline_chunking(project, monkeypatch, env):
    """The read / upload / compute pipeline of refine3d and reconstruct3d (native and Python) with staging buffers of 1 MB (16
    images of 64^2 per chunk: 4 chunks, groups, a ragged tail) and of 4 GB (one chunk), odd reader-thread counts, the Python reader:
    the outputs of a range that starts in the middle of the stack do not depend on any of it (parameters to the bit, accumulators to rounding)."""
    d, vol, imgs, truth, start = project
    used = truth.copy()
    used[:, cistem.COL["SCORE"]] = 20.0
    cistem.write_parameters(str(d / "c_r01_used.cistem"), used)

    def rec(tag):
        lines = ["p_stack.mrc", "c_r01_used.cistem", "null", "p_r01.mrc", "c_map1.mrc", "c_map2.mrc", "output.mrc", f"c_{tag}.res", "C1", 7, 55, PX, 300, 0,
                 PX * N / 2, 2 * PX, 0, 2.0, "no", 0, -1, "no", 0, 1, 1, "yes", "no", "no", "no", "no", "yes", "no", "no", "no", "no", "yes",
                 f"{d}/c_{tag}_map1_n1.mrc", f"{d}/c_{tag}_map2_n1.mrc", 1]
        return "\n".join(str(x) for x in lines) + "\n"

    def ref(tag):
        s = refine_script(7, 55, False, out=f"c_{tag}_0000007_0000055.cistem")
        return s
    tag = "base"
    if not (d / "c_base_map1_n1.mrc").exists():
        assert run("reconstruct3d", rec(tag), d, "c_rec.log") == 0 and run("refine3d", ref(tag), d, "c_ref.log") == 0
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    tag = "v%d" % (abs(hash(tuple(sorted(env.items())))) % 100000)
    assert run("reconstruct3d", rec(tag), d, "c_rec.log") == 0 and run("refine3d", ref(tag), d, "c_ref.log") == 0
    for k in (1, 2):          # insertion sums chunk by chunk in single precision: equal to rounding, not to the bit
        x, y = open(d / f"c_{tag}_map{k}_n1.mrc", "rb").read(), open(d / f"c_base_map{k}_n1.mrc", "rb").read()
        assert x[:24] == y[:24] and len(x) == len(y)
        fx, fy = np.frombuffer(x, "<f4", offset=24), np.frombuffer(y, "<f4", offset=24)
        assert np.abs(fx - fy).max() <= 2e-6 * np.abs(fy).max()
    a, b = cistem.read_parameters(str(d / f"c_{tag}_0000007_0000055.cistem")), cistem.read_parameters(str(d / "c_base_0000007_0000055.cistem"))
    assert a.shape == (49, 32) and np.array_equal(a, b)


def test_native_refine3d_equals_the_python_implementation(project, monkeypatch):
    """bin/refine3d is the compiled front end of the default call (pyp_amd/csrc/refine3d_main.cpp) and hands everything else to
    bin/refine3d.py: the same range through both (PPM_NATIVE=0 forces the hand-over) must give the same parameter and changes files
    to the byte; the log carries the classification limit (answer 22) and LOGP differs from a run with another limit."""
    d, vol, imgs, truth, start = project
    assert run("refine3d", refine_script(4, 40, True, out="nat_out.cistem").replace("p_r01_0000004_0000040_changes", "nat_changes"), d, "ref_nat.log") == 0
    monkeypatch.setenv("PPM_NATIVE", "0")
    assert run("refine3d", refine_script(4, 40, True, out="py_out.cistem").replace("p_r01_0000004_0000040_changes", "py_changes"), d, "ref_py.log") == 0
    monkeypatch.delenv("PPM_NATIVE")
    ln, lp = open(d / "ref_nat.log").read(), open(d / "ref_py.log").read()
    assert "libpypmatch, native" in ln and "libpypmatch, native" not in lp and "Refine3D: Normal termination" in ln and "Refine3D: Normal termination" in lp
    assert open(d / "nat_out.cistem", "rb").read() == open(d / "py_out.cistem", "rb").read()
    assert open(d / "nat_changes.cistem", "rb").read() == open(d / "py_changes.cistem", "rb").read()
    tab = lambda s: [x for x in s.splitlines() if len(x) == 70 and x[:7].strip().isdigit()]
    assert tab(ln) == tab(lp) and len(tab(ln)) == 37
    # answer 22: the classification limit moves LOGP / SIGMA and nothing else
    lines = refine_script(4, 40, True, out="nat_cls.cistem").splitlines()
    lines[21] = str(PX * N / 12)
    assert run("refine3d", "\n".join(lines) + "\n", d, "ref_cls.log") == 0
    a, b = cistem.read_parameters(str(d / "nat_out.cistem")), cistem.read_parameters(str(d / "nat_cls.cistem"))
    C = cistem.COL
    same = [c for c in range(32) if c not in (C["LOGP"], C["SIGMA"])]
    assert np.array_equal(a[:, same], b[:, same]) and not np.allclose(a[:, C["LOGP"]], b[:, C["LOGP"]])


def test_refine3d_fraction_refines_a_subset_and_copies_the_rest(project):
    """Answer 14 (frealign.py:3934 sends 1): with 0.5 about half of the range is refined, the other rows come out as they went in."""
    d, vol, imgs, truth, start = project
    lines = refine_script(1, M, True, out="frac_out.cistem").splitlines()
    lines[13] = "0.5"
    assert run("refine3d", "\n".join(lines) + "\n", d, "frac.log") == 0, open(d / "frac.log").read()[-1500:]
    out = cistem.read_parameters(str(d / "frac_out.cistem"))
    from pyp_amd.surface import cli
    use = cli.fraction_mask(start[:, 0], 0.5)
    assert 0 < use.sum() < M and out.shape == start.shape
    assert np.array_equal(out[~use], cistem.read_parameters(str(d / "p_r01.cistem"))[~use])
    assert np.all(out[use, cistem.COL["SCORE"]] != start[use, cistem.COL["SCORE"]])
    assert "of %d rows are refined" % M in open(d / "frac.log").read()


@pytest.mark.parametrize("prog", ["reconstruct3d", "refine3d"])
def test_native_executables_end_with_error_when_an_upload_fails(project, monkeypatch, prog):
    """A failing uploader (PPM_TEST_FAIL_UPLOAD=k: the k-th upload fails like a device error) used to leave the reader waiting for a
    page-locked buffer for ever, with PYP waiting on the process: the run must end in an ERROR line, a non-zero exit and no output."""
    d, vol, imgs, truth, start = project
    used = truth.copy()
    cistem.write_parameters(str(d / "f_used.cistem"), used)
    monkeypatch.setenv("PPM_IO_CHUNK_MB", "1")          # 16 images per chunk: four chunks, three staging buffers
    monkeypatch.setenv("PPM_IO_GROUP", "1")
    monkeypatch.setenv("PPM_TEST_FAIL_UPLOAD", "1")
    if prog == "reconstruct3d":
        lines = ["p_stack.mrc", "f_used.cistem", "null", "p_r01.mrc", "f_map1.mrc", "f_map2.mrc", "output.mrc", "f.res", "C1", 1, M, PX, 300, 0,
                 PX * N / 2, 2 * PX, 0, 2.0, "no", 0, -1, "no", 0, 1, 1, "yes", "no", "no", "no", "no", "yes", "no", "no", "no", "no", "yes",
                 f"{d}/f_map1_n1.mrc", f"{d}/f_map2_n1.mrc", 1]
        script, outs = "\n".join(str(x) for x in lines) + "\n", ["f_map1_n1.mrc", "f_map2_n1.mrc"]
    else:
        script, outs = refine_script(1, M, False, out="f_out.cistem"), ["f_out.cistem"]
    cmd = f"{BIN}/{prog} << eot >> fail_{prog}.log 2>&1\n{script}eot\n"
    r = subprocess.run(cmd, shell=True, cwd=d, timeout=120)              # a hang fails the test through the timeout
    log = open(d / f"fail_{prog}.log").read()
    assert r.returncode != 0 and "ERROR" in log and "PPM_TEST_FAIL_UPLOAD" in log and "native" in log
    assert not any((d / o).exists() for o in outs)


def test_resident_server_keeps_the_stack_and_the_reference_between_calls(project, monkeypatch, tmp_path):
    """PPM_STACK_CACHE=1 (pyp_amd/csrc/dropin_server.h): the executables are clients of bin/ppm_server.  An iteration as PYP runs it -
    reconstruct3d, then refine3d, then reconstruct3d again over the same stack - uploads the range once: the later calls find it
    resident (and refine3d's second call finds its reference prepared).  Files equal those of the one-shot executables; a stack
    that changes on disk is uploaded again; --stop frees the device."""
    d, vol, imgs, truth, start = project
    server = os.path.join(BIN, "ppm_server")
    monkeypatch.setenv("PPM_LOCK_DIR", str(tmp_path))
    used = truth.copy()
    used[:, cistem.COL["SCORE"]] = 20.0
    cistem.write_parameters(str(d / "s_used.cistem"), used)

    def rec(tag):
        lines = ["p_stack.mrc", "s_used.cistem", "null", "p_r01.mrc", "s_map1.mrc", "s_map2.mrc", "output.mrc", f"s_{tag}.res", "C1", 5, 50, PX, 300, 0,
                 PX * N / 2, 2 * PX, 0, 2.0, "no", 0, -1, "no", 0, 1, 1, "yes", "no", "no", "no", "no", "yes", "no", "no", "no", "no", "yes",
                 f"{d}/s_{tag}_map1_n1.mrc", f"{d}/s_{tag}_map2_n1.mrc", 1]
        return "\n".join(str(x) for x in lines) + "\n"
    # one-shot references
    assert run("reconstruct3d", rec("one"), d, "s_one.log") == 0
    assert run("refine3d", refine_script(5, 50, True, out="s_one.cistem"), d, "s_one_ref.log") == 0
    monkeypatch.setenv("PPM_STACK_CACHE", "1")
    monkeypatch.setenv("PPM_STACK_CACHE_IDLE_S", "120")
    try:
        assert run("reconstruct3d", rec("a"), d, "s_a.log") == 0                       # starts the server, uploads, keeps
        assert run("refine3d", refine_script(5, 50, True, out="s_a.cistem"), d, "s_a_ref.log") == 0
        assert run("reconstruct3d", rec("b"), d, "s_b.log") == 0
        assert run("refine3d", refine_script(8, 40, True, out="s_b.cistem"), d, "s_b_ref.log") == 0       # a sub-range of what is resident
        la, lar, lb, lbr = (open(d / f).read() for f in ("s_a.log", "s_a_ref.log", "s_b.log", "s_b_ref.log"))
        assert "resident server" in la and "uploaded and kept resident" in la and "Reconstruct3D: Normal termination" in la
        assert "are resident in device memory" in lar and "Refine3D: Normal termination" in lar and "is prepared already" not in lar
        assert "are resident in device memory" in lb and "are resident in device memory" in lbr and "is prepared already" in lbr
        for k in (1, 2):
            for tag in ("a", "b"):
                x, y = open(d / f"s_{tag}_map{k}_n1.mrc", "rb").read(), open(d / f"s_one_map{k}_n1.mrc", "rb").read()
                assert x[:24] == y[:24] and len(x) == len(y)
                fx, fy = np.frombuffer(x, "<f4", offset=24), np.frombuffer(y, "<f4", offset=24)
                assert np.abs(fx - fy).max() <= 2e-6 * np.abs(fy).max()                # one call over the range instead of several: float32 sums in another order
        assert open(d / "s_a.res").read() == open(d / "s_one.res").read()
        a, b = cistem.read_parameters(str(d / "s_a.cistem")), cistem.read_parameters(str(d / "s_one.cistem"))
        assert np.array_equal(a, b)
        sub = cistem.read_parameters(str(d / "s_b.cistem"))
        assert np.array_equal(sub, b[3:36])
        st = subprocess.run([server, "--stats"], capture_output=True, text=True).stdout
        assert "served 4 calls" in st and "3 resident hits" in st and "1 uploads" in st and "particles 5..50" in st
        # the stack changes on disk (same path, new contents): its identity differs, the range is uploaded again
        stack2 = imgs.copy(); stack2[10] *= -1.0
        os.remove(d / "p_stack.mrc")
        mrc.write(stack2, str(d / "p_stack.mrc"), pixel_size=PX)
        assert run("reconstruct3d", rec("c"), d, "s_c.log") == 0
        assert "uploaded and kept resident" in open(d / "s_c.log").read()
        fc, fo = (np.frombuffer(open(d / f"s_{t}_map1_n1.mrc", "rb").read(), "<f4", offset=24) for t in ("c", "one"))
        fc2, fo2 = (np.frombuffer(open(d / f"s_{t}_map2_n1.mrc", "rb").read(), "<f4", offset=24) for t in ("c", "one"))
        assert not (np.allclose(fc, fo) and np.allclose(fc2, fo2))
        # a request carries its client's umask (and PPM_* settings): files are created as the one-shot run of that client would create them
        old = os.umask(0o027)
        try:
            assert run("reconstruct3d", rec("m"), d, "s_m.log") == 0
        finally:
            os.umask(old)
        assert (os.stat(d / "s_m_map1_n1.mrc").st_mode & 0o777) == 0o640 and (os.stat(d / "s_a_map1_n1.mrc").st_mode & 0o777) == (0o666 & ~old)
        # errors come back through the server with the contract of the executables
        bad = rec("bad").replace("s_used.cistem", "s_missing.cistem")
        assert run("reconstruct3d", bad, d, "s_bad.log") != 0 and "ERROR" in open(d / "s_bad.log").read() and not (d / "s_bad_map1_n1.mrc").exists()
    finally:
        r = subprocess.run([server, "--stop"], capture_output=True, text=True, timeout=60)
        mrc.write(imgs, str(d / "p_stack.mrc"), pixel_size=PX)                          # the module's fixture as it was
    assert "stopping" in r.stdout
