"""Every answer of the refine3d / reconstruct3d scripts is honoured or refused loudly (SURVEY.md 7 "must fail loudly, not
silently"; frealign.py:3934-3945 refine3d answers 14 / 17 / 22, :1763-1770 and :1796-1808 reconstruct3d answers 14, 17, 20 / 21, 24).
One test per answer: a non-default value either changes the result, or ends in a line containing ERROR, a non-zero exit and no
output file - through the Python implementation and through the compiled front ends (which hand such calls over)."""
import io
import os
import subprocess

import numpy as np
import pytest

from pyp_amd.formats import cistem, mrc
from pyp_amd.surface import cli, prompts

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, PX, M = 32, 2.0, 8


def refine_lines(out="out.cistem"):
    """The 50 answers for a tiny project in the working directory."""
    return ["s.mrc", "p.cistem", "null", "ref.mrc", "statistics_r01.txt", "no", "no", "match.mrc", out, "changes.cistem", "C1", 1, M, 1, PX, 300, 0,
            0.4 * N * PX, 0, PX * N / 12, 30.0, 8.0, 0.4 * N * PX, PX * N / 8, 15.0, 20, 6.0, 6.0, 0, 0, 0, 0, 500, 50.0, 1, "yes", "no",
            "yes", "yes", "yes", "yes", "yes", "no", "no", "no", "yes", "no", "no", "no", "no"]


def recon_lines():
    """The 39 answers."""
    return ["s.mrc", "p.cistem", "null", "ref.mrc", "m1.mrc", "m2.mrc", "out.mrc", "r.res", "C1", 1, M, PX, 300, 0, 30.0, 4.0, 0, 2.0, "no", 0, -1, "no", 0, 1, 1,
            "yes", "no", "no", "no", "no", "yes", "no", "no", "no", "no", "yes", "d1.mrc", "d2.mrc", 1]


@pytest.fixture()
def project(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("PPM_LOCK_DIR", str(tmp_path))
    rows = cistem.default_rows(M, PX, 300.0, 2.7, 0.07)
    rows[:, cistem.COL["TIND"]] = np.arange(M) % 4
    cistem.write_parameters("p.cistem", rows)
    mrc.write(np.zeros((M, N, N), np.float32), "s.mrc", pixel_size=PX)
    mrc.write(np.zeros((N, N, N), np.float32), "ref.mrc", pixel_size=PX)
    return tmp_path, rows


def _run_py(main, lines, capsys):
    with pytest.raises(SystemExit) as e:
        main(stdin=io.StringIO("\n".join(str(x) for x in lines) + "\n"))
    return e.value.code, capsys.readouterr().out


# ------------------------------------------------------------------------------------------------ refine3d
def test_refine3d_answer_17_inner_radius_is_refused(project, capsys):
    ls = refine_lines(); ls[16] = 12.0
    code, out = _run_py(cli.refine3d_main, ls, capsys)
    assert code != 0 and "ERROR" in out and "inner_radius" in out and not os.path.exists("out.cistem")


def test_refine3d_answer_14_fraction_selects_a_fixed_subset(project, capsys):
    """fraction < 1: a deterministic subset is refined (the same rows whatever the particle range), out-of-range values are refused."""
    pos = np.arange(1, 20001)
    m = cli.fraction_mask(pos, 0.25)
    assert abs(m.mean() - 0.25) < 0.02
    assert np.array_equal(m[100:200], cli.fraction_mask(pos[100:200], 0.25))          # independent of the range it is asked about
    assert cli.fraction_mask(pos, 1.0).all() and (cli.fraction_mask(pos, 0.5) >= m).all()      # nested: a larger fraction keeps the smaller one's rows
    for bad in (0.0, 1.5, -1):
        ls = refine_lines(); ls[13] = bad
        code, out = _run_py(cli.refine3d_main, ls, capsys)
        assert code != 0 and "ERROR" in out and "fraction" in out and not os.path.exists("out.cistem")


def test_refine3d_answer_22_reaches_the_library_settings(project):
    """class_rhcls (frealign.py:3945) -> ppm_refine_cfg.res_classification; a negative limit is refused."""
    d = prompts.parse_refine3d([str(x) for x in refine_lines()])
    assert d["res_classification"] == 8.0
    cfg = cli.refine_cfg_from_answers(d, N)
    assert abs(cfg.res_classification - 8.0) < 1e-6 and abs(cfg.res_high - PX * N / 12) < 1e-5


def test_refine3d_answer_22_negative_is_refused(project, capsys):
    ls = refine_lines(); ls[21] = -4
    code, out = _run_py(cli.refine3d_main, ls, capsys)
    assert code != 0 and "ERROR" in out and "classification" in out


def test_oracle_logp_follows_the_classification_limit():
    """LOGP / SIGMA are evaluated over res_low .. res_classification, SCORE and the pose over the refinement band: with the limit
    at res_high (or 0) nothing changes; a coarser limit changes LOGP and SIGMA only, and LOGP equals the stated formula."""
    from oracle import oracle
    from pyp_amd import synth
    from pyp_amd.abi import RefineCfg
    n, px = 48, 2.0
    vol, stack, rows = synth.make_dataset(n, 3, pixel=px, snr=0.3)
    imgs = stack.numpy()
    base = dict(box=n, pixel_size=px, mask_radius=0.4 * n * px, res_high=px * n / 18, global_search=0, local_refine=1, res_signed_cc=30.0)
    ref = oracle.Reference(vol, n / 2)
    out0, _ = oracle.refine_batch(ref, RefineCfg.make(**base), imgs, rows)
    outh, _ = oracle.refine_batch(ref, RefineCfg.make(res_classification=px * n / 18, **base), imgs, rows)
    outc, _ = oracle.refine_batch(ref, RefineCfg.make(res_classification=px * n / 9, **base), imgs, rows)
    C = cistem.COL
    assert np.array_equal(out0, outh)
    pose = [C[k] for k in ("PSI", "THETA", "PHI", "X_SHIFT", "Y_SHIFT", "SCORE")]
    assert np.array_equal(out0[:, pose], outc[:, pose])
    assert not np.allclose(out0[:, C["LOGP"]], outc[:, C["LOGP"]]) and not np.allclose(out0[:, C["SIGMA"]], outc[:, C["SIGMA"]])
    res = outc[:, C["SIGMA"]] ** 2
    assert np.allclose(outc[:, C["LOGP"]], -0.5 * np.pi * 9.0 ** 2 * (np.log(2 * np.pi * res) + 1.0), rtol=1e-5)       # the limit travels as a float32
    # the classification-band correlation is the oracle's own score at that band
    cfg9 = RefineCfg.make(**dict(base, res_high=px * n / 9))
    sc = oracle.score_batch(oracle.Reference(vol, n / 2), cfg9, imgs, outc)
    assert np.allclose(np.sqrt(1 - sc ** 2), outc[:, C["SIGMA"]], atol=2e-3)       # whitening weights differ slightly between the two bands


# ------------------------------------------------------------------------------------------------ reconstruct3d
@pytest.mark.parametrize("index,value,key", [(13, 5.0, "inner_radius"), (16, 6.0, "res_reference"), (23, 2.0, "smoothing")])
def test_reconstruct3d_unsupported_answers_are_refused(project, capsys, index, value, key):
    ls = recon_lines(); ls[index] = value
    code, out = _run_py(cli.reconstruct3d_main, ls, capsys)
    assert code != 0 and "ERROR" in out and key in out and not os.path.exists("d1.mrc") and not os.path.exists("d2.mrc")


def test_reconstruct3d_answers_20_21_are_the_tilt_window(project):
    """min / max tilt-particle score = csp_UseImagesForRefinementMin / Max (frealign.py:1763-1766): rows outside the TIND window
    are switched off; PYP's 0 / -1 keeps everything."""
    _, rows = project
    r = rows.copy()
    assert cli.apply_tilt_window(r, 0, -1) == 0 and np.array_equal(r, rows)
    r = rows.copy()
    assert cli.apply_tilt_window(r, 1, 2) == int(((rows[:, cistem.COL["TIND"]] < 1) | (rows[:, cistem.COL["TIND"]] > 2)).sum())
    t = r[:, cistem.COL["TIND"]]
    assert np.all(r[(t >= 1) & (t <= 2), cistem.COL["OCCUPANCY"]] == 100.0) and np.all(r[(t < 1) | (t > 2), cistem.COL["OCCUPANCY"]] == 0.0)
    r = rows.copy()
    assert cli.apply_tilt_window(r, 2, -1) == int((rows[:, cistem.COL["TIND"]] < 2).sum())


# ------------------------------------------------------------------------------------------------ compiled front ends
def _native(prog):
    exe = os.path.join(ROOT, "bin", prog)
    if not os.path.exists(exe) or open(exe, "rb").read(4) != b"\x7fELF":
        pytest.skip(f"bin/{prog} is built by __graft_entry__.build()")
    return exe


def _run_exe(exe, lines, cwd):
    return subprocess.run([exe], input="\n".join(str(x) for x in lines) + "\n", cwd=cwd, capture_output=True, text=True, timeout=120)


@pytest.mark.parametrize("index,value,key", [(13, 5.0, "inner_radius"), (16, 6.0, "res_reference"), (23, 2.0, "smoothing")])
def test_native_reconstruct3d_hands_refused_answers_to_python(project, index, value, key):
    """None of the answers is parsed into a dummy: a non-default value takes the call out of the compiled fast path, and the Python
    implementation's refusal (one definition of the message) is what the caller sees."""
    d, _ = project
    ls = recon_lines(); ls[index] = value
    r = _run_exe(_native("reconstruct3d"), ls, d)
    assert r.returncode != 0 and "ERROR" in r.stdout and key in r.stdout and "native" not in r.stdout and not (d / "d1.mrc").exists()


def test_native_reconstruct3d_hands_a_tilt_window_to_python(project):
    d, _ = project
    ls = recon_lines(); ls[19], ls[20] = 1, 2
    r = _run_exe(_native("reconstruct3d"), ls, d)
    assert "native" not in r.stdout and "min_tilt_score" in r.stdout          # the Python implementation's answer table


def _gpu_present():
    try:
        import torch
        return torch.cuda.device_count() > 0
    except Exception:          # noqa: BLE001
        return False


@pytest.mark.skipif(_gpu_present(), reason="the no-device message needs a host without a GPU")
def test_native_refine3d_fast_path_and_hand_over(project):
    """bin/refine3d (compiled, pyp_amd/csrc/refine3d_main.cpp): the default call stays in the compiled program (without a device it
    ends in the library's ERROR line, non-zero, no output); a non-default answer, the .par surface or a malformed answer reaches
    bin/refine3d.py with the same stdin as a child process whose exit status is passed on."""
    d, _ = project
    exe = _native("refine3d")
    r = _run_exe(exe, refine_lines(), d)
    assert r.returncode != 0 and "ERROR" in r.stdout and "native" in r.stdout and not (d / "out.cistem").exists()
    for index, value, needle in ((16, 12.0, "inner_radius"), (13, 0.0, "fraction"), (35, "maybe", "must be yes or no")):
        ls = refine_lines(); ls[index] = value
        r = _run_exe(exe, ls, d)
        assert r.returncode != 0 and "ERROR" in r.stdout and needle in r.stdout and "native" not in r.stdout and not (d / "out.cistem").exists()
    r = _run_exe(exe, refine_lines()[:10], d)
    assert r.returncode != 0 and "expected 50 answers" in r.stdout
