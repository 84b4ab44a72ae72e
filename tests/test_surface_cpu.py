"""CPU tests of the call surface: positional answer scripts exactly as PYP writes them
(src/pyp/refine/frealign/frealign.py:3918-3994, :1780-1824, :1878-1888, :2075-2093;
src/pyp/system/wrapper_functions.py:512-561), dump files, log table format, error behaviour."""
import io
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from pyp_amd.surface import cli, prompts

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "golden_r03.json")))


def golden_script(group, case):
    """The answer lines of a here-doc the reference's own string builder produced (tests/golden/gen_golden_r03.py)."""
    cmd = next(c for c in GOLD[group] if c["case"] == case)["script"]
    return cmd.split("\n", 1)[1]


# frealign.mrefine_version with PYP's defaults at box 64 / 4.32 A (frealign.py:3918-3994), generated, not typed
REFINE_CISTEM = golden_script("refine3d_scripts", "dropin_box64")

# verbatim from src/pyp/system/wrapper_functions.py:512-561
REFINE_PAR = """../spr_frames_00_04_stack.mrc
spr_frames_00_04_r01_08.par
spr_frames_00_04_r01_07.mrc
statistics_r01.txt
yes
spr_frames_00_04_r01_08_match.mrc_0000001_0000027
spr_frames_00_04_r01_08.par_0000001_0000027
/dev/null
O
1
27
1.08
300.0
2.7
0.07
400.0
65
100.0
3.0
30.0
8
97.5
3.0
200
20
0
0
0
0
0
0
500.0
50.0
1
no
yes
yes
yes
yes
yes
yes
no
no
no
no
"""

def test_refine3d_cistem_script():
    d = prompts.parse_refine3d(prompts.read_answers(io.StringIO(REFINE_CISTEM)))
    assert d["surface"] == "cistem" and d["symmetry"] == "D7" and (d["first"], d["last"]) == (1, 143)
    assert d["pixel_size"] == 4.32 and d["outer_radius"] == 85 and d["res_high"] == 8 and d["res_search"] == 8
    assert d["angular_step"] == 20.0 and d["top_hits"] == 20 and d["global_search"] and not d["local_refine"]
    assert d["refine_psi"] and d["refine_y"] and not d["invert"] and d["search_mask_radius"] == 127.5
    assert d["output_params"] == "t20s_r01_01_0000001_0000143.cistem"


def test_focus_mask_answers_reach_the_library_configuration():
    """class_focusmask "X,Y,Z,R" (frealign.py:3958) with answer 44 "apply 2D masking" = yes (:3846-3849): corner-based Angstrom
    become centre-based ones in ppm_refine_cfg.focus; masking = no leaves the mask off whatever the four numbers say."""
    from pyp_amd.surface import cli
    lines = REFINE_CISTEM.splitlines()
    lines[28:32] = ["300.0", "250.5", "276.48", "60"]            # answers 29-32
    d = prompts.parse_refine3d(prompts.read_answers(io.StringIO("\n".join(lines) + "\n")))
    assert (d["focus_x"], d["focus_y"], d["focus_z"], d["focus_r"]) == (300.0, 250.5, 276.48, 60.0) and not d["mask_2d"]
    assert list(cli.refine_cfg_from_answers(d, 128).focus) == [0.0, 0.0, 0.0, 0.0]
    lines[43] = "yes"                                              # answer 44
    d = prompts.parse_refine3d(prompts.read_answers(io.StringIO("\n".join(lines) + "\n")))
    assert d["mask_2d"]
    f = list(cli.refine_cfg_from_answers(d, 128).focus)           # box 128 x 4.32 A: centre at 276.48 A
    assert np.allclose(f, [300.0 - 276.48, 250.5 - 276.48, 0.0, 60.0], atol=1e-4)


def test_refine3d_par_script_verbatim_from_reference():
    d = prompts.parse_refine3d(prompts.read_answers(io.StringIO(REFINE_PAR)))
    assert d["surface"] == "par" and d["symmetry"] == "O" and (d["first"], d["last"]) == (1, 27)
    assert (d["pixel_size"], d["voltage"], d["cs"], d["amplitude_contrast"]) == (1.08, 300.0, 2.7, 0.07)
    assert d["outer_radius"] == 65 and d["res_high"] == 3.0 and d["res_signed_cc"] == 30.0 and d["angular_step"] == 200
    assert not d["global_search"] and d["local_refine"] and d["use_statistics"] and d["output_changes"] == "/dev/null"


def test_reconstruct3d_script_with_and_without_dose_weighting():
    """split_reconstruction's own strings (frealign.py:1780-1824); every variant is checked in tests/test_golden_r03.py."""
    d = prompts.parse_reconstruct3d(prompts.read_answers(io.StringIO(golden_script("reconstruct3d_scripts", "defaults"))))
    assert not d["dose_weighting"] and d["res_limit"] == 4.32 and d["score_bfactor"] == 2.0 and d["threads"] == 1
    assert d["dump_1"].endswith("_map1_n1.mrc") and d["split_even_odd"] and d["per_particle_splitting"] and d["adjust_scores"]
    d = prompts.parse_reconstruct3d(prompts.read_answers(io.StringIO(golden_script("reconstruct3d_scripts", "dose_weighting_no_file"))))
    assert d["dose_weighting"] and d["dose_fraction"] == 0.5 and d["dump_2"].endswith("_map2_n3.mrc") and d["threads"] == 1


def test_short_or_bad_scripts_raise():
    with pytest.raises(prompts.PromptError, match="ERROR"):
        prompts.parse_refine3d(["a.mrc", "b.cistem"])
    bad = REFINE_CISTEM.replace("\nyes\nno\nyes\nyes", "\nmaybe\nno\nyes\nyes", 1)
    with pytest.raises(prompts.PromptError, match="yes or no"):
        prompts.parse_refine3d(prompts.read_answers(io.StringIO(bad)))
    with pytest.raises(prompts.PromptError):
        prompts.parse_merge3d(["a", "b", "c", "d", "700", "0", "x", "s1", "s2", "3"])


def test_dump_roundtrip_and_names(tmp_path):
    assert prompts.dump_name("/s/ds_r01_map1_n.mrc", 12) == "/s/ds_r01_map1_n12.mrc"
    data = np.arange(32 * 32 * 17 * 3, dtype=np.float32)
    p = str(tmp_path / "x_map1_n1.mrc")
    cli.write_dump(p, 32, 1.5, 77, data)
    box, px, cnt, back = cli.read_dump(p)
    assert (box, px, cnt) == (32, 1.5, 77) and np.array_equal(back, data)


def test_local_merge3d_sums_dumps(tmp_path, capsys):
    for k in (1, 2, 3):
        cli.write_dump(str(tmp_path / f"t_map1_n{k}.mrc"), 32, 2.0, 10 * k, np.full(32 * 32 * 17 * 3, float(k), np.float32))
        cli.write_dump(str(tmp_path / f"t_map2_n{k}.mrc"), 32, 2.0, k, np.full(32 * 32 * 17 * 3, 0.5 * k, np.float32))
    script = "\n".join([str(tmp_path / "o_map1_n1.mrc"), str(tmp_path / "o_map2_n1.mrc"), str(tmp_path / "t_map1_n.mrc"),
                        str(tmp_path / "t_map2_n.mrc"), "3"]) + "\n"
    assert cli.local_merge3d_main(stdin=io.StringIO(script)) == 0
    assert "LocalMerge3D: Normal termination" in capsys.readouterr().out
    b, px, cnt, d = cli.read_dump(str(tmp_path / "o_map1_n1.mrc"))
    assert cnt == 60 and np.all(d == 6.0)
    b, px, cnt, d = cli.read_dump(str(tmp_path / "o_map2_n1.mrc"))
    assert cnt == 6 and np.all(d == 3.0)


def test_merge_log_table_parses_like_the_caller(tmp_path):
    """Replays the slicing of src/pyp/refine/frealign/frealign.py:2557-2567 on our log text."""
    from io import StringIO
    stats = np.array([[b, 64.0 / b, b / 64.0, 1.0 - 0.01 * b, 0.99, 12.5, 300.0 / b] for b in range(1, 32)])
    log = "blah\n   NO.   RESOL  RING RAD       FSC  Part_FSC Part_SSNR  Rec_SSNR\n" + "\n".join(cli.format_stats_table(stats)) + "\n\n\nMerge3D: Normal termination\n"
    A = log
    Afsc = A[A.find("Rec_SSNR") + 9: A.find("Merge3D: Normal termination") - 3]
    widths = [5, 8, 10, 10, 10, 10, 10]
    rows = len(Afsc.split("\n"))
    cur = np.genfromtxt(StringIO(Afsc), delimiter=widths).reshape((rows, len(widths)))[:, list(range(1, 4, 2))].astype("float")
    assert rows == 31
    assert np.allclose(cur[:, 0], np.round(stats[:, 1], 2)) and np.allclose(cur[:, 1], np.round(stats[:, 3], 4))


def test_executables_fail_loudly_without_inputs(tmp_path):
    """Non-zero exit, a line containing ERROR, and no output file (SURVEY.md §8b 'Errors')."""
    script = REFINE_CISTEM.replace("t20s_r01_01_0000001_0000143.cistem", str(tmp_path / "out.cistem"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bin", "refine3d.py")], input=script, capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode != 0 and "ERROR" in r.stdout and not (tmp_path / "out.cistem").exists()
    exe = os.path.join(ROOT, "bin", "refine3d")              # the compiled front end hands the call (missing inputs) to the same implementation
    if os.path.exists(exe):
        r = subprocess.run([exe], input=script, capture_output=True, text=True, cwd=tmp_path)
        assert r.returncode != 0 and "ERROR" in r.stdout and not (tmp_path / "out.cistem").exists()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bin", "merge3d")], input="a\nb\n", capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode != 0 and "ERROR" in r.stdout


def test_beam_tilt_rows_are_accepted(tmp_path):
    """BEAM_TILT_X / Y columns (cistem_star_file.py:596-628) are honoured by the library (phase term removed in k_prep): rows that
    carry them pass the range selection like any other."""
    from pyp_amd.formats import cistem
    rows = cistem.default_rows(4, 2.0, 300.0, 2.7, 0.07)
    rows[2, cistem.COL["BEAM_TILT_X"]] = 0.3
    assert len(cli._select(rows, 1, 4)) == 4


def test_gpu_lock_serialises_processes(tmp_path, monkeypatch):
    """Two holders of the per-GPU lock never overlap (concurrent refine3d processes of one node)."""
    import multiprocessing as mp
    import time
    monkeypatch.setenv("PPM_LOCK_DIR", str(tmp_path))

    def worker(tag, out):
        with cli.gpu_lock(0):
            t0 = time.time(); time.sleep(0.3); out.put((tag, t0, time.time()))
    q = mp.Queue()
    ps = [mp.Process(target=worker, args=(i, q)) for i in range(2)]
    [p.start() for p in ps]; [p.join() for p in ps]
    a, b = sorted([q.get(), q.get()], key=lambda r: r[1])
    assert b[1] >= a[2] - 1e-3


def _gpu_present():
    try:
        import torch
        return torch.cuda.device_count() > 0
    except Exception:          # noqa: BLE001
        return False


@pytest.mark.skipif(_gpu_present(), reason="the no-device message needs a host without a GPU")
def test_native_reconstruct3d_hands_over_and_fails_loudly_without_a_device(tmp_path):
    """bin/reconstruct3d (compiled, pyp_amd/csrc/reconstruct3d_main.cpp): anything outside its fast path reaches bin/reconstruct3d.py
    with the same stdin (here: an answer that is not yes / no, and too few answers - the Python parser's own messages); its fast
    path on a host without a GPU ends in the library's ERROR line, a non-zero exit and no output file."""
    import subprocess
    from pyp_amd.formats import cistem, mrc
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bin", "reconstruct3d")
    if not os.path.exists(exe):
        pytest.skip("bin/reconstruct3d is built by __graft_entry__.build()")
    rows = cistem.default_rows(6, 2.0, 300.0, 2.7, 0.07)
    cistem.write_parameters(str(tmp_path / "p.cistem"), rows)
    mrc.write(np.zeros((6, 32, 32), np.float32), str(tmp_path / "s.mrc"), pixel_size=2.0)
    lines = ["s.mrc", "p.cistem", "null", "ref.mrc", "m1.mrc", "m2.mrc", "out.mrc", "r.res", "C1", 1, 6, 2.0, 300, 0, 30.0, 4.0, 0, 2.0, "no", 0, -1, "no", 0, 1, 1,
             "yes", "no", "no", "no", "no", "yes", "no", "no", "no", "no", "yes", "d1.mrc", "d2.mrc", 1]

    def run(ls):
        return subprocess.run([exe], input="\n".join(str(x) for x in ls) + "\n", cwd=tmp_path, capture_output=True, text=True)
    r = run(lines)
    assert r.returncode != 0 and "ERROR" in r.stdout and "native" in r.stdout and not (tmp_path / "d1.mrc").exists()
    bad = list(lines); bad[25] = "maybe"
    r = run(bad)
    assert r.returncode != 0 and "must be yes or no" in r.stdout and "native" not in r.stdout
    r = run(lines[:10])
    assert r.returncode != 0 and "expected 22 answers" in r.stdout
