"""Host-side rows H1 / H5 / H6 / H9 / H13 / f-2 pinned to the reference's OWN output (tests/golden/gen_golden_r03.py ran
frealign.mrefine_version, split_reconstruction, local_merge_reconstruction, merge_reconstructions, scores.shape_phase_residuals,
metadata.core.compute_global_weights, local_run.create_csp_split_commands, particle_cspt.split_parameter_file and
cistem_star_file.Parameters.merge in the build container): the answer scripts are consumed verbatim by the parsers, the
selection and dose tables are reproduced exactly."""
import json
import os

import numpy as np
import pytest

from pyp_amd import dose, select
from pyp_amd.formats import cistem
from pyp_amd.surface import cli, csp_cli, prompts

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GOLD = json.load(open(os.path.join(HERE, "golden_r03.json")))
SEL = np.load(os.path.join(HERE, "select_r03.npz"))
C = cistem.COL


def _case(group, name):
    return next(c for c in GOLD[group] if c["case"] == name)


# ---- H1: refine3d ---------------------------------------------------------------------------------------------------

def test_every_refine3d_script_of_mrefine_version_parses():
    """src/pyp/refine/frealign/frealign.py:3918-3994, six parameter sets; the here-doc is taken as PYP builds it."""
    for c in GOLD["refine3d_scripts"]:
        prog, log, answers = prompts.split_heredoc(c["script"])
        assert prog == "/opt/pyp/external/cistem2/refine3d" and log == "../log/%s_msearch_n.log_%s" % (c["name"], c["ranger"])
        assert len(answers) == 50, c["case"]
        d = prompts.parse_refine3d(answers)
        assert d["surface"] == "cistem" and (d["first"], d["last"]) == (c["first"], c["last"]) and d["fraction"] == 1.0
        assert d["input_params"] == c["name"] + ".cistem" and d["reference"] == c["name"] + ".mrc"
        assert d["output_params"] == "%s_%s.cistem" % (c["name"], c["ranger"])
        assert d["output_changes"] == "%s_%s_changes.cistem" % (c["name"], c["ranger"])
        assert d["match_out"] == "%s_match.mrc_%s" % (c["name"], c["ranger"])
        assert d["pixel_size"] == pytest.approx(c["overrides"].get("scope_pixel", 1.08) * 2) and d["top_hits"] == 20 and d["defocus_step"] == 50.0
        assert not (d["exclude_edges"] or d["normalize_reference"] or d["threshold_reference"])


def test_refine3d_defaults_are_pyps_defaults():
    d = prompts.parse_refine3d(prompts.split_heredoc(_case("refine3d_scripts", "defaults_global_D7")["script"])[2])
    assert d["stack"] == "../t20s_stack.mrc" and d["global_stats"] == "null" and d["statistics"] == "statistics_r01.txt"
    assert not d["use_statistics"] and not d["use_priors"] and d["symmetry"] == "D7" and d["molecular_mass"] == 700.0
    assert (d["inner_radius"], d["outer_radius"]) == (0.0, 85.0) and (d["res_low"], d["res_high"], d["res_search"]) == (100.0, 10.0, 10.0)
    assert d["res_signed_cc"] == 30.0 and d["res_classification"] == 8.0 and d["search_mask_radius"] == 127.5      # 1.5 x radius
    assert d["angular_step"] == 20.0 and (d["search_range_x"], d["search_range_y"]) == (0.0, 0.0)                   # 0 = mask radius
    assert (d["focus_x"], d["focus_y"], d["focus_z"], d["focus_r"]) == (0, 0, 0, 0) and d["defocus_range"] == 500.0 and d["padding"] == 1.0
    assert d["global_search"] and not d["local_refine"]
    assert all(d[k] for k in ("refine_psi", "refine_theta", "refine_phi", "refine_x", "refine_y"))
    assert not (d["calc_match"] or d["mask_2d"] or d["refine_defocus"] or d["normalize"] or d["invert"])
    cfg = cli.refine_cfg_from_answers(d, 128)
    assert cfg.global_search == 1 and cfg.box == 128 and abs(cfg.pixel_size - 2.16) < 1e-6 and abs(cfg.res_search - 10.0) < 1e-6


def test_refine3d_variants():
    d = prompts.parse_refine3d(prompts.split_heredoc(_case("refine3d_scripts", "local_C1")["script"])[2])
    assert d["local_refine"] and not d["global_search"] and d["symmetry"] == "C1" and (d["first"], d["last"]) == (144, 286)
    d = prompts.parse_refine3d(prompts.split_heredoc(_case("refine3d_scripts", "focus_mask_fboost")["script"])[2])
    assert (d["focus_x"], d["focus_y"], d["focus_z"], d["focus_r"]) == (300.0, 250.5, 276.48, 60.0) and d["mask_2d"] and d["res_signed_cc"] == 12.0
    f = list(cli.refine_cfg_from_answers(d, 256).focus)            # box 256 x 2.16 A: centre at 276.48 A
    assert np.allclose(f, [300.0 - 276.48, 250.5 - 276.48, 0.0, 60.0], atol=1e-4)
    d = prompts.parse_refine3d(prompts.split_heredoc(_case("refine3d_scripts", "schedules_it4")["script"])[2])
    assert (d["res_high"], d["res_search"], d["angular_step"], d["padding"], d["search_mask_radius"]) == (6.0, 6.0, 7.5, 2.0, 120.0)
    assert (d["search_range_x"], d["search_range_y"]) == (12.0, 12.0)
    # refine_mask "1,0,1,0,0" at iteration 4: psi <- flag 0, theta AND phi <- flag 1 (the reference's quirk, frealign.py:3805-3817)
    assert (d["refine_psi"], d["refine_theta"], d["refine_phi"], d["refine_x"], d["refine_y"]) == (True, False, False, False, False)
    d = prompts.parse_refine3d(prompts.split_heredoc(_case("refine3d_scripts", "fssnr_stat_priors_defocus_match_invert")["script"])[2])
    assert d["use_statistics"] and d["use_priors"] and d["global_stats"] == "t20s_r01_02_stat.cistem" and d["refine_defocus"]
    assert d["defocus_range"] == 750.0 and d["calc_match"] and d["invert"]
    d = prompts.parse_refine3d(prompts.split_heredoc(_case("refine3d_scripts", "fssnr_without_statistics_file")["script"])[2])
    assert not d["use_statistics"] and d["global_stats"] == "t20s_r01_02_stat.cistem"      # class_num 2 + file present
    d = prompts.parse_refine3d(prompts.split_heredoc(_case("refine3d_scripts", "stack_on_scratch")["script"])[2])
    assert d["stack"] == GOLD["scratch_token"] + "/t20s_stack.mrc"


# ---- H5: reconstruct3d ----------------------------------------------------------------------------------------------

def test_every_reconstruct3d_script_of_split_reconstruction_parses():
    """frealign.py:1780-1824 with run=False; 39 answers, 43 with the dose-weighting block."""
    for c in GOLD["reconstruct3d_scripts"]:
        prog, log, answers = prompts.split_heredoc(c["script"])
        assert prog == "/opt/pyp/external/cistem2/reconstruct3d" and log == "t20s_r01_%07d_%07d_mreconst.log" % (c["first"], c["last"])
        d = prompts.parse_reconstruct3d(answers)
        assert len(answers) == (43 if d["dose_weighting"] else 39), c["case"]
        assert (d["first"], d["last"]) == (c["first"], c["last"]) and d["input_params"] == "../t20s_r01_used.cistem"
        assert d["reference"] == "../t20s_r01.mrc" and d["res_file"] == "t20s_r01_n%d.res" % c["first"] and d["threads"] == 1
        assert d["dump"] and d["dump_1"] == "%s/t20s_r01_map1_n%d.mrc" % (GOLD["scratch_token"], c["count"])
        assert d["dump_2"] == "%s/t20s_r01_map2_n%d.mrc" % (GOLD["scratch_token"], c["count"])
        assert d["split_even_odd"] and not d["center_mass"] and not d["exclude_edges"] and not d["threshold_reference"]
        assert (d["smoothing"], d["padding"], d["score_threshold"], d["min_tilt_score"], d["max_tilt_score"]) == (1, 1, 0, 0, -1)


def test_reconstruct3d_variants():
    d = prompts.parse_reconstruct3d(prompts.split_heredoc(_case("reconstruct3d_scripts", "defaults")["script"])[2])
    # PYP's default applies the point group in the reconstruction (reconstruct_apply_symmetry defaults to true)
    assert d["symmetry"] == "D7" and d["res_limit"] == pytest.approx(4.32) and d["outer_radius"] == pytest.approx(138.24)
    assert d["score_bfactor"] == 2.0 and not d["score_weighting"] and not d["dose_weighting"] and d["stack"] == "../t20s_stack.mrc"
    assert not d["normalize"] and d["adjust_scores"] and not d["invert"] and not d["crop"] and d["per_particle_splitting"]
    assert not d["likelihood_blurring"] and d["global_stats"] == "null"
    d = prompts.parse_reconstruct3d(prompts.split_heredoc(_case("reconstruct3d_scripts", "apply_symmetry_rrec_radrec")["script"])[2])
    assert (d["res_limit"], d["outer_radius"], d["score_bfactor"]) == (4.5, 120.0, 4.0) and d["score_weighting"] and d["likelihood_blurring"]
    assert d["adjust_scores"] and d["crop"] and d["invert"]
    d = prompts.parse_reconstruct3d(prompts.split_heredoc(_case("reconstruct3d_scripts", "dose_weighting_no_file")["script"])[2])
    assert d["dose_weighting"] and d["dose_weights_file"] == "/scratch/not_provided" and d["dose_multiply"]
    assert (d["dose_fraction"], d["dose_transition"]) == (0.5, 4.0)
    d = prompts.parse_reconstruct3d(prompts.split_heredoc(_case("reconstruct3d_scripts", "dose_weighting_external_file")["script"])[2])
    assert d["dose_weights_file"].endswith("global_weight.txt") and not d["dose_multiply"] and (d["dose_fraction"], d["dose_transition"]) == (1.0, 2.5)
    d = prompts.parse_reconstruct3d(prompts.split_heredoc(_case("reconstruct3d_scripts", "stack_on_scratch")["script"])[2])
    assert d["stack"] == GOLD["scratch_token"] + "/t20s_stack.mrc"


# ---- H6: local_merge3d / merge3d ----------------------------------------------------------------------------------------

def test_local_merge3d_script_and_dump_names():
    """frealign.py:1870-1888: the dumps are renamed to temp_map{1,2}_n1..nK before the call; our dump_name() must find them."""
    g = GOLD["local_merge3d"]
    prog, log, answers = prompts.split_heredoc(g["script"])
    assert prog.endswith("/cistem2/local_merge3d") and log == "local_merge3d.log"
    d = prompts.parse_local_merge3d(answers)
    assert (d["out_dump_1"], d["out_dump_2"], d["n_dumps"]) == ("dumpfile_map1.mrc", "dumpfile_map2.mrc", 4) and g["returned"] == 4
    want = sorted(prompts.dump_name(d["dump_seed_1"], k) for k in range(1, 5)) + sorted(prompts.dump_name(d["dump_seed_2"], k) for k in range(1, 5))
    assert want == g["files_seen_by_program"]
    assert g["files_after"] == ["t20s_r01_map1_n1.mrc", "t20s_r01_map2_n1.mrc"]      # outputs take the first dump's name (:1898-1899)


def test_merge3d_scripts():
    for key, rad, it, n in (("merge3d", 138.24, 2, 3), ("merge3d_radrec", 150.0, 3, 2)):
        prog, log, answers = prompts.split_heredoc(GOLD[key]["script"])
        assert prog.endswith("/cistem2/merge3d") and log == "../log/t20s_r01_%02d_mreconst.log" % it
        d = prompts.parse_merge3d(answers)
        name = "t20s_r01_%02d" % it
        assert (d["half1"], d["half2"], d["filtered"], d["statistics"]) == (name + "_half1.mrc", name + "_half2.mrc", name + ".mrc", name + "_statistics.txt")
        assert d["molecular_mass"] == 700.0 and d["inner_radius"] == 0 and d["outer_radius"] == pytest.approx(rad) and d["n_dumps"] == n
        assert d["dump_seed_1"] == GOLD["scratch_token"] + "/t20s_r01_map1_n.mrc" and d["dump_seed_2"] == GOLD["scratch_token"] + "/t20s_r01_map2_n.mrc"


# ---- H9 / f-2: score selection --------------------------------------------------------------------------------------------

def test_shape_phase_residuals_occupancies_exactly():
    """src/pyp/analysis/scores.py:300-761 on 240 SPA rows (8 films) and 336 tomography rows (8 series x 6 particles x 7 tilts):
    thresholds 0 / fraction / 1, orientation x defocus groups, every window, odd / even; OCC column exact, POSITION_IN_STACK as
    the reference leaves it (renumbered 1..M only when the output name ends in `_used.cistem`)."""
    tilts = [-30.0, -20.0, -10.0, 0.0, 10.0, 20.0, 30.0]
    for c in GOLD["selection_cases"]:
        rows = SEL[c["input"] + "_in"].astype(np.float32).astype(np.float64)     # what the reference read back from its .cistem file
        a = c["args"]
        if c["input"] == "spa":
            table = {str(f): {str(t): 0.0 for t in range(5)} for f in range(8)}
        else:
            table = {str(f): {str(t): ang for t, ang in enumerate(tilts)} for f in range(8)}
        ta = select.tilt_angles_from_table(rows, table)
        np.random.seed(1234)
        got = select.select_particles(rows, c["threshold"], angles=c["angles"], defocuses=c["defocuses"], mindefocus=a["mindefocus"],
                                      maxdefocus=a["maxdefocus"], firstframe=a["firstframe"], lastframe=a["lastframe"], mintilt=a["mintilt"],
                                      maxtilt=a["maxtilt"], minazh=a["minazh"], maxazh=a["maxazh"], minscore=a["minscore"], maxscore=a["maxscore"],
                                      odd=a["odd"], even=a["even"], renumber=c["output_name"].endswith("_used.cistem"), tilt_angles=ta)
        want_occ = SEL[c["key"] + "_occ"]
        assert np.array_equal(got[:, C["OCCUPANCY"]], want_occ), (c["tag"], c["threshold"], int((got[:, C["OCCUPANCY"]] != want_occ).sum()))
        assert np.array_equal(got[:, C["POSITION_IN_STACK"]], SEL[c["key"] + "_pos"]), c["tag"]
        assert int((got[:, C["OCCUPANCY"]] == 0).sum()) == c["zeroed"] and c["other_columns_unchanged"]
        keep = [j for j in range(32) if j not in (C["OCCUPANCY"], C["POSITION_IN_STACK"])]
        assert np.array_equal(got[:, keep], rows[:, keep])


def test_compute_global_weights_text():
    """src/pyp/inout/metadata/core.py:3039-3075: the external weights file, incl. -1.0 for exposure indices without rows."""
    for g in GOLD["global_weights"]:
        t = SEL[g["input"] + "_in"].copy()                 # (this function is handed the float64 table, not a file)
        t[::g["occ_zero_stride"], C["OCCUPANCY"]] = 0.0
        if g["tind_5_moved_to"] is not None:
            t[t[:, C["TIND"]] == 5, C["TIND"]] = g["tind_5_moved_to"]
        w = dose.compute_global_weights(t)
        want = [float(x) for x in g["text"].split("\n")]
        assert len(w) == len(want) and np.allclose(w, want, rtol=1e-13, atol=0)
        assert [x == -1.0 for x in w] == [x == -1.0 for x in want]


# ---- H13: csp argv, region parameter files ------------------------------------------------------------------------------------

def test_csp_argv_of_both_branches_parse():
    """local_run.py:364-376 / :392-404 (region branch) and :451-463 (global branch): every command line goes through the
    program's own argv parser; output names follow what merge_alignment_parameters globs (`_??????_??????`,
    `_region????_??????_??????`, align/core.py:1027, :1119)."""
    n = 0
    for key, g in GOLD["csp_split_commands"].items():
        for cmd in g["commands"]:
            argv, log = csp_cli.split_command(cmd)
            a = csp_cli.parse_argv(argv[1:])
            assert argv[0].endswith("/CSP/csp") and a["ext_file"] == a["param_file"].replace(".cistem", "_extended.cistem")
            if key.startswith("region"):
                assert "_region" in a["param_file"] and a["first"] == a["last"] and a["images"] == "frealign/ts1.mrc" and a["flag"] == "1"
                assert a["mode"] == {"region_mode2": 5, "region_mode3": 6, "region_mode4": 4}[key]
                import fnmatch
                assert fnmatch.fnmatch(os.path.basename(csp_cli._out_names(a["param_file"], a["first"], a["last"])[0]),
                                       "ts1_r01_02_region????_??????_??????.cistem")
            else:
                frames = key.endswith("frames1")
                assert a["images"] == ("frames_csp.txt" if frames else "frealign/ts1.mrc")
                assert a["mode"] == {"global_mode-2": -2, "global_mode2": 5, "global_mode3": 3 if frames else 6}[key.split("_frames")[0]]
                assert a["flag"] == ("0" if (frames and a["mode"] == 3) else "1")
            n += 1
    assert n > 40
    # region jobs: one per PIND of the region (particle modes) or per TIND (micrograph modes), last region first
    reg = GOLD["regions"]["files"]
    cm = [csp_cli.parse_argv(csp_cli.split_command(c)[0][1:]) for c in GOLD["csp_split_commands"]["region_mode2"]["commands"]]
    want = [(r["file"], p) for r in reg[::-1] for p in r["pind"]]
    assert [(a["param_file"], a["first"]) for a in cm] == want
    cm = [csp_cli.parse_argv(csp_cli.split_command(c)[0][1:]) for c in GOLD["csp_split_commands"]["region_mode3"]["commands"]]
    assert [(a["param_file"], a["first"]) for a in cm] == [(r["file"], t) for r in reg[::-1] for t in r["tind"]]


def test_region_parameter_files_of_split_parameter_file_read_back():
    """particle_cspt.py:141-208: a region file holds the rows of the region's particles with RIND = the region's index, its
    extended file ALL particles and the tilts re-keyed (TIND, new RIND)."""
    full = cistem.read_parameters(os.path.join(HERE, "r03_ts1_r01_02.cistem"))
    fext = cistem.read_extended(os.path.join(HERE, "r03_ts1_r01_02_extended.cistem"))
    seen = []
    for k, r in enumerate(GOLD["regions"]["files"]):
        rows = cistem.read_parameters(os.path.join(HERE, "r03_" + os.path.basename(r["file"])))
        ext = cistem.read_extended(os.path.join(HERE, "r03_" + os.path.basename(r["file"]).replace(".cistem", "_extended.cistem")))
        assert sorted(set(rows[:, C["PIND"]].astype(int))) == r["pind"] and np.all(rows[:, C["RIND"]] == k)
        assert ext["particles"].shape == fext["particles"].shape and np.array_equal(ext["particles"], fext["particles"])
        assert np.all(ext["tilts"][:, 1] == k) and sorted(ext["tilts"][:, 0].astype(int)) == r["tind"]
        sub = full[np.isin(full[:, C["PIND"]], r["pind"])]
        keep = [j for j in range(32) if j != C["RIND"]]
        assert np.array_equal(rows[:, keep], sub[:, keep])
        seen += r["pind"]
    assert sorted(seen) == list(range(12))


def test_merge_of_region_job_outputs_like_parameters_merge():
    """cistem_star_file.py:655-692 through particle_cspt.py:96-138: job outputs are concatenated in sorted file order, the
    extended blocks merged over the original (later files win per key)."""
    g = GOLD["region_merge"]
    jobs = [os.path.join(HERE, "r03_job_" + j) for j in g["jobs"]]
    rows, particles, tilts = csp_cli.merge_alignment_parameters(
        jobs, [os.path.join(HERE, "r03_ts1_r01_02_extended.cistem")] + [j.replace(".cistem", "_extended.cistem") for j in jobs])
    assert np.allclose(rows, np.array(g["rows"]), rtol=0, atol=0)
    assert {str(float(int(p[0]))): float(p[1]) for p in particles} == g["particle_shift_x"]
    keys = {}
    for t in tilts:
        keys.setdefault(str(float(int(t[0]))), []).append(int(t[1]))
    assert {k: sorted(v) for k, v in keys.items()} == g["tilt_keys"]
