"""Round-2 golden fixtures produced by running the reference's own Python (tests/golden/gen_golden_r02.py): particle ranges of
create_split_commands, the bimodal score threshold, cclin parameter files, the statistics file format."""
import json
import os

import numpy as np

from pyp_amd import dist, select
from pyp_amd.formats import parfile
from pyp_amd.surface import cli

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GOLD = json.load(open(os.path.join(HERE, "golden_r02.json")))


def test_split_ranges_match_reference_create_split_commands():
    """src/pyp/system/local_run.py:507-516 run on (frames, cores) incl. the cases SURVEY.md §8c(6) lists."""
    for key, want in GOLD["split_ranges"].items():
        frames, cores = (int(x) for x in key.split(","))
        got = dist.split_ranges(frames, cores)
        assert [list(r) for r in got] == want["ranges"] and len(got) == want["count"], key
        assert ["%07d_%07d" % r for r in got] == want["ranger"]                        # the range tag of the output file names
    assert len(GOLD["split_ranges"]["100000,64"]["ranges"]) == 64


def test_optimal_threshold_matches_reference_statistics():
    for case in GOLD["optimal_threshold"]:
        if "constant" in case:
            assert select.optimal_threshold(np.full(case["n"], case["constant"])) == case["threshold"] == 1
            continue
        g = np.random.default_rng(case["seed"])
        s = np.concatenate([g.normal(case["means"][0], case["sigmas"][0], case["counts"][0]),
                            g.normal(case["means"][1], case["sigmas"][1], case["counts"][1])])
        np.random.seed(1234)                                        # the mixture's k-means initialisation draws from numpy's global state
        got = select.optimal_threshold(s, "optimal")
        assert abs(got - case["threshold"]) < 1e-6 * max(1.0, abs(case["threshold"])), case


def test_automatic_cutoff_and_tomo_rules():
    rng = np.random.default_rng(3)
    m = 600
    rows = np.zeros((m, 32))
    rows[:, 0] = np.arange(1, m + 1)
    rows[:, 11] = 100.0
    rows[:, 6] = 15000.0
    rows[:, 14] = np.concatenate([rng.normal(6, 1.0, 200), rng.normal(18, 2.0, 400)])       # junk and good particles
    np.random.seed(1)
    out = select.select_particles(rows, threshold=0)
    kept = out[:, 11] > 0
    assert kept[200:].mean() > 0.97 and kept[:200].mean() < 0.05
    # tomography: 40 particles x 15 tilts; the decision is per particle, from the low tilts only
    npart, tl = 40, np.arange(-42, 43, 6.0)
    rows = np.zeros((npart * len(tl), 32))
    rows[:, 0] = np.arange(1, len(rows) + 1)
    rows[:, 11] = 100.0
    rows[:, 6] = 15000.0
    rows[:, 26] = np.repeat(np.arange(npart), len(tl))
    tilt = np.tile(tl, npart)
    good = np.repeat(np.arange(npart) >= 10, len(tl))
    rows[:, 14] = np.where(good, 20.0, 8.0) + rng.normal(0, 0.5, len(rows)) - 0.1 * np.abs(tilt)
    out = select.select_particles(rows, threshold=0.7, tilt_angles=tilt)
    occ = out[:, 11].reshape(npart, len(tl))
    # the cut sits at sorted(particle means over |tilt| <= 12)[int(39 * 0.3)] = among the good particles: every junk particle
    # goes as a whole, high tilts included; a good particle is kept or dropped as a whole too
    gone = (occ == 0).all(axis=1)
    assert gone[:10].all() and gone[10:].sum() <= 3 and ((occ == 0) | (occ == 100)).all() and ((occ == 0).any(axis=1) == gone).all()
    out = select.select_particles(rows, threshold=1.0, tilt_angles=tilt, mintilt=-30, maxtilt=30)
    assert ((out[:, 11] > 0) == (np.abs(tilt) <= 30)).all()


def test_cclin_parameter_files_read_like_the_reference():
    for key, ncol in (("cclin", 13), ("cclin_ext", 42)):
        data, version, ext, pro, epi = parfile.read(os.path.join(HERE, f"par_{key}.par"))
        want = np.load(os.path.join(HERE, f"par_{key}_readback.npy"))
        assert data.shape == want.shape == (4, ncol) and np.array_equal(data, want)
        assert version == GOLD[f"par_{key}_version"][0] == "cclin"
        assert len(pro) == 3 and pro[0].startswith("C FREALIGN")


def test_statistics_file_format_matches_reference_writer(tmp_path):
    """merge3d's <name>_statistics.txt rows = numpy.savetxt(fmt 7 x %14.5f) as the reference rewrites the file
    (src/pyp/postprocess/core.py:219-221)."""
    import io
    buf = io.StringIO()
    np.savetxt(buf, np.array(GOLD["statistics_txt"]["input"]), fmt="%12.6f")          # the generator handed the reference this text
    st = np.loadtxt(io.StringIO(buf.getvalue()))
    assert cli.format_statistics_rows(st) == GOLD["statistics_txt"]["text"]
    assert np.loadtxt(__import__("io").StringIO(GOLD["statistics_txt"]["text"]), comments=["C"]).shape == st.shape
