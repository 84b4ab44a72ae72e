"""Region-based constrained refinement, caller side (tests/caller_regions.py) against the reference's own output on a toy tilt series
(tests/golden/gen_golden_r03.py: findSpecimenBounds, divide2regions, sort_particles_regions, split_parameter_file,
create_csp_split_commands run in the build container)."""
import json
import os

import numpy as np

import caller_regions as regions
from pyp_amd.formats import cistem

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GOLD = json.load(open(os.path.join(HERE, "golden_r03.json")))


def _series():
    rows = cistem.read_parameters(os.path.join(HERE, "r03_ts1_r01_02.cistem"))
    ext = cistem.read_extended(os.path.join(HERE, "r03_ts1_r01_02_extended.cistem"))
    return rows, ext["particles"], ext["tilts"]


def test_bounds_grid_and_sorting_match_reference():
    rows, particles, tilts = _series()
    g = GOLD["regions"]
    assert np.allclose(particles[:, 7:10], np.array(g["positions"]), atol=1e-3)
    bl, tr = regions.find_specimen_bounds(particles, [1000, 1000, 300])
    assert [list(map(float, bl)), list(map(float, tr))] == g["bounds"]
    corners, size = regions.divide_regions(bl, tr, split_x=2, split_y=2, split_z=1)
    assert [list(map(float, c)) for c in corners] == g["corners"] and list(map(float, size)) == g["size"]
    assert regions.sort_particles_regions(particles, corners, size) == g["sorted_particles"]
    assert regions.sort_particles_regions(particles, corners, size, per_particle=True) == [[i] for i in range(12)]
    # overlap and a z split: the grid formula itself
    c2, s2 = regions.divide_regions([0, 0, 0], [100, 60, 30], 4, 3, 2, overlap=0.2)
    assert len(c2) == 24 and np.isclose(s2[0], 100 / (4 - 0.8 + 0.2)) and np.isclose(c2[-1][0] + s2[0], 100) and np.isclose(c2[-1][2] + s2[2], 30)


def test_region_files_are_byte_identical_to_the_references(tmp_path):
    rows, particles, tilts = _series()
    g = GOLD["regions"]
    pf = str(tmp_path / "ts1_r01_02.cistem")
    split = regions.split_parameter_file(rows, particles, tilts, pf, g["sorted_particles"])
    assert len(split) == len(g["files"])
    for (fn, pinds, tinds), want in zip(split, g["files"]):
        assert os.path.basename(fn) == os.path.basename(want["file"]) and list(pinds) == want["pind"] and list(tinds) == want["tind"]
        for suffix in (".cistem", "_extended.cistem"):
            ours = open(fn.replace(".cistem", suffix), "rb").read()
            theirs = open(os.path.join(HERE, "r03_" + os.path.basename(fn).replace(".cistem", suffix)), "rb").read()
            assert ours == theirs, fn + suffix


def test_command_lines_equal_create_csp_split_commands():
    g = GOLD["csp_split_commands"]
    csp, pf, stack = "/opt/pyp/external/CSP/csp", "frealign/maps/ts1_r01_02.cistem", "frealign/ts1_stack.mrc"
    ptl, scan = list(range(10)), list(range(5))
    for mode, frames in ((-2, False), (2, False), (3, False), (3, True), (2, True)):
        cmds, movies = regions.csp_split_commands(csp, pf, mode, "ts1_r01_02", stack, ptl, scan, increment=3, use_frames=frames)
        want = g[f"global_mode{mode}_frames{int(frames)}"]
        assert cmds == want["commands"] and movies == want["movie_list"] and len(cmds) + 1 == want["count"], (mode, frames)
    split = [(r["file"], r["pind"], r["tind"]) for r in GOLD["regions"]["files"]]
    for mode in (2, 3, 4):
        cmds, movies = regions.csp_split_commands(csp, split, mode, "ts1_r01_02", stack, ptl, scan)
        assert cmds == g[f"region_mode{mode}"]["commands"] and movies == [], mode
