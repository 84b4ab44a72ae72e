"""Format parity with the reference's own Python I/O (fixtures made by tests/golden/gen_golden.py,
which ran src/pyp/inout/metadata/{cistem_star_file,frealign_parfile}.py and src/pyp/inout/image/mrc.py)."""
import json
import os

import numpy as np
import pytest

from pyp_amd.formats import cistem, mrc, parfile


@pytest.fixture(scope="module")
def G(golden_dir):
    with open(os.path.join(golden_dir, "golden.json")) as f:
        return json.load(f)


def test_cistem_main_bytes_identical(golden_dir, tmp_path, G):
    data = np.load(os.path.join(golden_dir, "params8_data.npy"))
    out = tmp_path / "mine.cistem"
    cistem.write_parameters(str(out), data)
    ref = open(os.path.join(golden_dir, "params8.cistem"), "rb").read()
    assert out.read_bytes() == ref
    assert len(ref) == G["cistem_main_bytes"] == 8 + 32 * 9 + 8 * 128


def test_cistem_main_read_matches_reference_reader(golden_dir):
    got = cistem.read_parameters(os.path.join(golden_dir, "params8.cistem"))
    want = np.load(os.path.join(golden_dir, "params8_readback.npy"))
    assert got.shape == want.shape == (8, 32)
    assert np.array_equal(got, want)


def test_cistem_extended_roundtrip_and_bytes(golden_dir, tmp_path, G):
    ext = cistem.read_extended(os.path.join(golden_dir, "params8_extended.cistem"))
    assert ext["particles"].shape == (4, 12) and ext["tilts"].shape == (3, 6)
    for k, v in G["ext_particles"].items():
        row = ext["particles"][ext["particles"][:, 0] == float(k)][0]
        assert np.allclose(row, v, rtol=0, atol=0)
    for t, d in G["ext_tilts"].items():
        row = ext["tilts"][ext["tilts"][:, 0] == float(t)][0]
        assert np.allclose(row, d["0.0"], rtol=0, atol=0)
    out = tmp_path / "e_extended.cistem"
    cistem.write_extended(str(out), ext["particles"], ext["tilts"])
    assert out.read_bytes() == open(os.path.join(golden_dir, "params8_extended.cistem"), "rb").read()


def test_cistem_rejects_broken(tmp_path):
    p = tmp_path / "bad.cistem"
    p.write_bytes(b"\x01\x00")
    with pytest.raises(IOError):
        cistem.read_parameters(str(p))
    p.write_bytes(np.array([1, 1], dtype="<i4").tobytes() + np.array([(999, 3)], dtype=[("c", "<i8"), ("d", "<i1")]).tobytes())
    with pytest.raises(IOError):
        cistem.read_parameters(str(p))


def test_cistem_merge_sorts_by_position(tmp_path):
    d = cistem.default_rows(6, 1.0, 300, 2.7, 0.07)
    cistem.write_parameters(str(tmp_path / "a_0000004_0000006.cistem"), d[3:])
    cistem.write_parameters(str(tmp_path / "a_0000001_0000003.cistem"), d[:3])
    m = cistem.merge_parameters([str(tmp_path / "a_0000004_0000006.cistem"), str(tmp_path / "a_0000001_0000003.cistem")])
    assert list(m[:, 0]) == [1, 2, 3, 4, 5, 6]


@pytest.mark.parametrize("key,version,ext", [("new", "new", False), ("new_ext", "new", True),
                                             ("frealignx", "frealignx", False), ("frealignx_ext", "frealignx", True)])
def test_par_text_identical_and_readback(golden_dir, tmp_path, G, key, version, ext):
    arr = np.load(os.path.join(golden_dir, f"par_{key}_in.npy"))
    out = tmp_path / "mine.par"
    parfile.write(str(out), arr, version=version, extended=ext)
    assert out.read_text() == open(os.path.join(golden_dir, f"par_{key}.par")).read()
    data, v, e, pro, epi = parfile.read(os.path.join(golden_dir, f"par_{key}.par"))
    assert [v, e] == G[f"par_{key}_version"]
    want = np.load(os.path.join(golden_dir, f"par_{key}_readback.npy"))
    assert np.allclose(data, want, rtol=1e-5, atol=1e-8)      # the reference tests' tolerance, tests/test_pyp.py:232-267
    assert len(pro) == 3 and epi == []


def test_par_epilogue_and_errors(tmp_path):
    arr = np.zeros((2, 16)); arr[:, 0] = [1, 2]
    p = tmp_path / "x.par"
    parfile.write(str(p), arr, epilogue=["C  NO.  RESOL  RING RAD", "C   1   100.0  0.01"])
    data, v, e, pro, epi = parfile.read(str(p))
    assert data.shape == (2, 16) and len(epi) == 2
    (tmp_path / "empty.par").write_text("C only header\n")
    with pytest.raises(IOError):
        parfile.read(str(tmp_path / "empty.par"))
    (tmp_path / "ragged.par").write_text("1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16\n1 2 3\n")
    with pytest.raises(IOError):
        parfile.read(str(tmp_path / "ragged.par"))


def test_par_cistem_conversion_roundtrip():
    rows = cistem.default_rows(3, 1.2, 300, 2.7, 0.07)
    rows[:, cistem.COL["PSI"]] = [10, 20, 30]
    rows[:, cistem.COL["X_SHIFT"]] = [1.5, -2.5, 0]
    rows[:, cistem.COL["DEFOCUS_1"]] = 15000
    p = parfile.cistem_to_par(rows, parfile.NEW)
    back = parfile.par_to_cistem(p, parfile.NEW, 1.2, 300, 2.7, 0.07)
    for c in ("POSITION_IN_STACK", "PSI", "X_SHIFT", "DEFOCUS_1", "OCCUPANCY", "SCORE", "PIXEL_SIZE"):
        assert np.array_equal(back[:, cistem.COL[c]], rows[:, cistem.COL[c]])


def test_mrc_bytes_identical_and_read(golden_dir, tmp_path, G):
    stack = np.load(os.path.join(golden_dir, "stack_4x8x8.npy"))
    out = tmp_path / "s.mrc"
    mrc.write(stack, str(out))
    ref = open(os.path.join(golden_dir, "stack_4x8x8.mrc"), "rb").read()
    mine = out.read_bytes()
    assert len(mine) == len(ref) == 1024 + 4 * 8 * 8 * 4
    assert mine[1024:] == ref[1024:]
    hm, hr = mrc.read_header(str(out)), mrc.read_header(os.path.join(golden_dir, "stack_4x8x8.mrc"))
    for k, v in G["mrc_header"].items():
        assert hr[k] == pytest.approx(v, rel=1e-6, abs=1e-7)
        assert hm[k] == pytest.approx(v, rel=1e-6, abs=1e-6), k
    assert mine[:1024] == ref[:1024]
    assert np.array_equal(mrc.read(os.path.join(golden_dir, "stack_4x8x8.mrc")), stack)
    assert np.array_equal(mrc.read(str(out), 1, 2), stack[1:3])
    assert np.array_equal(np.asarray(mrc.mmap(str(out))), stack)


def test_mrc_pixel_size_and_errors(tmp_path):
    v = np.zeros((4, 4, 4), np.float32)
    mrc.write(v, str(tmp_path / "v.mrc"), pixel_size=1.5)
    assert mrc.read_header(str(tmp_path / "v.mrc"))["pixel_size"] == pytest.approx(1.5)
    (tmp_path / "t.mrc").write_bytes(b"\0" * 100)
    with pytest.raises(IOError):
        mrc.read_header(str(tmp_path / "t.mrc"))
    with pytest.raises(IOError):
        mrc.read(str(tmp_path / "v.mrc"), 0, 9)
