"""CPU-side checks of the host layer: the C-ABI library loads and exports every symbol include/ppm.h
declares, the ctypes structs match the C layout, and the product path fails loudly without a GPU."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from pyp_amd import abi, lib, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "ppm.h")).read()
    declared = set(re.findall(r"\b(ppm_[a-z_]+)\s*\(", hdr))
    assert declared == set(lib.EXPORTS)
    L = lib.load()
    for name in declared:
        assert hasattr(L, name), name
    assert b"gfx950" in L.ppm_version()


def test_struct_layout_matches_header():
    src = r'''
    #include <stdio.h>
    #include <stddef.h>
    #include "ppm.h"
    int main(void){
      printf("%zu %zu %zu\n", sizeof(ppm_refine_cfg), sizeof(ppm_recon_cfg), sizeof(ppm_final_cfg));
      printf("%zu %zu %zu %zu\n", offsetof(ppm_refine_cfg, top_hits), offsetof(ppm_refine_cfg, global_search), offsetof(ppm_refine_cfg, mask_falloff), offsetof(ppm_refine_cfg, local_shift_step));
      printf("%zu %zu\n", offsetof(ppm_recon_cfg, split_by_pind), offsetof(ppm_recon_cfg, mask_radius));
      return 0; }'''
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", os.path.join(d, "t"), os.path.join(d, "t.c")])
        out = subprocess.check_output([os.path.join(d, "t")]).decode().split()
    got = [C.sizeof(abi.RefineCfg), C.sizeof(abi.ReconCfg), C.sizeof(abi.FinalCfg),
           abi.RefineCfg.top_hits.offset, abi.RefineCfg.global_search.offset, abi.RefineCfg.mask_falloff.offset,
           abi.RefineCfg.local_shift_step.offset, abi.ReconCfg.split_by_pind.offset, abi.ReconCfg.mask_radius.offset]
    assert got == [int(x) for x in out]


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(lib.PpmError) as e:
        lib.init(0)
    assert "ERROR" in str(e.value)
    from pyp_amd import host
    with pytest.raises(lib.PpmError):
        host.Reference(np.zeros((32, 32, 32), np.float32))


def test_product_never_imports_the_oracle():
    """Nothing under pyp_amd/ may import, link, dlopen or call the oracle (it is test infrastructure)."""
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|libppm_oracle|ppm_oracle\.c|oracle\.py|\borc_[a-z_]+\s*\(", re.M)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pyp_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", "Makefile")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not pat.search(txt), (dirpath, f)


def test_euler_convention_golden(golden_dir):
    import json
    g = json.load(open(os.path.join(golden_dir, "golden.json")))
    for e in g["euler"]:
        psi, theta, phi = e["in"]
        m = synth.pyp_matrix(psi, theta, phi)
        assert np.allclose(m, np.array(e["matrix"]), atol=1e-12)
        assert np.allclose(synth.angles_from_pyp_matrix(np.array(e["matrix"])), e["out"], atol=1e-9)
        # the matrix PYP writes down is ours with all three angles negated
        assert np.allclose(m, synth.euler_matrix(-psi, -theta, -phi), atol=1e-12)


def test_class_occupancies_match_reference_golden(golden_dir):
    """tests/golden/occupancy_3class.npz was written by the reference's occupancy_extended (gen_golden_occ.py); the .cistem
    codec stores float32, so inputs are taken at float32 precision and outputs compared at float32 resolution."""
    import os
    from pyp_amd import classify
    g = np.load(os.path.join(golden_dir, "occupancy_3class.npz"))
    f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
    logp, sigma, occ_in = f32(g["logp"]), f32(g["sigma"]), f32(g["occ_in"])
    occ, sig = classify.occupancies_from_logp(logp, sigma, occ_in.mean(axis=1))
    assert np.allclose(occ, g["occ_out"], rtol=2e-6, atol=2e-5)
    assert np.allclose(sig[None, :].repeat(3, 0), g["sigma_out"], rtol=2e-6, atol=1e-6)
    assert np.allclose(occ.sum(axis=0), 100.0)
    assert occ[1, 3] == 0.0 and occ[2, 3] > 0.0                     # the delta < 10 window
    # table-level wrapper: only OCCUPANCY and SIGMA change
    K, M = logp.shape
    tabs = []
    for k in range(K):
        t = np.zeros((M, 32)); t[:, 0] = np.arange(1, M + 1); t[:, 11] = occ_in[k]; t[:, 12] = logp[k]; t[:, 13] = sigma[k]; t[:, 14] = 7.0
        tabs.append(t)
    new = classify.update_class_rows(tabs)
    for k in range(K):
        assert np.allclose(new[k][:, 11], g["occ_out"][k], rtol=2e-6, atol=2e-5) and np.allclose(new[k][:, 13], g["sigma_out"][k], rtol=2e-6, atol=1e-6)
        keep = [c for c in range(32) if c not in (11, 13)]
        assert np.array_equal(new[k][:, keep], tabs[k][:, keep])


def test_score_selection_rules():
    """pyp_amd.select (unpinned restatement of scores.py shape_phase_residuals, SPA branch)."""
    from pyp_amd import select
    rng = np.random.default_rng(3)
    M = 400
    rows = np.zeros((M, 32)); rows[:, 0] = np.arange(1, M + 1)
    rows[:, 2] = rng.uniform(0, 360, M); rows[:, 6] = rng.uniform(8000, 24000, M); rows[:, 11] = 100.0
    rows[:, 14] = rng.uniform(0, 30, M); rows[:, 27] = np.arange(M) % 40
    out = select.select_particles(rows, threshold=0.75)
    kept = out[:, 11] > 0
    cut = np.sort(rows[:, 14])[int((M - 1) * 0.25)]
    assert np.array_equal(kept, rows[:, 14] >= cut) and abs(kept.mean() - 0.75) < 0.01
    keep_cols = [c for c in range(32) if c != 11]
    assert np.array_equal(out[:, keep_cols], rows[:, keep_cols])
    assert (select.select_particles(rows, threshold=1.0)[:, 11] > 0).all()
    assert (select.select_particles(rows, threshold=250)[:, 11] > 0).all()        # absolute counts: no-op, like the reference
    d = select.select_particles(rows, threshold=1.0, mindefocus=10000, maxdefocus=20000)
    assert np.array_equal(d[:, 11] > 0, (rows[:, 6] >= 10000) & (rows[:, 6] <= 20000))
    a = select.select_particles(rows, threshold=1.0, minazh=30, maxazh=150)
    assert np.array_equal(a[:, 11] > 0, (np.mod(rows[:, 2], 180) >= 30) & (np.mod(rows[:, 2], 180) <= 150))
    fr = select.select_particles(rows, threshold=1.0, firstframe=5, lastframe=20)
    assert np.array_equal(fr[:, 11] > 0, (rows[:, 27] >= 5) & (rows[:, 27] <= 20))
    assert (select.select_particles(rows, threshold=1.0, odd=True)[::2, 11] == 0).all()
    assert (select.select_particles(rows, threshold=1.0, even=True)[1::2, 11] == 0).all()
    s = select.select_particles(rows, threshold=1.0, minscore=0.1, maxscore=0.9)
    lo, hi = rows[:, 14].min(), rows[:, 14].max()
    assert np.array_equal(s[:, 11] > 0, (rows[:, 14] >= lo + 0.1 * (hi - lo)) & (rows[:, 14] <= hi - 0.1 * (hi - lo)))
    g = select.select_particles(rows, threshold=0.5, angles=3, defocuses=2)          # grouped thresholds keep roughly half
    assert 0.3 < (g[:, 11] > 0).mean() < 0.8
    auto = select.select_particles(rows, threshold=0)          # reconstruct_cutoff = 0: the bimodal automatic threshold
    assert auto.shape == rows.shape and 0 < (auto[:, 11] > 0).sum() <= len(rows)


def test_host_read_fills_a_buffer_from_a_file_with_any_thread_count(tmp_path):
    """ppm_host_read (the executables' reader stage): no device call, so it runs here.  Parts start at MB boundaries; offsets and
    lengths that are not multiples of anything; a short file and an empty request."""
    import os
    from pyp_amd import lib
    L = lib.load()
    a = np.random.default_rng(0).integers(0, 255, (9 << 20) + 12345, dtype=np.uint8)
    p = tmp_path / "blob.bin"
    p.write_bytes(a.tobytes())
    fd = os.open(p, os.O_RDONLY)
    try:
        for nt in (1, 3, 8, 16, 99):
            out = np.zeros(len(a) - 1001, np.uint8)
            assert L.ppm_host_read(fd, 1001, out.ctypes.data, out.size, nt) == 0
            assert np.array_equal(out, a[1001:])
        out = np.zeros(len(a) + 10, np.uint8)
        assert L.ppm_host_read(fd, 0, out.ctypes.data, out.size, 4) == -5 and "short read" in lib.last_error()
        assert L.ppm_host_read(fd, 0, None, 0, 4) == 0
        assert L.ppm_host_read(-1, 0, out.ctypes.data, 10, 4) == -22
    finally:
        os.close(fd)
