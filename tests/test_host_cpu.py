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
