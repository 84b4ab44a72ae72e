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
