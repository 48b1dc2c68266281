"""CPU: the C-ABI library loads and exports every symbol include/signnet_hip.h declares; the ctypes
mirrors of the parameter structs have the C layout.  No compute calls (no GPU here)."""
import ctypes
import os
import re
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "signnet_hip.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sn_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    from signnet_basisnet_amd import build
    build.build()
    from signnet_basisnet_amd import _lib
    return _lib.lib()


def test_every_declared_symbol_is_exported(lib):
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in signnet_hip.h but not exported by libsignnet_hip.so"


def test_python_binding_covers_the_header(lib):
    from signnet_basisnet_amd import _lib
    bound = set(_lib.SIGNATURES) | {"sn_last_error", "sn_packed_weight_floats", "sn_split_packed_bytes", "sn_phi_bins_bound",
                                     "sn_ign_contract_scratch_floats", "sn_evd_work_ints",
                                     "sn_linear_wgrad_scratch_floats", "sn_layernorm_bwd_scratch_floats",
                                     "sn_bn_act_bwd_scratch_floats", "sn_embedding_bwd_scratch_floats", "sn_gatedgcn_max_edges"}
    assert set(declared_symbols()) == bound


def test_version_and_error_string(lib):
    assert lib.sn_version() == 1
    # argument validation happens on the host before any launch: callable without a GPU
    rc = lib.sn_pack_weight_f32(None, 4, 4, 4, None, None)
    assert rc == -1 and b"sn_pack_weight_f32" in lib.sn_last_error()
    assert lib.sn_packed_weight_floats(128, 128) == 64 * 256
    assert lib.sn_evd_work_ints(128) == 3 * 128 + 8
    rc = lib.sn_laplacian_evd_f32(None, 0, None, 0, 0, 0, None, None, None, 0, None, 0, 0, None, None, None)
    assert rc == -1 and b"sn_laplacian_evd_f32" in lib.sn_last_error()
    assert lib.sn_packed_weight_floats(108, 6) == 7 * 1 * 256
    assert lib.sn_split_packed_bytes(128, 128) == 8 * (3 * 4 + 3) * 1024
    assert lib.sn_split_packed_bytes(108, 40) == 7 * (3 * 2 + 3) * 1024
    assert lib.sn_phi_bins_bound(128, 16) == 128 * 16 + 1 and lib.sn_phi_bins_bound(10, 0) == 641


def test_struct_layouts_match_the_header():
    """sizeof/offsetof of the parameter structs, C compiler vs ctypes."""
    from signnet_basisnet_amd import fused, ops
    prog = r'''
#include <stdio.h>
#include <stddef.h>
#include "signnet_hip.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(sn_plan_bins), sizeof(sn_phi_layer), sizeof(sn_phi_params),
         sizeof(sn_rho_layer), sizeof(sn_rho_params), sizeof(sn_gnn_layer), sizeof(sn_gnn_params),
         offsetof(sn_gnn_params, layers));
  printf("%zu %zu %zu\n", offsetof(sn_phi_params, layers), offsetof(sn_rho_params, layers), offsetof(sn_rho_params, pe_w1));
  return 0;
}'''
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "t.c")
        open(c, "w").write(prog)
        exe = os.path.join(td, "t")
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe], check=True)
        out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()
    got = [int(v) for v in out]
    S = ctypes.sizeof
    want = [S(ops._PlanBinsC), S(fused._PhiLayer), S(fused._PhiParams), S(fused._RhoLayer), S(fused._RhoParams),
            S(fused._GnnLayer), S(fused._GnnParams), fused._GnnParams.layers.offset,
            fused._PhiParams.layers.offset, fused._RhoParams.layers.offset, fused._RhoParams.pe_w1.offset]
    assert got == want


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from signnet_basisnet_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU/PyTorch fallback"):
        _lib.lib()
