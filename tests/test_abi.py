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
                                     "sn_bn_act_bwd_scratch_floats", "sn_embedding_bwd_scratch_floats", "sn_embedding_bwd_layers_scratch_floats", "sn_ign_mlp_supported",
                                     "sn_gatedgcn_max_edges",
                                     "sn_train_linear_bwd_part_floats", "sn_train_scalar_mlp_work_doubles"}
    # (sn_batch_plan_early_supported is in SIGNATURES: it returns 0 / 1)
    assert set(declared_symbols()) == bound


def test_version_and_error_string(lib):
    from signnet_basisnet_amd import _lib as L
    hdr = open(os.path.join(ROOT, "include", "signnet_hip.h")).read()
    import re
    assert lib.sn_version() == L.ABI_VERSION == int(re.search(r"#define SN_ABI_VERSION (\d+)", hdr).group(1)) == 2
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


def test_round2_entry_points_validate_their_arguments_on_the_host(lib):
    """Every entry point added in round 2 rejects null / out-of-range arguments before any launch (rc -1, message naming the function):
    callable without a GPU."""
    import ctypes as C
    i64 = (C.c_int64 * 1)(4)
    cases = {
        "sn_dense_attention_f32": (None, None, None, 1, 8, 2, 4, None, None, None),
        "sn_dense_attention_bwd_f32": (None,) * 6 + (1, 8, 2, 4) + (None,) * 5,
        "sn_deepsigns_phi_f32": (None, None, 8, None, None, None, None, 8, None, None),
        "sn_mlp_chain_f32": (None, 8, 4, 8, None, 0, None, 2, 64, None, 8, 8, None),
        "sn_gin_net_fused_f32": (None, None, 0, None, None, 8, 8, None, 4, None, None, None, None, None, None, 0, None),
        "sn_transformer_net_fused_f32": (None, None, 0, None, None, 8, 8, None, 64, None, 4, None, None, None, None, None, None, 0, None),
        "sn_gatedgcn_fused_f32": (None,) * 4 + (1,) + (None,) * 6,
        "sn_eigenspace_group": (None, 4, 5) + (None,) * 8,
        "sn_eigenspace_projectors_f32": (None, 4, 4, None, None, 1, None, None),
        "sn_ign_contract_eigvecs_f32": (None, 4, 4, None, None, 1, 1, None, None),
        "sn_pna_aggregate_f32": (None, 4, None, 4, 4, 2, None, None, 1.0, None, 52, None),
        "sn_ign_mlp_f32": (None, 1, 8, 32, 1, None, None, None),
        "sn_deepsets_tail_f32": (None, 8, None, None, None),
        "sn_masked_linear_blockbias_f32": (None, 4, 4, 4, None, 4, None, None, 2, 4, 0, None, None, None, 4, None),
        "sn_pna_aggregate_gather_f32": (None, 4, None, 4, None, 4, None, 4, 4, 2, None, None, None, 1.0, None, 52, 0, None),
        "sn_grouped_linear_f32": (None, 8, 4, 2, 4, 4, None, None, None, None, None, None, 8, None),
        "sn_pna_aggregate_bwd_f32": (None, 4, 4, 2, None, None, 1.0, None, 52, None, None, None),
        "sn_edge_attention_f32": (None,) * 4 + (2, 2, 4) + (None,) * 5,
        "sn_edge_attention_bwd_f32": (None,) * 6 + (2, 2, 2, 4) + (None,) * 12,
        "sn_edge_rows_sum_f32": (None, 4, 4, 2, None, None, None, 4, None),
        "sn_act_bwd_f32": (None, None, 2, 4, None, 1, 0.01, None, None),
        "sn_pointwise_f32": (None, 4, 2, 4, None, None, None, 0, 0.0, None, 4, None, 4, None),
        "sn_embedding_sum_bwd_f32": (None, 1, 1, 4, None, i64, 8, None, None, None, None),
        "sn_embedding_sum_layers_f32": (None, 1, 1, 4, 2, None, i64, 8, None, None, None),
        "sn_embedding_sum_bwd_layers_f32": (None, 1, 1, 4, 2, None, i64, 8, None, None, None, None),
    }
    cases.update({      # round 6
        "sn_train_reduce_jobs_f32": (None, 0, None),
        "sn_train_bn_bwd_f32": (None, 4, None, 4, 2, 1, 4, None, 0, None, None, 0, None, None, None, None, None, 0, None),
        "sn_clock_probe": (0, None, None),
    })
    for name, args in cases.items():
        rc = getattr(lib, name)(*args)
        assert rc == -1 and name.encode() in lib.sn_last_error(), (name, rc, lib.sn_last_error())
    assert lib.sn_gatedgcn_max_edges(68) == 176 and lib.sn_gatedgcn_max_edges(128) == 0
    # ceil(1000/64) = 16 chunks: 64 partial rows x 8 channels (per gradient plane) + 64 ids + 1 count each, + 16
    assert lib.sn_embedding_bwd_scratch_floats(1000, 1, i64, 8) == 16 * 64 * 8 + 16 * 64 + 16 + 16
    assert lib.sn_embedding_bwd_layers_scratch_floats(1000, 6, 8) == 6 * 16 * 64 * 8 + 16 * 64 + 16 + 16
    assert lib.sn_phi_bins_bound(128, -8) == 128 * 8 + 1                              # full-slot mode: |kmax| bins per column


def test_struct_layouts_match_the_header():
    """sizeof/offsetof of the parameter structs, C compiler vs ctypes."""
    from signnet_basisnet_amd import basisnet, dgl_nets, fused, ops, train_stage
    prog = r'''
#include <stdio.h>
#include <stddef.h>
#include "signnet_hip.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(sn_plan_bins), sizeof(sn_phi_layer), sizeof(sn_phi_params),
         sizeof(sn_rho_layer), sizeof(sn_rho_params), sizeof(sn_gnn_layer), sizeof(sn_gnn_params),
         offsetof(sn_gnn_params, layers));
  printf("%zu %zu %zu\n", offsetof(sn_phi_params, layers), offsetof(sn_rho_params, layers), offsetof(sn_rho_params, pe_w1));
  printf("%zu %zu %zu %zu\n", sizeof(sn_gatedgcn_layer), sizeof(sn_gatedgcn_params), offsetof(sn_gatedgcn_params, layers),
         offsetof(sn_gatedgcn_params, ro_w0));
  printf("%zu %zu\n", sizeof(sn_ign_mlp_params), offsetof(sn_ign_mlp_params, fc2_b));
  printf("%zu %zu %zu\n", sizeof(sn_deepsets_tail_params), offsetof(sn_deepsets_tail_params, width), offsetof(sn_deepsets_tail_params, gamma));
  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(sn_train_linear_args), offsetof(sn_train_linear_args, in_scale),
         offsetof(sn_train_linear_args, stat_part), sizeof(sn_train_linear_bwd_args), offsetof(sn_train_linear_bwd_args, x_mean),
         offsetof(sn_train_linear_bwd_args, dot_part), sizeof(sn_train_scalar_mlp_args), offsetof(sn_train_scalar_mlp_args, w2),
         offsetof(sn_train_scalar_mlp_args, column_state));
  printf("%zu %zu %zu\n", sizeof(sn_train_post_args), offsetof(sn_train_post_args, sums_part), offsetof(sn_train_post_args, dot_part));
  printf("%zu %zu %zu %zu %zu %zu\n", offsetof(sn_train_linear_args, fin_eps), offsetof(sn_train_linear_args, fin_count),
         offsetof(sn_train_linear_bwd_args, fin_coef), offsetof(sn_train_linear_bwd_args, fin_dot_out), sizeof(sn_train_reduce_job),
         offsetof(sn_train_reduce_job, out));
  printf("%zu %zu\n", offsetof(sn_train_linear_bwd_args, merge_sums), offsetof(sn_train_linear_bwd_args, merge_accumulate));
  printf("%zu %zu %zu %zu\n", sizeof(sn_plan_early), offsetof(sn_plan_early, max_graph_edges), offsetof(sn_plan_early, host), offsetof(sn_plan_bins, phi_bin_mem));
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
            fused._PhiParams.layers.offset, fused._RhoParams.layers.offset, fused._RhoParams.pe_w1.offset,
            S(dgl_nets._GatedLayerC), S(dgl_nets._GatedParamsC), dgl_nets._GatedParamsC.layers.offset, dgl_nets._GatedParamsC.ro_w0.offset,
            S(basisnet._IgnMlpParams), basisnet._IgnMlpParams.fc2_b.offset,
            S(basisnet._DeepSetsTailParams), basisnet._DeepSetsTailParams.width.offset, basisnet._DeepSetsTailParams.gamma.offset,
            S(train_stage._LinArgs), train_stage._LinArgs.in_scale.offset, train_stage._LinArgs.stat_part.offset,
            S(train_stage._BwdArgs), train_stage._BwdArgs.x_mean.offset, train_stage._BwdArgs.dot_part.offset,
            S(train_stage._SMlpArgs), train_stage._SMlpArgs.w2.offset, train_stage._SMlpArgs.column_state.offset,
            S(train_stage._PostArgs), train_stage._PostArgs.sums_part.offset, train_stage._PostArgs.dot_part.offset,
            train_stage._LinArgs.fin_eps.offset, train_stage._LinArgs.fin_count.offset, train_stage._BwdArgs.fin_coef.offset,
            train_stage._BwdArgs.fin_dot_out.offset, S(train_stage._ReduceJob), train_stage._ReduceJob.out.offset,
            train_stage._BwdArgs.merge_sums.offset, train_stage._BwdArgs.merge_accumulate.offset,
            S(ops._PlanEarlyC), ops._PlanEarlyC.max_graph_edges.offset, ops._PlanEarlyC.host.offset, ops._PlanBinsC.phi_bin_mem.offset]
    assert got == want


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from signnet_basisnet_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU/PyTorch fallback"):
        _lib.lib()
