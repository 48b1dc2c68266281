"""ctypes binding of libsignnet_hip.so (the C ABI declared in include/signnet_hip.h).

There is NO fallback: if the shared library is missing or an entry point fails, the op raises.
PyTorch is used only for device memory and streams.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsignnet_hip.so")

EPI_BIAS, EPI_RELU_PRE, EPI_AFFINE, EPI_RELU, EPI_RESIDUAL = 1, 2, 4, 8, 16
EPI_RESIDUAL_PRE = 64
EPI_LEAKY = 128

_p, _i, _l, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float
ABI_VERSION = 2          # = SN_ABI_VERSION of include/signnet_hip.h (struct layouts of the parameter blocks included)

# name -> argtypes (restype int unless noted).  Must mirror include/signnet_hip.h exactly;
# tests/test_abi.py cross-checks the symbol list against the header.
SIGNATURES = {
    "sn_version": [],
    "sn_device_info": [C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)],
    "sn_clock_probe": [_l, _p, _p],
    "sn_batch_plan": [_p, _l, _l, _p, _l, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    "sn_batch_plan_ex": [_p, _l, _l, _p, _l, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    "sn_batch_plan_early_supported": [_l, _l, _l],                # (returns 0 / 1, not a status)
    "sn_pack_eig_f32": [_p, _p, _p, _p, _p, _p, _l, _i, _p, _p, _p],
    "sn_pack_weight_f32": [_p, _i, _i, _i, _p, _p],
    "sn_pack_weight_t_f32": [_p, _i, _i, _i, _p, _p],
    "sn_pack_split_f32": [_p, _i, _i, _i, _p, _p, _p, _p, _p],
    "sn_gin_aggregate_f32": [_p, _p, _l, _i, _p, _p, _p, _i, _p],
    "sn_gin_aggregate_slab_f32": [_p, _p, _l, _i, _l, _p, _p, _p, _p, _i, _p],
    "sn_gine_aggregate_f32": [_p, _p, _p, _l, _i, _p, _p, _p, _p, _p],
    "sn_masked_linear_f32": [_p, _i, _l, _i, _p, _i, _p, _p, _i, _i, _p, _p, _p, _i, _p, _i, _p],
    "sn_ign_mlp_f32": [_p, _l, _i, _i, _i, _p, _p, _p],
    "sn_deepsets_tail_f32": [_p, _i, _p, _p, _p],
    "sn_masked_linear_blockbias_f32": [_p, _i, _l, _i, _p, _i, _p, _p, _l, _i, _i, _p, _p, _p, _i, _p],
    "sn_bn_fold_f32": [_p, _p, _p, _p, _f, _i, _i, _p, _p, _p],
    "sn_bn_running_update_f32": [_p, _p, _p, _f, _i, _p, _p, _p],
    "sn_colstats_blocks": [_l],
    "sn_phi_fused_f32": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _p, _p],
    "sn_rho_fused_f32": [_p, _p, _p, _p, _l, _l, _p, _i, _i, _p, _p],
    "sn_gnn_fused_f32": [_p, _p, _i, _p, _i, _p, _p, _l, _p, _p, _p, _p, _p, _p, _i, _p, _p],
    "sn_gin_net_fused_f32": [_p, _p, _i, _p, _p, _i, _i, _p, _l, _p, _p, _p, _p, _p, _p, _i, _p],
    "sn_transformer_net_fused_f32": [_p, _p, _i, _p, _p, _i, _i, _p, _i, _p, _l, _p, _p, _p, _p, _p, _p, _i, _p],
    "sn_masked_colstats_f32": [_p, _i, _l, _i, _p, _i, _p, _p, _p, _p, _p],
    "sn_masked_affine_f32": [_p, _i, _l, _i, _p, _i, _i, _p, _p, _p, _i, _p, _i, _p],
    "sn_masked_layernorm_f32": [_p, _p, _l, _i, _p, _p, _f, _p, _i, _p, _p],
    "sn_set_attention_f32": [_p, _p, _p, _l, _i, _i, _i, _p, _p, _p, _p],
    "sn_slot_sum_f32": [_p, _l, _i, _i, _p, _p],
    "sn_plan_double_i32": [_p, _p, _l, _l, _p, _p, _p],
    "sn_keep_mask_f32": [_p, _l, _f, _f, _p],
    "sn_embedding_sum_f32": [_p, _i, _i, _l, C.POINTER(_p), C.POINTER(C.c_int64), _i, _p, _p, _p],
    "sn_segment_pool_f32": [_p, _l, _i, _p, _i, _p, _p],
    "sn_ign_contract_2to1_f32": [_p, _l, _i, _p, _p, _p],
    "sn_laplacian_evd_f32": [_p, _l, _p, _l, _l, _i, _p, _p, _p, _l, _p, _i, _i, _p, _p, _p],
    "sn_linear_bn_scratch_floats": [_l, _i, _i],
    "sn_linear_bn_train_f32": [_p, _i, _l, _i, _p, _i, _p, _p, _i, _p, _i, _p, _p, _f, _f, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    "sn_bn_train_stats_f32": [_p, _i, _l, _i, _p, _i, _p, _p, _f, _f, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    "sn_linear_wgrad_f32": [_p, _i, _p, _i, _l, _i, _i, _p, _i, _p, _p, _p, _p],
    "sn_bn_act_bwd_f32": [_p, _i, _p, _i, _l, _i, _p, _i, _p, _p, _p, _p, _i, _p, _p, _p, _i, _p, _p],
    "sn_relu_bwd_f32": [_p, _p, _l, _i, _p, _i, _p, _p],
    "sn_masked_layernorm_bwd_f32": [_p, _p, _p, _l, _i, _p, _f, _p, _i, _p, _p, _p, _p, _p],
    "sn_set_attention_bwd_f32": [_p, _p, _p, _p, _l, _i, _i, _i, _p, _p, _p, _p, _p, _p],
    "sn_gine_aggregate_bwd_f32": [_p, _p, _p, _l, _i, _p, _p, _p, _p, _p, _p, _p],
    "sn_slot_broadcast_f32": [_p, _l, _i, _i, _p, _p, _p],
    "sn_segment_broadcast_f32": [_p, _l, _i, _p, _i, _p, _p],
    "sn_embedding_sum_bwd_f32": [_p, _i, _i, _l, _p, C.POINTER(C.c_int64), _i, _p, _p, _p, _p],
    "sn_embedding_sum_layers_f32": [_p, _i, _i, _l, _i, _p, C.POINTER(C.c_int64), _i, _p, _p, _p],
    "sn_embedding_sum_bwd_layers_f32": [_p, _i, _i, _l, _i, _p, C.POINTER(C.c_int64), _i, _p, _p, _p, _p],
    "sn_dot_f32": [_p, _p, _l, _p, _p, _p],
    "sn_pna_aggregate_f32": [_p, _i, _p, _i, _i, _l, _p, _p, _f, _p, _i, _p],
    "sn_pna_aggregate_gather_f32": [_p, _i, _p, _i, _p, _i, _p, _i, _i, _l, _p, _p, _p, _f, _p, _i, _i, _p],
    "sn_grouped_linear_f32": [_p, _i, _l, _i, _i, _i, _p, _p, _p, _p, _p, _p, _i, _p],
    "sn_edge_attention_f32": [_p, _p, _p, _p, _l, _i, _i, _p, _p, _p, _p, _p],
    "sn_edge_attention_strided_f32": [_p, _p, _p, _i, _p, _i, _l, _i, _i, _p, _p, _p, _p, _p],
    "sn_pointwise_f32": [_p, _i, _l, _i, _p, _p, _p, _i, _f, _p, _i, _p, _i, _p],
    "sn_deepsigns_phi_f32": [_p, _p, _i, _p, _p, _p, _p, _i, _p, _p],
    "sn_mlp_chain_f32": [_p, _i, _l, _i, _p, _i, _p, _i, _i, _p, _i, _i, _p],
    "sn_gat_aggregate_f32": [_p, _p, _p, _p, _l, _i, _i, _f, _i, _p, _p, _p, _p, _p],
    "sn_gat_aggregate_bwd_f32": [_p, _p, _p, _p, _p, _p, _p, _l, _l, _i, _i, _f, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    "sn_edge_rows_sum_f32": [_p, _i, _i, _l, _p, _p, _p, _i, _p],
    "sn_pna_aggregate_bwd_f32": [_p, _i, _i, _l, _p, _p, _f, _p, _i, _p, _p, _p],
    "sn_edge_attention_bwd_f32": [_p, _p, _p, _p, _p, _p, _l, _l, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    "sn_act_bwd_f32": [_p, _p, _l, _i, _p, _i, _f, _p, _p],
    "sn_gatedgcn_fused_f32": [_p, _p, _p, _p, _l, _p, _p, _p, _p, _p, _p],
    "sn_dense_attention_f32": [_p, _p, _p, _l, _i, _i, _i, _p, _p, _p],
    "sn_dense_attention_bwd_f32": [_p, _p, _p, _p, _p, _p, _l, _i, _i, _i, _p, _p, _p, _p, _p],
    "sn_eigenspace_group": [_p, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p],
    "sn_eigenspace_projectors_f32": [_p, _i, _i, _p, _p, _i, _p, _p],
    "sn_ign_contract_eigvecs_f32": [_p, _i, _i, _p, _p, _i, _i, _p, _p],
    "sn_gated_aggregate_f32": [_p, _p, _p, _p, _i, _p, _l, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    "sn_gated_aggregate_bwd_f32": [_p, _p, _p, _p, _p, _p, _p, _l, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    "sn_adam_step_f32": [_p, _p, _p, _p, _l, _f, _f, _f, _f, _f, _i, _f, _p],
    "sn_train_post_link_f32": [_p, _p],
    "sn_train_linear_blocks": [_l, _i],                  # (returns a count, not a status)
    "sn_train_linear_bwd_blocks": [_l, _i],
    "sn_train_bn_bwd_blocks": [_l, _i],
    "sn_train_linear_f32": [_p, _p],
    "sn_train_bn_finish_f32": [_p, _i, _i, _i, _p, _p, _f, _f, _p, _p, _p, _p, _p],
    "sn_train_linear_bwd_f32": [_p, _p],
    "sn_train_bn_bwd_sums_f32": [_p, _i, _p, _i, _l, _i, _i, _p, _i, _p, _i, _p, _p],
    "sn_train_bn_bwd_finish_f32": [_p, _i, _i, _i, _p, _p, _p, _p, _p, _p, _i, _p],
    "sn_train_bn_apply_f32": [_p, _i, _l, _i, _i, _p, _i, _p, _i, _p, _i, _p, _i, _p],
    "sn_train_reduce_parts_f32": [_p, _i, _l, _l, _p, _i, _p],
    "sn_train_dot_finish_f64": [_p, _i, _p, _i, _p],
    "sn_train_reduce_jobs_f32": [_p, _i, _p],
    "sn_train_dot_jobs_f64": [_p, _i, _p],
    "sn_train_bn_bwd_f32": [_p, _i, _p, _i, _l, _i, _i, _p, _i, _p, _p, _i, _p, _p, _p, _p, _p, _i, _p],
    "sn_masked_layernorm_bwd_acc_f32": [_p, _p, _p, _l, _i, _p, _f, _p, _i, _p, _p, _p, _p, _p],
    "sn_gin_aggregate_add_f32": [_p, _p, _p, _l, _i, _p, _p, _p, _p],
    "sn_gine_aggregate_bwd_add_f32": [_p, _p, _p, _p, _l, _i, _p, _p, _p, _p, _p, _p, _p],
    "sn_train_scalar_mlp_stats_f32": [_p, _f, _p, _p, _f, _p, _p, _p, _p],
    "sn_train_scalar_mlp_apply_f32": [_p, _p, _p],
    "sn_train_scalar_mlp_bwd_f32": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p, _p],
}
_SPECIAL_RESTYPE = {"sn_last_error": C.c_char_p, "sn_packed_weight_floats": C.c_int64}

_lib = None


def lib():
    """Load (once) and return the ctypes library.  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m signnet_basisnet_amd.build` "
                "(hipcc --offload-arch=gfx950).  There is no CPU/PyTorch fallback.")
        L = C.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = C.c_int
        L.sn_last_error.argtypes = []
        L.sn_last_error.restype = C.c_char_p
        L.sn_packed_weight_floats.argtypes = [_i, _i]
        L.sn_packed_weight_floats.restype = C.c_int64
        L.sn_split_packed_bytes.argtypes = [_i, _i]
        L.sn_split_packed_bytes.restype = C.c_int64
        L.sn_phi_bins_bound.argtypes = [_l, _i]
        L.sn_phi_bins_bound.restype = C.c_int64
        L.sn_linear_wgrad_scratch_floats.argtypes = [_l, _i, _i]
        L.sn_linear_wgrad_scratch_floats.restype = C.c_int64
        L.sn_bn_act_bwd_scratch_floats.argtypes = [_l, _i]
        L.sn_bn_act_bwd_scratch_floats.restype = C.c_int64
        L.sn_linear_bn_scratch_floats.restype = C.c_int64
        L.sn_layernorm_bwd_scratch_floats.argtypes = [_l, _i]
        L.sn_layernorm_bwd_scratch_floats.restype = C.c_int64
        L.sn_embedding_bwd_scratch_floats.argtypes = [_l, _i, C.POINTER(C.c_int64), _i]
        L.sn_embedding_bwd_scratch_floats.restype = C.c_int64
        L.sn_ign_mlp_supported.argtypes = [_i, _i, _i]
        L.sn_ign_mlp_supported.restype = C.c_int
        L.sn_embedding_bwd_layers_scratch_floats.argtypes = [_l, _i, _i]
        L.sn_embedding_bwd_layers_scratch_floats.restype = C.c_int64
        L.sn_gatedgcn_max_edges.argtypes = [_i]
        L.sn_gatedgcn_max_edges.restype = C.c_int
        L.sn_evd_work_ints.argtypes = [_l]
        L.sn_evd_work_ints.restype = C.c_int64
        L.sn_ign_contract_scratch_floats.argtypes = [_l, _i]
        L.sn_ign_contract_scratch_floats.restype = C.c_int64
        L.sn_train_linear_bwd_part_floats.argtypes = [_l, _i, _i, _i]
        L.sn_train_linear_bwd_part_floats.restype = C.c_int64
        L.sn_train_scalar_mlp_work_doubles.argtypes = [_l, _i, _i]
        L.sn_train_scalar_mlp_work_doubles.restype = C.c_int64
        if L.sn_version() != ABI_VERSION:
            raise RuntimeError(f"libsignnet_hip.so ABI version {L.sn_version()} != {ABI_VERSION} (include/signnet_hip.h: SN_ABI_VERSION): "
                               "rebuild with `python -m signnet_basisnet_amd.build --force`")
        _lib = L
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().sn_last_error()
        raise RuntimeError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


_EMPTY_BUF = {}


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  An EMPTY device tensor (zero rows: a batch without edges, an empty shard) has a NULL
    data pointer too; the entry points take NULL for "argument absent" and reject it for mandatory ones, so an empty tensor is passed
    as a valid pointer to a small per-device buffer — with zero rows nothing is read or written through it."""
    if t is None:
        return None
    p = t.data_ptr()
    if p == 0 and t.is_cuda:
        buf = _EMPTY_BUF.get(t.device)
        if buf is None:
            buf = _EMPTY_BUF[t.device] = torch.zeros(64, dtype=torch.int64, device=t.device)
        return buf.data_ptr()
    return p


_stream_handle = None      # set by stream_scope: one torch.cuda.current_stream() lookup (~8 us) per forward instead of per launch


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """The HIP stream handle the next launch goes to: torch's CURRENT stream of the current device.  (torch.cuda.current_stream() builds
    a Stream object through several Python layers: ~8 us per call, ~180 calls in an eager training step — the raw-handle query is
    0.3 us.  The autograd engine runs the adjoints on its own thread with the forward's stream made current, so the query must stay
    per launch there.)"""
    if _stream_handle is not None:
        return _stream_handle
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


class stream_scope:
    """Pin the launch stream for a sequence of launches issued from one Python call (the caller does not switch
    torch streams in between)."""

    def __enter__(self):
        global _stream_handle
        self.prev = _stream_handle
        # (no GPU: leave it unset — the first op's require_cuda raises the proper 'GPU only' error)
        if not torch.cuda.is_available():
            _stream_handle = None
        elif _raw_stream is not None:
            _stream_handle = _raw_stream(torch.cuda.current_device())
        else:
            _stream_handle = torch.cuda.current_stream().cuda_stream
        return self

    def __exit__(self, *a):
        global _stream_handle
        _stream_handle = self.prev
        return False


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("signnet_basisnet_amd ops run on the GPU only (tensor is on %s); "
                               "there is no CPU fallback" % t.device)
