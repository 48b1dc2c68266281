"""Thin Python wrappers over the C ABI (one per entry point of include/signnet_hip.h).

Each wrapper validates dtype/contiguity, allocates the output with torch (device memory is
PyTorch's job here), and launches on torch's current stream.  No arithmetic happens in Python.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import torch

from . import _lib
from ._lib import (EPI_AFFINE, EPI_BIAS, EPI_LEAKY, EPI_RELU, EPI_RELU_PRE, EPI_RESIDUAL, EPI_RESIDUAL_PRE, check, lib, ptr,
                   require_cuda, stream)

__all__ = ["KernelTimer", "KERNEL_ROOFLINE", "GraphPlan", "PlanBins", "build_plan", "pack_eig", "pack_weight", "gin_aggregate", "gine_aggregate",
           "masked_linear", "masked_colstats", "masked_affine", "masked_layernorm", "set_attention",
           "slot_sum", "embedding_sum", "segment_pool", "PackedLinear",
           "EPI_BIAS", "EPI_RELU_PRE", "EPI_AFFINE", "EPI_RELU", "EPI_RESIDUAL"]


# ----------------------------------------------------------------------------- kernel timing (bench.py)
class _NoSpan:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


_NOSPAN = _NoSpan()
_active_timer = None


class _Span:
    __slots__ = ("timer", "name", "e0")

    def __init__(self, timer, name):
        self.timer, self.name = timer, name

    def __enter__(self):
        self.e0 = torch.cuda.Event(enable_timing=True)
        self.e0.record()          # torch's current stream == the stream the kernel is launched on
        return self

    def __exit__(self, *a):
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self.timer.spans.append((self.name, self.e0, e1))
        return False


class KernelTimer:
    """HIP-event timing of C-ABI launches on torch's current stream (the stream every launch uses).
    `only` restricts recording to the named entry points and `stride` to every stride-th launch of each, so the timed
    region is barely perturbed (an event pair costs the stream ~4-5 us of marker packets)."""

    def __init__(self, only=None, stride=1):
        self.only = set(only) if only else None
        self.stride = max(1, int(stride))
        self.seen = {}
        self.spans = []

    def __enter__(self):
        global _active_timer
        _active_timer = self
        return self

    def __exit__(self, *a):
        global _active_timer
        _active_timer = None
        return False

    def summary(self):
        torch.cuda.synchronize()
        acc = {}
        for name, e0, e1 in self.spans:
            n, t = acc.get(name, (0, 0.0))
            acc[name] = (n + 1, t + e0.elapsed_time(e1))
        return {k: (n, t / n) for k, (n, t) in acc.items()}


def _span(name):
    t = _active_timer
    if t is None or (t.only is not None and name not in t.only):
        return _NOSPAN
    if t.stride > 1:
        k = t.seen.get(name, 0)
        t.seen[name] = k + 1
        if k % t.stride:
            return _NOSPAN
    return _Span(t, name)


def _f32c(t, name):
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise ValueError(f"{name}: expected a contiguous float32 tensor, got {t.dtype} contiguous={t.is_contiguous()}")
    return t


@dataclass
class PlanBins:
    """Work bins of the fused phi / rho stages (device arrays written by sn_batch_plan; see signnet_hip.h)."""
    phi_bin_col: torch.Tensor    # int32 [phi_max_bins]
    phi_max_bins: int
    phi_col_bin0: torch.Tensor   # int32 [B+1]
    phi_col_mem: torch.Tensor    # int32 [B*8]
    phi_col_off: torch.Tensor    # int32 [B*8]
    rho_bin0: torch.Tensor       # int32 [B+1]
    meta: torch.Tensor           # int32 [8]
    cstruct: object              # ctypes mirror (sn_plan_bins) holding the device pointers
    phi_bin_mem: torch.Tensor = None   # int32 [phi_max_bins * 16] member records of every bin (meta[7] bins)


@dataclass
class GraphPlan:
    """Device-resident structure of one batch (built once per batch by `build_plan`)."""
    N: int
    B: int
    E: int
    kmax: int
    graph_ptr: torch.Tensor   # int32 [B+1]
    node_graph: torch.Tensor  # int32 [N]
    nvalid: torch.Tensor      # int32 [N]   valid eigenvector slots per node
    evoff: torch.Tensor       # int64 [B+1] offset of each graph's n_b x n_b eigenvector block
    rowptr: torch.Tensor      # int32 [N+1] dst-sorted CSR
    col: torch.Tensor         # int32 [E]   source node of each in-edge
    eperm: torch.Tensor       # int32 [E]   original edge id of each CSR slot
    status: torch.Tensor      # int32 [8]   [err bits, max nodes/graph, max in-degree, fused-stage flags, gnn completion counter,
                              #              embedding index out of range (layer path), -, -]
    bins: PlanBins | None
    flags: torch.Tensor = None   # status (+ bins meta) as one contiguous block

    def check(self):
        """Synchronising validity check (raises on malformed batches)."""
        st = self.status.tolist()
        if st[0]:
            bits = [n for b, n in ((1, "batch not sorted"), (2, "graph id out of range"),
                                   (4, "edge endpoint out of range"), (8, "edge crosses graphs")) if st[0] & b]
            raise ValueError("malformed graph batch: " + ", ".join(bits))
        if st[5] or (st[3] & 4):
            raise IndexError(EMBEDDING_INDEX_ERROR)
        return st


class _PlanBinsC(C.Structure):
    _fields_ = [("phi_bin_col", C.c_void_p), ("phi_max_bins", C.c_int64), ("phi_col_bin0", C.c_void_p),
                ("phi_col_mem", C.c_void_p), ("phi_col_off", C.c_void_p), ("rho_bin0", C.c_void_p), ("meta", C.c_void_p),
                ("node_graph", C.c_void_p), ("phi_bin_mem", C.c_void_p)]


class _PlanEarlyC(C.Structure):
    _fields_ = [("node_ids", C.c_void_p), ("n_node_ids", C.c_int64), ("node_vocab", C.c_int64),
                ("edge_ids", C.c_void_p), ("n_edge_ids", C.c_int64), ("edge_vocab", C.c_int64),
                ("max_graph_edges", C.c_int), ("reserved", C.c_int), ("host", C.c_void_p)]


class EarlyReport:
    """The batch's flags as sn_batch_plan_ex reports them to pinned host memory while the rest of the forward is still running
    (include/signnet_hip.h: sn_plan_early).  `wait()` polls the four done words (the plan is the first ~20 us of a forward: they
    are normally set by the time the host has queued the stage kernels) and returns the flag words as a list."""
    ERR, NMAX, DEGMAX, EDGES, PHI, RHO, IDS, DONE = 0, 1, 2, 3, 4, 5, 6, 8
    _pool = []

    def __init__(self):
        if EarlyReport._pool:
            self.t, self.v = EarlyReport._pool.pop()
        else:
            self.t = torch.zeros(16, dtype=torch.int32, pin_memory=True)
            self.v = self.t.numpy()
        self.v[:] = 0
        self.cstruct = None
        self._keep = None

    def arm(self, node_ids=None, node_vocab=0, edge_ids=None, edge_vocab=0, max_graph_edges=0):
        self._keep = (node_ids, edge_ids)
        self.cstruct = _PlanEarlyC(ptr(node_ids), 0 if node_ids is None else node_ids.numel(), int(node_vocab),
                                   ptr(edge_ids), 0 if edge_ids is None else edge_ids.numel(), int(edge_vocab),
                                   int(max_graph_edges), 0, self.t.data_ptr())
        return self

    def wait(self, timeout_s=5.0):
        v = self.v
        if not (v[8] and v[9] and v[10] and v[11]):
            import time
            t_end = time.perf_counter() + timeout_s
            while not (v[8] and v[9] and v[10] and v[11]):
                if time.perf_counter() > t_end:
                    torch.cuda.synchronize()
                    if not (v[8] and v[9] and v[10] and v[11]):
                        raise RuntimeError("sn_batch_plan_ex did not report its flags")
        return v[:8].tolist()

    def wait_nmax(self):
        """Only the largest graph (the rho-bins workgroup's word): what the all-eigenvector mode sizes its tensors from."""
        v = self.v
        if not v[10]:
            import time
            t_end = time.perf_counter() + 5.0
            while not v[10]:
                if time.perf_counter() > t_end:
                    torch.cuda.synchronize()
                    if not v[10]:
                        raise RuntimeError("sn_batch_plan_ex did not report the largest graph")
        return int(v[1])

    def release(self):
        """Back to the pool (only once the launch that writes it has completed: after wait())."""
        self._keep = self.cstruct = None
        EarlyReport._pool.append((self.t, self.v))


def early_supported(N: int, E: int, B: int) -> bool:
    return bool(lib().sn_batch_plan_early_supported(int(N), int(E), int(B)))


def build_plan(batch: torch.Tensor, edge_index: torch.Tensor, num_graphs: int, kmax: int = 0, bins: bool = False,
               early: "EarlyReport | None" = None, columns: bool = False) -> GraphPlan:
    """bins=True also lays out the work bins of the fused stages (same launch, no host sync).  early: an armed EarlyReport — the
    batch's flags are also written to its pinned buffer by the plan kernel itself (one-launch plans only: early_supported()).
    columns=True: the planner's column arrays (phi_bin_col, phi_col_bin0, phi_col_mem, phi_col_off) are written too — the stage kernels
    walk the per-bin member records (phi_bin_mem) only, so the forward leaves them out."""
    require_cuda(batch, edge_index)
    if batch.dtype != torch.int64 or edge_index.dtype != torch.int64:
        raise ValueError("build_plan: batch and edge_index must be int64 (the reference's index dtype)")
    batch = batch.contiguous()
    edge_index = edge_index.contiguous()
    N, E, B = batch.numel(), edge_index.shape[1] if edge_index.numel() else 0, int(num_graphs)
    dev = batch.device
    # one int32 arena, carved into the plan arrays (single allocation per batch)
    sizes = [B + 1, N, N, N + 1, E, E, N + 8, 8]          # status last: [status(8) | bins meta(8)] is one 16-int block
    mb = 0
    if bins:
        mb = int(lib().sn_phi_bins_bound(B, int(kmax)))
        sizes += [8, mb if columns else 0, (B + 1) if columns else 0, (8 * B) if columns else 0, (8 * B) if columns else 0, B + 1, 16 * mb]
    offs = [0]
    for s in sizes:
        offs.append(offs[-1] + ((s + 3) // 4) * 4)
    arena = torch.empty(offs[-1], dtype=torch.int32, device=dev)
    parts = [arena[offs[i]:offs[i] + sizes[i]] for i in range(len(sizes))]
    graph_ptr, node_graph, nvalid, rowptr, col, eperm, scratch, status = parts[:8]
    evoff = torch.empty(B + 1, dtype=torch.int64, device=dev)
    pb = None
    if bins:
        meta, bc, cb0, mem, off, rb0, bmem = parts[8:15]
        if columns:
            cs = _PlanBinsC(bc.data_ptr(), mb, cb0.data_ptr(), mem.data_ptr(), off.data_ptr(), rb0.data_ptr(), meta.data_ptr(),
                            node_graph.data_ptr(), bmem.data_ptr())
        else:
            bc = cb0 = mem = off = None
            cs = _PlanBinsC(None, mb, None, None, None, rb0.data_ptr(), meta.data_ptr(), node_graph.data_ptr(), bmem.data_ptr())
        pb = PlanBins(bc, mb, cb0, mem, off, rb0, meta, cs, bmem)
    with _span("sn_batch_plan"):
        check(lib().sn_batch_plan_ex(ptr(batch), N, B, ptr(edge_index), E, int(kmax), ptr(graph_ptr), ptr(node_graph),
                                     ptr(nvalid), ptr(evoff), ptr(rowptr), ptr(col), ptr(eperm), ptr(status),
                                     C.byref(pb.cstruct) if pb is not None else None, ptr(scratch),
                                     C.byref(early.cstruct) if early is not None else None, stream()), "sn_batch_plan")
    plan = GraphPlan(N, B, E, int(kmax), graph_ptr, node_graph, nvalid, evoff, rowptr, col, eperm, status, pb)
    plan.flags = arena[offs[7]:offs[7] + 16] if bins else status      # [status(8) | meta(8)] contiguous
    return plan


def pack_eig(plan: GraphPlan, eigen_vectors, eigen_values, K: int, want_values: bool):
    require_cuda(eigen_vectors)
    ev = _f32c(eigen_vectors, "eigen_vectors")
    x0 = torch.empty(plan.N, K, dtype=torch.float32, device=ev.device)
    s0 = torch.empty_like(x0) if want_values else None
    es = _f32c(eigen_values, "eigen_values") if want_values else None
    with _span("sn_pack_eig_f32"):
        check(lib().sn_pack_eig_f32(ptr(ev), ptr(es), ptr(plan.graph_ptr), ptr(plan.node_graph), ptr(plan.nvalid),
                                    ptr(plan.evoff), plan.N, K, ptr(x0), ptr(s0), stream()), "sn_pack_eig_f32")
    return x0, s0


PHI_BIN_ROWS = 64     # SN_PHI_BIN_ROWS


def bn_fold(bn, c_pad=None):
    """Eval-mode nn.BatchNorm1d -> (scale, shift), zero padded to c_pad channels."""
    Cc = bn.num_features
    cp = Cc if c_pad is None else int(c_pad)
    dev = bn.running_mean.device
    require_cuda(bn.running_mean)
    scale = torch.empty(cp, dtype=torch.float32, device=dev)
    shift = torch.empty(cp, dtype=torch.float32, device=dev)
    w = bn.weight.detach() if bn.affine else None
    b = bn.bias.detach() if bn.affine else None
    check(lib().sn_bn_fold_f32(ptr(w), ptr(b), ptr(bn.running_mean), ptr(bn.running_var), float(bn.eps), Cc, cp,
                               ptr(scale), ptr(shift), stream()), "sn_bn_fold_f32")
    return scale, shift


def pad_vec(v, c_pad):
    """Zero-pad a per-channel vector to c_pad floats (a copy, no arithmetic)."""
    v = v.detach().reshape(-1)
    out = torch.zeros(c_pad, dtype=torch.float32, device=v.device)
    out[:v.numel()].copy_(v)
    return out


def packed_floats(d_out: int, d_in: int) -> int:
    return int(lib().sn_packed_weight_floats(d_out, d_in))


def pack_weight(W: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """nn.Linear weight [d_out, d_in] -> MFMA fragment order (see csrc/common.hpp)."""
    require_cuda(W)
    if W.dtype != torch.float32 or W.dim() != 2 or W.stride(1) != 1:
        raise ValueError("pack_weight: expected a float32 [d_out, d_in] matrix with unit inner stride")
    d_out, d_in = W.shape
    n = packed_floats(d_out, d_in)
    if out is None:
        out = torch.empty(n, dtype=torch.float32, device=W.device)
    check(lib().sn_pack_weight_f32(ptr(W), d_out, d_in, W.stride(0), ptr(out), stream()), "sn_pack_weight_f32")
    return out


def pack_weight_t(W: torch.Tensor) -> torch.Tensor:
    """Packed W^T of a [rows, cols] matrix, read in place (no transposed copy): the weight of dX = dY @ W."""
    require_cuda(W)
    if W.dtype != torch.float32 or W.dim() != 2 or W.stride(1) != 1:
        raise ValueError("pack_weight_t: expected a float32 matrix with unit inner stride")
    rows, cols = W.shape
    out = torch.empty(packed_floats(cols, rows), dtype=torch.float32, device=W.device)
    check(lib().sn_pack_weight_t_f32(ptr(W), rows, cols, W.stride(0), ptr(out), stream()), "sn_pack_weight_t_f32")
    return out


def pack_split(W: torch.Tensor, e0=None, e1=None, e2=None) -> torch.Tensor:
    """nn.Linear weight [d_out, d_in] (+ up to three per-output-channel epilogue vectors) -> the split-packed buffer
    of sn_pack_split_f32: exact 3 x bf16 significand split of every weight, fragment order of the fused phi / rho
    kernels (csrc/fused_common.hpp).  Returns a uint8 tensor that owns the buffer."""
    require_cuda(W)
    if W.dtype != torch.float32 or W.dim() != 2 or W.stride(1) != 1:
        raise ValueError("pack_split: expected a float32 [d_out, d_in] matrix with unit inner stride")
    d_out, d_in = W.shape
    vecs = []
    for e in (e0, e1, e2):
        if e is not None:
            e = e.detach().to(device=W.device, dtype=torch.float32).contiguous()
            if e.numel() < d_out:
                raise ValueError("pack_split: epilogue vector shorter than d_out")
        vecs.append(e)
    out = torch.empty(int(lib().sn_split_packed_bytes(d_out, d_in)), dtype=torch.uint8, device=W.device)
    check(lib().sn_pack_split_f32(ptr(W), d_out, d_in, W.stride(0), ptr(vecs[0]), ptr(vecs[1]), ptr(vecs[2]), ptr(out),
                                  stream()), "sn_pack_split_f32")
    return out


@dataclass
class PackedLinear:
    wp: torch.Tensor
    d_out: int
    d_in: int
    bias: torch.Tensor | None = None


def gin_aggregate(x, plan: GraphPlan, eps=None, negate=False, slab=False):
    """x [N, ...] -> (1+eps) x_i + sum_{j->i} x_j over the node axis (trailing dims flattened)."""
    require_cuda(x)
    x = _f32c(x, "x")
    N = x.shape[0]
    F = x.numel() // max(N, 1)
    out = torch.empty_like(x)
    if slab:
        with _span("sn_gin_aggregate_slab_f32"):
            check(lib().sn_gin_aggregate_slab_f32(ptr(x), ptr(out), N, F, plan.B, ptr(plan.graph_ptr), ptr(plan.rowptr),
                                                  ptr(plan.col), ptr(eps), int(negate), stream()),
                  "sn_gin_aggregate_slab_f32")
    else:
        with _span("sn_gin_aggregate_f32"):
            check(lib().sn_gin_aggregate_f32(ptr(x), ptr(out), N, F, ptr(plan.rowptr), ptr(plan.col), ptr(eps),
                                             int(negate), stream()), "sn_gin_aggregate_f32")
    return out


def doubled_plan(plan: GraphPlan):
    """The CSR of two disjoint copies of the batch (nodes N..2N-1 = the second copy): the phi(+x) / phi(-x) passes stacked group-major
    aggregate in ONE launch over [2N, K*d].  Index plumbing only (one launch: sn_plan_double_i32), kept on the plan."""
    d = getattr(plan, "_doubled", None)
    if d is None:
        import types
        E = plan.col.numel()
        rowptr2 = torch.empty(2 * plan.N + 1, dtype=torch.int32, device=plan.rowptr.device)
        col2 = torch.empty(2 * E, dtype=torch.int32, device=plan.rowptr.device)
        check(lib().sn_plan_double_i32(ptr(plan.rowptr), ptr(plan.col), plan.N, E, ptr(rowptr2), ptr(col2), stream()), "sn_plan_double_i32")
        d = types.SimpleNamespace(N=2 * plan.N, B=2 * plan.B, E=2 * E, rowptr=rowptr2, col=col2)
        plan._doubled = d
    return d


def gine_aggregate(x, ea, plan: GraphPlan, eps=None):
    require_cuda(x, ea)
    x, ea = _f32c(x, "x"), _f32c(ea, "edge_attr")
    if ea.shape != (plan.E, x.shape[1]):
        raise ValueError("gine_aggregate: edge features must be [E, C] with C == x.shape[1]")
    out = torch.empty_like(x)
    with _span("sn_gine_aggregate_f32"):
        check(lib().sn_gine_aggregate_f32(ptr(x), ptr(ea), ptr(out), x.shape[0], x.shape[1], ptr(plan.rowptr),
                                          ptr(plan.col), ptr(plan.eperm), ptr(eps), stream()), "sn_gine_aggregate_f32")
    return out


def gated_aggregate(Ah, Bh, Dh, Eh, Ce, plan: GraphPlan, want_den=False, epilogue=None):
    """GatedGCN message passing (gatedgcn_layer.py:51-56): -> (h [N,C], e [E,C][, den [N,C]]).
    Ah/Bh/Dh/Eh may be column blocks (views) of one [N, 4C] tensor.  epilogue = (h_scale, h_shift, e_scale, e_shift, h_res, e_res):
    eval-mode BatchNorm + ReLU + residual fused into the same pass (residuals may be None)."""
    require_cuda(Ah, Bh, Dh, Eh, Ce)
    N, Cc = Ah.shape
    ldn = Ah.stride(0)
    for t, n in ((Ah, "Ah"), (Bh, "Bh"), (Dh, "Dh"), (Eh, "Eh")):
        if t.dtype != torch.float32 or t.shape != (N, Cc) or t.stride(1) != 1 or t.stride(0) != ldn:
            raise ValueError(f"gated_aggregate: {n} must be a float32 [N, C] row block with the common row stride")
    Ce = _f32c(Ce, "Ce")
    if Ce.shape != (plan.E, Cc):
        raise ValueError("gated_aggregate: Ce must be [E, C]")
    h = torch.empty(N, Cc, dtype=torch.float32, device=Ah.device)
    e = torch.empty_like(Ce)
    den = torch.empty_like(h) if want_den else None
    ep = epilogue if epilogue is not None else (None,) * 6
    with _span("sn_gated_aggregate_f32"):
        check(lib().sn_gated_aggregate_f32(ptr(Ah), ptr(Bh), ptr(Dh), ptr(Eh), ldn, ptr(Ce), N, Cc, ptr(plan.rowptr), ptr(plan.col),
                                           ptr(plan.eperm), ptr(h), ptr(e), ptr(den), *(ptr(t) for t in ep), stream()),
              "sn_gated_aggregate_f32")
    return (h, e, den) if want_den else (h, e)


def masked_linear(x, pl: PackedLinear, nvalid=None, K=0, *, scale=None, shift=None, relu_pre=False, relu=False,
                  residual=None, use_bias=True, out=None, residual_pre=False, leaky=False):
    """y = epilogue(x @ W^T); x is a row matrix [..., d_in] (leading dims flattened to rows).
    residual_pre: the residual is added right behind the bias, in front of relu_pre / the affine / relu (BatchNorm(x + Linear(h)))."""
    require_cuda(x)
    x = _f32c(x, "x")
    if x.shape[-1] != pl.d_in:
        raise ValueError(f"masked_linear: x has {x.shape[-1]} channels, weight expects {pl.d_in}")
    R = x.numel() // pl.d_in
    flags = 0
    bias = pl.bias if use_bias else None
    if bias is not None:
        flags |= EPI_BIAS
    if relu_pre:
        flags |= EPI_RELU_PRE
    if scale is not None:
        flags |= EPI_AFFINE
    if relu:
        flags |= EPI_RELU
    if leaky:                   # LeakyReLU(0.01) in RELU's place (then + residual)
        flags |= EPI_LEAKY
    if residual is not None:
        flags |= EPI_RESIDUAL_PRE if residual_pre else EPI_RESIDUAL
        residual = _f32c(residual, "residual")
    if out is None:
        out = torch.empty(*x.shape[:-1], pl.d_out, dtype=torch.float32, device=x.device)
    if R == 0:
        return out
    with _span("sn_masked_linear_f32"):
        check(lib().sn_masked_linear_f32(ptr(x), pl.d_in, R, pl.d_in, ptr(pl.wp), pl.d_out, ptr(bias), ptr(nvalid),
                                         int(K), flags, ptr(scale), ptr(shift), ptr(residual), pl.d_out, ptr(out),
                                         pl.d_out, stream()), "sn_masked_linear_f32")
    return out


def linear_block_bias(x, pl: PackedLinear, block_bias, rows_per_block, *, scale=None, shift=None, relu_pre=False, relu=False):
    """y = epilogue(x @ W^T + b + block_bias[row // rows_per_block]): a Linear whose bias differs per block of consecutive rows (the
    equivariant 1->1 layers: Linear over cat[x, mean of the row's matrix] without the concatenation)."""
    require_cuda(x)
    x, block_bias = _f32c(x, "x"), _f32c(block_bias, "block_bias")
    R = x.numel() // pl.d_in
    if block_bias.shape[-1] != pl.d_out or block_bias.numel() // pl.d_out * int(rows_per_block) < R:
        raise ValueError("linear_block_bias: one bias row per block of rows expected")
    flags = (EPI_BIAS if pl.bias is not None else 0) | (EPI_RELU_PRE if relu_pre else 0) | (EPI_AFFINE if scale is not None else 0) | \
            (EPI_RELU if relu else 0)
    out = torch.empty(*x.shape[:-1], pl.d_out, dtype=torch.float32, device=x.device)
    with _span("sn_masked_linear_f32"):
        check(lib().sn_masked_linear_blockbias_f32(ptr(x), pl.d_in, R, pl.d_in, ptr(pl.wp), pl.d_out, ptr(pl.bias), ptr(block_bias),
                                                   int(rows_per_block), pl.d_out, flags, ptr(scale), ptr(shift), ptr(out), pl.d_out, stream()),
              "sn_masked_linear_blockbias_f32")
    return out


def masked_colstats(x, nvalid=None, K=0):
    """Per-channel mean / biased variance over the valid rows of a row matrix [..., C]."""
    require_cuda(x)
    x = _f32c(x, "x")
    Cc = x.shape[-1]
    R = x.numel() // Cc
    nb = int(lib().sn_colstats_blocks(R))
    mean = torch.empty(Cc, dtype=torch.float32, device=x.device)
    var = torch.empty_like(mean)
    count = torch.empty(1, dtype=torch.float32, device=x.device)
    scratch = torch.empty(nb * Cc + nb, dtype=torch.float32, device=x.device)
    check(lib().sn_masked_colstats_f32(ptr(x), Cc, R, Cc, ptr(nvalid), int(K), ptr(mean), ptr(var), ptr(count),
                                       ptr(scratch), stream()), "sn_masked_colstats_f32")
    return mean, var, count


# `num_batches_tracked += 1` is one tiny launch per BatchNorm site; a training forward touches dozens.  Inside `batched_bn_counters()`
# the increments are collected and applied by one multi-tensor add at exit (same buffers, same values).
_BN_PENDING = None


def _count_batch(bn, n=1):
    if _BN_PENDING is None:
        bn.num_batches_tracked += n
    else:
        t = bn.num_batches_tracked
        ent = _BN_PENDING.get(id(t))
        if ent is None:
            _BN_PENDING[id(t)] = [t, n]
        else:
            ent[1] += n              # (one entry per buffer: a tensor listed twice in a foreach add is NOT reliably incremented twice)


class batched_bn_counters:
    def __enter__(self):
        global _BN_PENDING
        self._outer = _BN_PENDING
        if self._outer is None:
            _BN_PENDING = {}
        return self

    def __exit__(self, *exc):
        global _BN_PENDING
        if self._outer is None:
            pending, _BN_PENDING = _BN_PENDING, None
            if pending:
                with torch.no_grad():
                    torch._foreach_add_([e[0] for e in pending.values()], [e[1] for e in pending.values()])
        return False


def bn_train_stats(x, bn, nvalid=None, K=0):
    """Batch statistics of a train-mode BatchNorm1d over the valid rows of x, the folded (scale, shift), rstd, and the
    running-statistics side effect on `bn` — one C call (sn_bn_train_stats_f32).  -> (mean, var, rstd, scale, shift, count)."""
    require_cuda(x)
    x = _f32c(x, "x")
    Cc = x.shape[-1]
    R = x.numel() // Cc
    nb = int(lib().sn_colstats_blocks(R))
    out = torch.empty(5 * Cc + 1, dtype=torch.float32, device=x.device)
    mean, var, rstd, scale, shift = (out[i * Cc:(i + 1) * Cc] for i in range(5))
    count = out[5 * Cc:]
    scratch = torch.empty(nb * Cc + nb, dtype=torch.float32, device=x.device)
    track = bn.track_running_stats and bn.running_mean is not None
    if track and bn.momentum is None:
        raise NotImplementedError("BatchNorm1d(momentum=None) (cumulative moving average) is not supported")
    g = None if bn.weight is None else bn.weight.detach()
    b = None if bn.bias is None else bn.bias.detach()
    check(lib().sn_bn_train_stats_f32(ptr(x), Cc, R, Cc, ptr(nvalid), int(K), ptr(g), ptr(b), float(bn.eps),
                                      float(bn.momentum or 0.0), ptr(bn.running_mean) if track else None,
                                      ptr(bn.running_var) if track else None, ptr(mean), ptr(var), ptr(rstd), ptr(scale), ptr(shift),
                                      ptr(count), ptr(scratch), stream()), "sn_bn_train_stats_f32")
    if track:
        _count_batch(bn)
    return mean, var, rstd, scale, shift, count


def linear_bn_train(x, pl: PackedLinear, bn, nvalid=None, K=0):
    """z = x @ W^T + b (masked) and the train-mode BatchNorm1d statistics of z in one C call (sn_linear_bn_train_f32): for large row
    counts the moments come out of the Linear kernel's accumulators.  -> (z, mean, var, rstd, scale, shift, count); updates bn's
    running statistics like bn_train_stats."""
    require_cuda(x)
    x = _f32c(x, "x")
    if x.shape[-1] != pl.d_in:
        raise ValueError(f"linear_bn_train: x has {x.shape[-1]} channels, weight expects {pl.d_in}")
    R = x.numel() // pl.d_in
    Cc = pl.d_out
    z = torch.empty(*x.shape[:-1], Cc, dtype=torch.float32, device=x.device)
    out = torch.empty(5 * Cc + 1, dtype=torch.float32, device=x.device)
    mean, var, rstd, scale, shift = (out[i * Cc:(i + 1) * Cc] for i in range(5))
    count = out[5 * Cc:]
    scratch = torch.empty(max(int(lib().sn_linear_bn_scratch_floats(R, pl.d_in, Cc)), 1), dtype=torch.float32, device=x.device)
    track = bn.track_running_stats and bn.running_mean is not None
    if track and bn.momentum is None:
        raise NotImplementedError("BatchNorm1d(momentum=None) (cumulative moving average) is not supported")
    g = None if bn.weight is None else bn.weight.detach()
    b = None if bn.bias is None else bn.bias.detach()
    with _span("sn_linear_bn_train_f32"):
        check(lib().sn_linear_bn_train_f32(ptr(x), pl.d_in, R, pl.d_in, ptr(pl.wp), Cc, ptr(pl.bias), ptr(nvalid), int(K), ptr(z), Cc,
                                           ptr(g), ptr(b), float(bn.eps), float(bn.momentum or 0.0),
                                           ptr(bn.running_mean) if track else None, ptr(bn.running_var) if track else None, ptr(mean),
                                           ptr(var), ptr(rstd), ptr(scale), ptr(shift), ptr(count), ptr(scratch), stream()),
              "sn_linear_bn_train_f32")
    if track:
        _count_batch(bn)
    return z, mean, var, rstd, scale, shift, count


def masked_affine(x, nvalid=None, K=0, *, scale=None, shift=None, relu_pre=False, relu=False, residual=None):
    require_cuda(x)
    x = _f32c(x, "x")
    Cc = x.shape[-1]
    R = x.numel() // Cc
    flags = (EPI_RELU_PRE if relu_pre else 0) | (EPI_AFFINE if scale is not None else 0) | \
            (EPI_RELU if relu else 0) | (EPI_RESIDUAL if residual is not None else 0)
    out = torch.empty_like(x)
    with _span("sn_masked_affine_f32"):
        check(lib().sn_masked_affine_f32(ptr(x), Cc, R, Cc, ptr(nvalid), int(K), flags, ptr(scale), ptr(shift),
                                         ptr(residual), Cc, ptr(out), Cc, stream()), "sn_masked_affine_f32")
    return out


def masked_layernorm(x, residual, gamma, beta, eps, nvalid=None, K=0):
    require_cuda(x)
    x = _f32c(x, "x")
    Cc = x.shape[-1]
    out = torch.empty_like(x)
    with _span("sn_masked_layernorm_f32"):
        check(lib().sn_masked_layernorm_f32(ptr(x), ptr(residual), x.numel() // Cc, Cc, ptr(gamma), ptr(beta),
                                            float(eps), ptr(nvalid), int(K), ptr(out), stream()),
              "sn_masked_layernorm_f32")
    return out


def attention_dropout_mask(N, K, heads, p, device):
    """Bernoulli keep-mask of the attention dropout (transformer_module.py:49,55), scaled: 0 or 1/(1-p), [N, heads, K, K].
    Drawn with torch's device generator (reproducible under torch.manual_seed); None when p == 0."""
    if not p:
        return None
    u = torch.rand(N, heads, K, K, device=device)
    # (the same draws as `(u >= p).float() * 1/(1-p)`, thresholded and scaled in place by one launch instead of three)
    check(lib().sn_keep_mask_f32(ptr(u), u.numel(), float(p), float(1.0 / (1.0 - p)), stream()), "sn_keep_mask_f32")
    return u


def set_attention(q, k, v, N, K, heads, nvalid=None, prob_mask=None):
    require_cuda(q, k, v)
    q, k, v = _f32c(q, "q"), _f32c(k, "k"), _f32c(v, "v")
    D = q.shape[-1]
    if prob_mask is not None and (prob_mask.shape != (N, heads, K, K) or not prob_mask.is_contiguous()):
        raise ValueError("set_attention: prob_mask must be a contiguous [N, heads, K, K] tensor")
    out = torch.empty_like(q)
    with _span("sn_set_attention_f32"):
        check(lib().sn_set_attention_f32(ptr(q), ptr(k), ptr(v), N, K, heads, D // heads, ptr(nvalid), ptr(prob_mask), ptr(out),
                                         stream()), "sn_set_attention_f32")
    return out


def slot_sum(x, N, K):
    require_cuda(x)
    x = _f32c(x, "x")
    Cc = x.shape[-1]
    out = torch.empty(N, Cc, dtype=torch.float32, device=x.device)
    with _span("sn_slot_sum_f32"):
        check(lib().sn_slot_sum_f32(ptr(x), N, K, Cc, ptr(out), stream()), "sn_slot_sum_f32")
    return out


EMBEDDING_INDEX_ERROR = ("index out of range in embedding: a discrete node / edge feature value lies outside its table "
                         "(DiscreteEncoder's max_num_values) — nn.Embedding raises IndexError here too")


_DEFERRED_STATUS = None      # a list while a caller collects the ops' own status words instead of having each op read its own


class defer_status:
    """`with ops.defer_status() as words:` — ops that would read their own status word back (one host wait each: embedding_sum with
    status=None) append it to `words` instead; the caller checks them later (`ops.raise_deferred(words)`).  For steps that must not
    synchronise: a HIP-graph capture (serving.GraphedDGLForward), a loop that checks once per epoch."""

    def __enter__(self):
        global _DEFERRED_STATUS
        self._prev, _DEFERRED_STATUS = _DEFERRED_STATUS, []
        self.words = _DEFERRED_STATUS
        return self.words

    def __exit__(self, *exc):
        global _DEFERRED_STATUS
        _DEFERRED_STATUS = self._prev
        return False


def deferring() -> bool:
    return _DEFERRED_STATUS is not None


def defer(kind, tensor, message=None):
    """Hand a device-side check over to the collector of `defer_status`: kind 'embed' (non-zero: an index outside its embedding table),
    'plan' (a GraphPlan: its check()), 'flag' (non-zero: ValueError(message))."""
    _DEFERRED_STATUS.append((kind, tensor, message))


def raise_deferred(words):
    """The collected checks, read back now (a host wait): raises what the ops themselves would have raised."""
    for kind, t, msg in words:
        if kind == "plan":
            t.check()
        elif kind == "embed":
            if int(t.reshape(-1)[0]):
                raise IndexError(EMBEDDING_INDEX_ERROR)
        elif bool(t.reshape(-1)[0]):
            raise ValueError(msg)


def embedding_sum(idx, tables, status=None):
    """sum_f tables[f][idx[:, f]] — DiscreteEncoder; idx int64 [R] or [R, F].  An index outside its table is never
    dereferenced (it contributes 0) and sets bit 0 of `status` (device int32, e.g. a slot of the batch plan's status block
    that the caller checks later); with status=None the op checks it itself — one host sync — and raises IndexError like
    nn.Embedding does."""
    require_cuda(idx)
    if idx.dtype != torch.int64:
        raise ValueError("embedding_sum: integer features must be int64")
    if idx.dim() == 1:
        idx = idx.unsqueeze(1)
    idx = idx.contiguous()
    R, nf = idx.shape
    if nf > len(tables):
        raise ValueError("embedding_sum: more feature columns than embedding tables")
    tabs = [_f32c(t, "embedding table") for t in tables[:nf]]
    Cc = tabs[0].shape[1]
    arr = (C.c_void_p * nf)(*[t.data_ptr() for t in tabs])
    rows = (C.c_int64 * nf)(*[t.shape[0] for t in tabs])
    out = torch.empty(R, Cc, dtype=torch.float32, device=idx.device)
    if R == 0:                      # (a batch without edges / an empty shard)
        return out
    own = status is None
    if own:
        status = torch.zeros(1, dtype=torch.int32, device=idx.device)
    with _span("sn_embedding_sum_f32"):
        check(lib().sn_embedding_sum_f32(ptr(idx), nf, nf, R, arr, rows, Cc, ptr(out), ptr(status), stream()), "sn_embedding_sum_f32")
    if own and _DEFERRED_STATUS is not None:
        defer("embed", status)
    elif own and int(status.item()):
        raise IndexError(EMBEDDING_INDEX_ERROR)
    return out


def segment_pool(x, plan: GraphPlan, mode="add"):
    require_cuda(x)
    x = _f32c(x, "x")
    out = torch.empty(plan.B, x.shape[1], dtype=torch.float32, device=x.device)
    with _span("sn_segment_pool_f32"):
        check(lib().sn_segment_pool_f32(ptr(x), plan.B, x.shape[1], ptr(plan.graph_ptr), 1 if mode == "mean" else 0,
                                        ptr(out), stream()), "sn_segment_pool_f32")
    return out


def ign_contract_2to1(X):
    """X [b, 1, n, n] or [b, n, n] -> ops [b, n, 5] (IGN 2->1 contractions, ign.py:344-374)."""
    require_cuda(X)
    X = _f32c(X, "X")
    n = X.shape[-1]
    if X.shape[-2] != n or (X.dim() == 4 and X.shape[1] != 1) or X.dim() not in (3, 4):
        raise ValueError("ign_contract_2to1: expected [b, 1, n, n] or [b, n, n]")
    b = X.shape[0]
    out = torch.empty(b, n, 5, dtype=torch.float32, device=X.device)
    scratch = torch.empty(int(lib().sn_ign_contract_scratch_floats(b, n)), dtype=torch.float32, device=X.device)
    with _span("sn_ign_contract_2to1_f32"):
        check(lib().sn_ign_contract_2to1_f32(ptr(X), b, n, ptr(out), ptr(scratch), stream()), "sn_ign_contract_2to1_f32")
    return out


def pna_aggregate(msg, hself, plan: GraphPlan, avg_log: float):
    """PNA tower aggregation (pna_layer.py:50-56,69): msg [E, C] per-edge messages in edge-id order, hself [N, C] the tower's
    input rows -> [N, 13*C] = cat[hself, scalers(aggregators(msg over in-edges))]."""
    require_cuda(msg)
    msg, hself = _f32c(msg, "msg"), _f32c(hself, "hself")
    Cc = msg.shape[1]
    out = torch.empty(plan.N, 13 * Cc, dtype=torch.float32, device=msg.device)
    with _span("sn_pna_aggregate_f32"):
        check(lib().sn_pna_aggregate_f32(ptr(msg), Cc, ptr(hself), Cc, Cc, plan.N, ptr(plan.rowptr), ptr(plan.eperm), float(avg_log),
                                         ptr(out), 13 * Cc, stream()), "sn_pna_aggregate_f32")
    return out


def grouped_linear(x, W, bias, G, *, rowscale=None, scale=None, shift=None):
    """y[:, g*dout:(g+1)*dout] = ((x[:, g*din:(g+1)*din] @ W[g]^T + b_g) * rowscale[row]) * scale + shift — G independent column groups
    (a block-diagonal Linear); W [G, dout, din] float32 contiguous, dout <= 16, din % 4 == 0 and <= 256."""
    require_cuda(x)
    x = _f32c(x, "x")
    Gn, dout, din = W.shape
    if Gn != G or x.shape[-1] != G * din:
        raise ValueError("grouped_linear: x must be [R, G*din] for W [G, dout, din]")
    R = x.numel() // (G * din)
    y = torch.empty(R, G * dout, dtype=torch.float32, device=x.device)
    with _span("sn_grouped_linear_f32"):
        check(lib().sn_grouped_linear_f32(ptr(x), G * din, R, G, din, dout, ptr(W), ptr(bias), ptr(rowscale), ptr(scale), ptr(shift), ptr(y),
                                          G * dout, stream()), "sn_grouped_linear_f32")
    return y


def pna_aggregate_gather(psd, qe, hself, plan: GraphPlan, avg_log: float, tower_width: int = 0, qe_layer=None):
    """PNA aggregation with the pretrans message formed in the kernel: psd [N, 2C] = [W_s h | W_d h] per node, qe [E, C] = W_e e + b,
    message (j -> n, e) = psd[j, :C] + psd[n, C:] + qe[e]; -> [N, 13*C] = cat[hself, scalers(aggregators(.))] (all towers side by side;
    tower_width = it > 0: tower-major columns [tower][13 blocks][it], the input layout of grouped_linear)."""
    require_cuda(psd)
    psd, qe, hself = _f32c(psd, "psd"), _f32c(qe, "qe"), _f32c(hself, "hself")
    Cc, ldq, qoff = qe.shape[1], qe.shape[1], 0
    if qe_layer is not None:    # qe is [E, L*C], every layer's edge term side by side: this layer's column block is read in place
        Cc = hself.shape[1]
        if qe.shape[1] % Cc or not 0 <= qe_layer < qe.shape[1] // Cc:
            raise ValueError("pna_aggregate_gather: qe must be [E, L*C] with 0 <= qe_layer < L")
        qoff = 4 * qe_layer * Cc
    if psd.shape[1] != 2 * Cc or hself.shape[1] != Cc:
        raise ValueError("pna_aggregate_gather: psd must be [N, 2C], hself [N, C] for qe [E, C]")
    out = torch.empty(plan.N, 13 * Cc, dtype=torch.float32, device=psd.device)
    with _span("sn_pna_aggregate_gather_f32"):
        check(lib().sn_pna_aggregate_gather_f32(psd.data_ptr(), 2 * Cc, psd.data_ptr() + 4 * Cc, 2 * Cc, qe.data_ptr() + qoff, ldq, ptr(hself), Cc, Cc, plan.N,
                                                ptr(plan.rowptr), ptr(plan.col), ptr(plan.eperm), float(avg_log), ptr(out), 13 * Cc, int(tower_width), stream()),
              "sn_pna_aggregate_gather_f32")
    return out


def edge_attention(Q, K, V, Ee, plan: GraphPlan, heads: int):
    """Sparse multi-head attention over the graph's edges with edge features (layers/transformer.py:150-228) -> [N, heads*dk]."""
    require_cuda(Q)
    Q, K, V, Ee = (_f32c(t, n) for t, n in ((Q, "Q"), (K, "K"), (V, "V"), (Ee, "E")))
    d = Q.shape[1]
    if d % heads or d // heads > 32:
        raise ValueError("edge_attention: heads must divide the width and the head width must be <= 32")
    out = torch.empty(plan.N, d, dtype=torch.float32, device=Q.device)
    with _span("sn_edge_attention_f32"):
        check(lib().sn_edge_attention_f32(ptr(Q), ptr(K), ptr(V), ptr(Ee), plan.N, int(heads), d // heads, ptr(plan.rowptr), ptr(plan.col),
                                          ptr(plan.eperm), ptr(out), stream()), "sn_edge_attention_f32")
    return out


def edge_attention_fused(qkv, Ee_all, layer: int, plan: GraphPlan, heads: int):
    """The same attention on the column blocks of fused projections: qkv [N, 3*d] = [Q | K | V] of one Linear over cat[W_Q; W_K; W_V],
    Ee_all [E, L*d] = every layer's E projection of the edge embedding side by side (layer `layer` is read) — no copies."""
    require_cuda(qkv)
    if qkv.dtype != torch.float32 or not qkv.is_contiguous() or Ee_all.dtype != torch.float32 or not Ee_all.is_contiguous():
        raise ValueError("edge_attention_fused: contiguous float32 matrices expected")
    d = qkv.shape[1] // 3
    if qkv.shape[1] != 3 * d or d % heads or d // heads > 32 or Ee_all.shape[1] % d or not 0 <= layer < Ee_all.shape[1] // d:
        raise ValueError("edge_attention_fused: qkv must be [N, 3*d], Ee_all [E, L*d], head width <= 32")
    out = torch.empty(plan.N, d, dtype=torch.float32, device=qkv.device)
    base, eb = qkv.data_ptr(), Ee_all.data_ptr() + 4 * d * layer
    with _span("sn_edge_attention_f32"):
        check(lib().sn_edge_attention_strided_f32(C.c_void_p(base), C.c_void_p(base + 4 * d), C.c_void_p(base + 8 * d), 3 * d, C.c_void_p(eb),
                                                  Ee_all.shape[1], plan.N, int(heads), d // heads, ptr(plan.rowptr), ptr(plan.col),
                                                  ptr(plan.eperm), ptr(out), stream()), "sn_edge_attention_strided_f32")
    return out


def gat_aggregate(feat, attn_l, attn_r, bias, plan: GraphPlan, heads: int, negative_slope=0.2, relu=True, want_lse=False):
    """DGL GATConv after its fc (gat_net.py:62-66): feat [N, heads*C] -> [N, heads*C]; see sn_gat_aggregate_f32."""
    require_cuda(feat)
    feat = _f32c(feat, "feat")
    d = feat.shape[1]
    if d % heads or d // heads > 64:
        raise ValueError("gat_aggregate: heads must divide the width and the head width must be <= 64")
    al, ar = _f32c(attn_l.reshape(-1), "attn_l"), _f32c(attn_r.reshape(-1), "attn_r")
    b = None if bias is None else _f32c(bias.reshape(-1), "bias")
    out = torch.empty(plan.N, d, dtype=torch.float32, device=feat.device)
    lse = torch.empty(plan.N, heads, dtype=torch.float32, device=feat.device) if want_lse else None
    with _span("sn_gat_aggregate_f32"):
        check(lib().sn_gat_aggregate_f32(ptr(feat), ptr(al), ptr(ar), ptr(b), plan.N, int(heads), d // heads, float(negative_slope), int(relu),
                                         ptr(plan.rowptr), ptr(plan.col), ptr(out), ptr(lse), stream()), "sn_gat_aggregate_f32")
    return (out, lse) if want_lse else out


def pointwise(x, *, rowscale=None, scale=None, shift=None, act="none", slope=0.01, residual=None):
    """y = act((x * rowscale[r]) * scale[c] + shift[c]) + residual, act in none / relu / leaky."""
    require_cuda(x)
    x = _f32c(x, "x")
    Cc = x.shape[-1]
    R = x.numel() // Cc
    out = torch.empty_like(x)
    a = {"none": 0, "relu": 1, "leaky": 2, "relu_sum": 3}[act]
    if rowscale is not None:
        rowscale = _f32c(rowscale.reshape(-1), "rowscale")
        if rowscale.numel() != R:
            raise ValueError("pointwise: one row scale per row expected")
    check(lib().sn_pointwise_f32(ptr(x), Cc, R, Cc, ptr(rowscale), ptr(scale), ptr(shift), a, float(slope),
                                 ptr(None if residual is None else _f32c(residual, "residual")), Cc, ptr(out), Cc, stream()),
          "sn_pointwise_f32")
    return out


def dense_attention(q, k, v, heads, want_lse=False):
    """softmax(q k^T / sqrt(dk)) v per head over whole sequences (LearningFilters/models.py:115-135's encoder attention).
    q, k, v: [Bt, L, heads*dk] (batch_first), dk <= 32."""
    require_cuda(q)
    q, k, v = _f32c(q, "q"), _f32c(k, "k"), _f32c(v, "v")
    if q.dim() != 3 or k.shape != q.shape or v.shape != q.shape or q.shape[-1] % heads:
        raise ValueError("dense_attention: q, k, v must be [Bt, L, heads*dk] of one shape")
    Bt, L, d = q.shape
    out = torch.empty_like(q)
    lse = torch.empty(Bt, heads, L, dtype=torch.float32, device=q.device)
    with _span("sn_dense_attention_f32"):
        check(lib().sn_dense_attention_f32(ptr(q), ptr(k), ptr(v), Bt, L, heads, d // heads, ptr(out), ptr(lse), stream()),
              "sn_dense_attention_f32")
    return (out, lse) if want_lse else out


class EigenspacePlan:
    """Device-side result of sn_eigenspace_group (LearningFilters/training.py:47-61 without the projectors) + the few numbers the
    host needs to size tensors (read back once: this is a per-graph one-off, as in the reference).
    mults: sorted distinct multiplicities; counts[i]: eigenspaces with multiplicity mults[i]; slot_base[i]: first slot of that
    multiplicity in the multiplicity-major stack (the order of the reference's {mult: torch.cat(projectors)} dict)."""

    def __init__(self, N, space_of, space_start, space_mult, space_slot, mults, counts, n_spaces, max_mult):
        self.N, self.space_of, self.space_start, self.space_mult, self.space_slot = N, space_of, space_start, space_mult, space_slot
        self.mults, self.counts, self.n_spaces, self.max_mult = mults, counts, n_spaces, max_mult
        self.slot_base = {}
        off = 0
        for m, c in zip(mults, counts):
            self.slot_base[m] = off
            off += c

    def group(self, stack, mult):
        """The rows of a [n_spaces, ...] slot-ordered stack that belong to multiplicity `mult`."""
        i = self.mults.index(mult)
        off = self.slot_base[mult]
        return stack[off:off + self.counts[i]]


def eigenspace_group(eigvals, decimals=5) -> EigenspacePlan:
    """`around(eigvals, 5)` + `unique(return_counts)` + the multiplicity grouping of training.py:47-73, on the device."""
    require_cuda(eigvals)
    ev = _f32c(eigvals, "eigvals")
    N = ev.numel()
    ints = torch.empty(6 * N + 8, dtype=torch.int32, device=ev.device)
    space_of, space_start, space_mult, space_slot = ints[:N], ints[N:2 * N + 1], ints[2 * N + 1:3 * N + 1], ints[3 * N + 1:4 * N + 1]
    mult_list, mult_count, meta = ints[4 * N + 1:5 * N + 1], ints[5 * N + 1:6 * N + 1], ints[6 * N + 1:6 * N + 5]
    check(lib().sn_eigenspace_group(ptr(ev), N, int(decimals), ptr(space_of), ptr(space_start), ptr(space_mult), ptr(space_slot),
                                    ptr(mult_list), ptr(mult_count), ptr(meta), stream()), "sn_eigenspace_group")
    ns, nm, err, mmax = meta.tolist()                   # one host sync (per graph, once)
    if err:
        raise ValueError("eigenspace_group: the eigenvalues must be in ascending order (as eigh returns them)")
    return EigenspacePlan(N, space_of, space_start, space_mult, space_slot, mult_list[:nm].tolist(), mult_count[:nm].tolist(), ns, mmax)


def eigenspace_projectors(eigvecs, plan: EigenspacePlan):
    """[n_spaces, N, N] stack of P_s = V_s V_s^T in multiplicity-major order (training.py:61-73)."""
    require_cuda(eigvecs)
    V = _f32c(eigvecs, "eigvecs")
    N = plan.N
    if V.shape != (N, N):
        raise ValueError("eigenspace_projectors: eigvecs must be [N, N] (V[node, eigenvector])")
    out = torch.empty(plan.n_spaces, N, N, dtype=torch.float32, device=V.device)
    with _span("sn_eigenspace_projectors_f32"):
        check(lib().sn_eigenspace_projectors_f32(ptr(V), N, N, ptr(plan.space_start), ptr(plan.space_slot), plan.n_spaces, ptr(out),
                                                 stream()), "sn_eigenspace_projectors_f32")
    return out


def ign_contract_eigvecs(eigvecs, plan: EigenspacePlan):
    """The 2->1 contractions [n_spaces, N, 5] of every projector V_s V_s^T, computed from V alone (no N x N matrix)."""
    require_cuda(eigvecs)
    V = _f32c(eigvecs, "eigvecs")
    N = plan.N
    out = torch.empty(plan.n_spaces, N, 5, dtype=torch.float32, device=V.device)
    with _span("sn_ign_contract_eigvecs_f32"):
        check(lib().sn_ign_contract_eigvecs_f32(ptr(V), N, N, ptr(plan.space_start), ptr(plan.space_slot), plan.n_spaces,
                                                plan.max_mult, ptr(out), stream()), "sn_ign_contract_eigvecs_f32")
    return out


EVD_STATUS = {1: "an edge leaves its graph (or a node id is out of range)", 2: "a graph has more than 64 nodes",
              4: "the Jacobi iteration did not converge", 8: "eigen_vectors buffer too small"}


def laplacian_evd(edge_index, graph_ptr, N, total, norm=None, pos_enc_dim=0, skip=1):
    """Batched Laplacian eigendecomposition on the device (transform.py:7-23 for every graph of a collated batch).

    edge_index [2,E] int64, graph_ptr [B+1] int32 (device), N nodes, total = sum n_b^2 (host int).
    Returns (eigen_values [N], eigen_vectors [total], evoff [B+1] int64, pos_enc [N,k] or None, status int32[4])."""
    require_cuda(edge_index, graph_ptr)
    if norm not in (None, "sym"):
        raise ValueError(f"unsupported normalization {norm!r} (None or 'sym')")
    if edge_index.dtype != torch.int64 or edge_index.dim() != 2 or edge_index.shape[0] != 2:
        raise ValueError("edge_index: expected int64 [2, E]")
    if graph_ptr.dtype != torch.int32:
        raise ValueError("graph_ptr: expected int32 [B+1]")
    edge_index = edge_index.contiguous()
    dev = edge_index.device
    B, E = graph_ptr.numel() - 1, edge_index.shape[1]
    val = torch.empty(N, dtype=torch.float32, device=dev)
    vec = torch.empty(total, dtype=torch.float32, device=dev)
    evoff = torch.empty(B + 1, dtype=torch.int64, device=dev)
    pe = torch.empty(N, pos_enc_dim, dtype=torch.float32, device=dev) if pos_enc_dim > 0 else None
    work = torch.empty(int(lib().sn_evd_work_ints(B)), dtype=torch.int32, device=dev)
    status = torch.empty(4, dtype=torch.int32, device=dev)
    with _span("sn_laplacian_evd_f32"):
        check(lib().sn_laplacian_evd_f32(ptr(edge_index), E, ptr(graph_ptr), B, N, 0 if norm is None else 1, ptr(evoff),
                                         ptr(val), ptr(vec), total, ptr(pe), int(pos_enc_dim), int(skip), ptr(work),
                                         ptr(status), stream()), "sn_laplacian_evd_f32")
    return val, vec, evoff, pe, status


def bn_fold_stats(weight, bias, mean, var, eps, c_pad=None):
    """(scale, shift) of a BatchNorm from explicit statistics (batch statistics of the train-mode / no-running-stats case)."""
    Cc = mean.numel()
    cp = Cc if c_pad is None else int(c_pad)
    scale = torch.empty(cp, dtype=torch.float32, device=mean.device)
    shift = torch.empty(cp, dtype=torch.float32, device=mean.device)
    check(lib().sn_bn_fold_f32(ptr(weight), ptr(bias), ptr(mean), ptr(var), float(eps), Cc, cp, ptr(scale), ptr(shift),
                               stream()), "sn_bn_fold_f32")
    return scale, shift


def bn_running_update(bn, mean, var, count):
    """Side effect of a train-mode BatchNorm1d forward on its buffers (momentum None = cumulative average is not supported)."""
    if not bn.track_running_stats or bn.running_mean is None:
        return
    if bn.momentum is None:
        raise NotImplementedError("BatchNorm1d(momentum=None) (cumulative moving average) is not supported")
    check(lib().sn_bn_running_update_f32(ptr(mean), ptr(var), ptr(count), float(bn.momentum), bn.num_features,
                                         ptr(bn.running_mean), ptr(bn.running_var), stream()), "sn_bn_running_update_f32")
    _count_batch(bn)


# ----------------------------------------------------------------------------- roofline accounting (bench.py)
MFMA_F32_PEAK_TF = 157.3     # MI355X_MICROARCH.md: fp32-input MFMA dense peak (= fp32 vector peak)
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E spec


# fp32 GEMMs evaluated as six bf16 partial products (csrc/fused_common.hpp): the bound is the dense bf16 MFMA peak
# (MI355X_MICROARCH.md: ~2.5 PFLOP/s) divided by the 6 MFMAs each fp32 multiply-add costs.
MFMA_BF16_PEAK_TF = 2500.0
MFMA_SPLIT_F32_PEAK_TF = MFMA_BF16_PEAK_TF / 6.0


def _mfma_roof(label, flops, mean_ms, per_step, peak=None, peak_note=None):
    peak = MFMA_F32_PEAK_TF if peak is None else peak
    t = mean_ms * per_step * 1e-3
    ach = flops / t / 1e12
    out = {"kernel": label, "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
           "frac": ach / peak, "traffic": None, "flops_per_launch": flops / per_step, "mean_launch_us": mean_ms * 1e3}
    if peak_note:
        out["peak_note"] = peak_note
    return out


_SPLIT_NOTE = ("fp32-equivalent flops; each fp32 product = 6 v_mfma_f32_16x16x32_bf16 partial products (exact 3 x bf16 "
               "operand split, fp32 accumulate): peak = 2500 TFLOP/s dense bf16 / 6; the fp32-input MFMA peak is 157.3")


def _roof_linear(fl, wl, host, mean_ms, per_step):
    """Layer path: all dense contractions go through sn_masked_linear_f32 (attention's K x K part excepted)."""
    flops = fl["total"] - wl["nl_rho"] * 4 * sum(n * min(n, wl["k"]) ** 2 for n in host.sizes) * wl["hidden"]
    return _mfma_roof("sn_masked_linear_f32 (k_linear, all launches of a step)", flops, mean_ms, per_step)


def _roof_phi(fl, wl, host, mean_ms, per_step):
    """Fused phi: algorithmic flops = 2 signs x (L-1) layers x 2 Linear x 2*d*d per VALID (node, slot) row
    (SURVEY.md §8(d)); padding rows of the work bins are not counted."""
    return _mfma_roof("sn_phi_fused_f32 (k_phi_fused)", fl["phi"], mean_ms, per_step, MFMA_SPLIT_F32_PEAK_TF, _SPLIT_NOTE)


def _roof_rho(fl, wl, host, mean_ms, per_step):
    return _mfma_roof("sn_rho_fused_f32 (k_rho_fused)", fl["rho"] - 2 * fl["N"] * wl["hidden"] ** 2, mean_ms, per_step,
                      MFMA_SPLIT_F32_PEAK_TF, _SPLIT_NOTE)


def _roof_gnn(fl, wl, host, mean_ms, per_step):
    return _mfma_roof("sn_gnn_fused_f32 (k_gnn_coop)", fl["gnn"] + 2 * fl["N"] * wl["hidden"] ** 2, mean_ms, per_step,
                      MFMA_SPLIT_F32_PEAK_TF, _SPLIT_NOTE)


KERNEL_ROOFLINE = {"sn_masked_linear_f32": _roof_linear, "sn_phi_fused_f32": _roof_phi, "sn_rho_fused_f32": _roof_rho,
                   "sn_gnn_fused_f32": _roof_gnn}
