"""Stage-level training ops: ONE pass over the rows per direction for a "Linear -> BatchNorm1d(train) -> ReLU" link (csrc/train.hip).

What `loss.backward()` runs through ATen for a MaskedMLP / MLP of the reference (Alchemy/sign_net/model_utils/masked_layers.py:34-64,
GINESignNetPyG/core/model_utils/elements.py:40-69; called from sign_net.py:28-44 for phi, model.py:36-64 for the GINE stack) is about ten
launches and ten passes over the [rows, d] activations per Linear and direction.  Here

    mlp2_bn : y = [relu](bn_b(W_b relu(bn_a(W_a x + b_a)) + b_b)) [+ residual]      forward 5 launches, backward 7
    linear  : y = [relu](W x + b)                                                     forward 1 launch,   backward 2

for G groups of rows at once (the phi(+x) / phi(-x) passes: shared weights, separate batch statistics).  Raw parameters go straight
to the kernels (no per-step weight packing); gradients are bitwise reproducible (no atomics).  torch records the graph and owns the
buffers; no arithmetic runs in ATen.
"""
from __future__ import annotations

import ctypes as C

import torch
from torch.autograd import Function

from . import ops
from ._lib import check, lib, ptr, stream


class _LinArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("ldx", C.c_int), ("R", C.c_int64), ("G", C.c_int), ("d_in", C.c_int), ("d_out", C.c_int),
                ("W", C.c_void_p), ("ldw", C.c_int), ("bias", C.c_void_p), ("nvalid", C.c_void_p), ("K", C.c_int),
                ("in_scale", C.c_void_p), ("in_shift", C.c_void_p), ("in_relu", C.c_int), ("out_relu", C.c_int),
                ("y", C.c_void_p), ("ldy", C.c_int), ("stat_part", C.c_void_p),
                ("fin_gamma", C.c_void_p), ("fin_beta", C.c_void_p), ("fin_eps", C.c_float), ("fin_momentum", C.c_float),
                ("fin_running_mean", C.c_void_p), ("fin_running_var", C.c_void_p), ("fin_state", C.c_void_p), ("fin_count", C.c_void_p)]


class _BwdArgs(C.Structure):
    _fields_ = [("R", C.c_int64), ("G", C.c_int), ("nvalid", C.c_void_p), ("K", C.c_int), ("d_in", C.c_int), ("d_out", C.c_int),
                ("dy", C.c_void_p), ("lddy", C.c_int), ("zo", C.c_void_p), ("ldzo", C.c_int),
                ("coef_a", C.c_void_p), ("coef_b", C.c_void_p), ("coef_c", C.c_void_p),
                ("mask_scale", C.c_void_p), ("mask_shift", C.c_void_p),
                ("x", C.c_void_p), ("ldx", C.c_int), ("x_scale", C.c_void_p), ("x_shift", C.c_void_p), ("x_relu", C.c_int),
                ("x_mean", C.c_void_p), ("W", C.c_void_p), ("ldw", C.c_int), ("gx", C.c_void_p), ("ldgx", C.c_int),
                ("sums_part", C.c_void_p), ("dw_part", C.c_void_p), ("want_db", C.c_int), ("gx_accumulate", C.c_int),
                ("dot_x", C.c_void_p), ("lddot", C.c_int), ("dot_part", C.c_void_p),
                ("fin_state", C.c_void_p), ("fin_count", C.c_void_p), ("fin_gamma", C.c_void_p), ("fin_coef", C.c_void_p),
                ("fin_dgamma", C.c_void_p), ("fin_dbeta", C.c_void_p), ("fin_accumulate", C.c_int), ("fin_dot_out", C.c_void_p),
                ("merge_sums", C.c_void_p), ("merge_nblk", C.c_int), ("merge_state", C.c_void_p), ("merge_count", C.c_void_p),
                ("merge_gamma", C.c_void_p), ("merge_dgamma", C.c_void_p), ("merge_dbeta", C.c_void_p), ("merge_accumulate", C.c_int)]


class _SMlpArgs(C.Structure):
    _fields_ = [("a", C.c_void_p), ("M", C.c_int64), ("G", C.c_int), ("negate_second", C.c_int), ("nvalid", C.c_void_p), ("K", C.c_int),
                ("d", C.c_int), ("w1", C.c_void_p), ("gamma_a", C.c_void_p), ("beta_a", C.c_void_p), ("eps_a", C.c_float),
                ("w2", C.c_void_p), ("b2", C.c_void_p), ("gamma_b", C.c_void_p), ("beta_b", C.c_void_p), ("eps_b", C.c_float),
                ("relu_b", C.c_int), ("scalar_state", C.c_void_p), ("column_state", C.c_void_p)]


# Launch structure of a link's reductions (environment, read once; profiles/scripts/train_ab.sh measures all four combinations):
#   SN_TRAIN_FUSE_FINISH=1  the finishes of a link's batch statistics / BatchNorm backward / eps gradient run INSIDE the link's launch, by
#                           its last-arriving workgroup (agent-scope ticket behind write-through partials): 175 -> 119 launches per step,
#                           the same reduction order (last-bit differences from FMA contraction) — and 0.10-0.29 ms SLOWER per replayed
#                           step (3.10 -> 3.31 ms, 3.22 -> 3.32, 3.08 -> 3.37 on three boxes): one
#                           workgroup pulling 130-260 KB of partials through one CU's memory path behind an atomic round trip costs more
#                           than the ~2 us boundary + 8-block finish kernel it replaces.  OFF by default; kept for the A/B and because
#                           a host-bound EAGER loop gains from it (4.32 -> 3.90 ms).
#   SN_TRAIN_DEFER_DW=0     the dW / db reduction behind every link instead of one launch at the end of loss.backward().
import os as _os
#   SN_TRAIN_MERGE=0        the BatchNorm-backward coefficients from a finish launch in front of the backward link instead of merged by the
#                           link itself in its prologue (the consumer-side form: removes the finish step instead of moving it).
FUSE_FINISH = _os.environ.get("SN_TRAIN_FUSE_FINISH", "0") == "1"
MERGE_COEF = _os.environ.get("SN_TRAIN_MERGE", "1") != "0"
DEFER_DW = _os.environ.get("SN_TRAIN_DEFER_DW", "1") != "0"


def supported(d_in: int, d_out: int) -> bool:
    return 0 < d_in <= 128 and 0 < d_out <= 128 and d_in % 4 == 0 and d_out % 4 == 0


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _w(W):
    W = W.detach()
    return W if W.stride(-1) == 1 else W.contiguous()


class _PostArgs(C.Structure):
    _fields_ = [("dw_part", C.c_void_p), ("nparts", C.c_int), ("stride", C.c_int64), ("n_w", C.c_int64), ("dw_out", C.c_void_p),
                ("n_b", C.c_int64), ("db_out", C.c_void_p),
                ("sums_part", C.c_void_p), ("nblk", C.c_int), ("G", C.c_int), ("C", C.c_int), ("state", C.c_void_p), ("count", C.c_void_p),
                ("gamma", C.c_void_p), ("coef", C.c_void_p), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p), ("accumulate_bn", C.c_int),
                ("dot_part", C.c_void_p), ("dot_n", C.c_int), ("dot_out", C.c_void_p)]


class _ReduceJob(C.Structure):
    _fields_ = [("part", C.c_void_p), ("nparts", C.c_int), ("stride", C.c_int64), ("n", C.c_int64), ("out", C.c_void_p),
                ("accumulate", C.c_int)]


class _DotJob(C.Structure):
    _fields_ = [("part", C.c_void_p), ("n", C.c_int), ("out", C.c_void_p)]


MAX_REDUCE_JOBS = 64


class _Deferred:
    """The dW / db partial sums of the backward links of ONE loss.backward(), reduced into the parameters' gradients by a single launch
    when the backward pass ends (sn_train_reduce_jobs_f32) instead of one small launch behind every link: nothing reads a weight
    gradient before the optimiser (or a gradient all-reduce: optim.FlatAdam flushes first).  Only gradients that are accumulated in
    place (`direct_grad`) are deferred; the arithmetic per parameter is that of sn_train_reduce_parts_f32, bit for bit."""
    jobs = []            # (partials tensor [kept alive], byte offset, nparts, stride, n, out tensor)
    dots = []            # (float64 partials [kept alive], out tensor): the eps gradients, one launch for all of them
    outs = set()
    armed = False
    task = -1            # autograd graph-task id of the backward pass the callback was queued in


def _defer_reduce(part, off_floats, nparts, stride, n, out):
    key = out.data_ptr()
    if key in _Deferred.outs or len(_Deferred.jobs) >= 4 * MAX_REDUCE_JOBS:       # two adds into one gradient must not share a launch
        flush_deferred()
    _Deferred.jobs.append((part, off_floats, nparts, stride, n, out))
    _Deferred.outs.add(key)
    _arm()


def _arm():
    # one end-of-backward callback per BACKWARD PASS (autograd's graph-task id): a pass that raised before its callback ran must not leave
    # the next one without (a callback that finds nothing queued does nothing)
    task = _graph_task_id()
    if not _Deferred.armed or task != _Deferred.task:
        try:        # runs on the caller's stream, inside a HIP-graph capture too
            torch.autograd.Variable._execution_engine.queue_callback(flush_deferred)
            _Deferred.armed, _Deferred.task = True, task
        except RuntimeError:        # not inside a backward pass (a direct call of linear_bwd): reduce now
            flush_deferred()


_graph_task_id = getattr(torch._C, "_current_graph_task_id", lambda: -1)


def _defer_dot(part, out):
    key = out.data_ptr()
    if key in _Deferred.outs or len(_Deferred.dots) >= MAX_REDUCE_JOBS:
        flush_deferred()
    _Deferred.dots.append((part, out))
    _Deferred.outs.add(key)
    _arm()


def flush_deferred():
    _Deferred.armed = False
    dots, _Deferred.dots = _Deferred.dots, []
    if dots:
        arr = (_DotJob * len(dots))()
        for j, (part, out) in enumerate(dots):
            arr[j] = _DotJob(part.data_ptr(), part.numel(), out.data_ptr())
        with ops._span("sn_train_dot_jobs_f64"):
            check(lib().sn_train_dot_jobs_f64(arr, len(dots), stream()), "sn_train_dot_jobs_f64")
    jobs, _Deferred.jobs, _Deferred.outs = _Deferred.jobs, [], set()
    for i in range(0, len(jobs), MAX_REDUCE_JOBS):
        chunk = jobs[i:i + MAX_REDUCE_JOBS]
        arr = (_ReduceJob * len(chunk))()
        for j, (part, off, nparts, stride, n, out) in enumerate(chunk):
            arr[j] = _ReduceJob(part.data_ptr() + 4 * off, nparts, stride, n, out.data_ptr(), 1)
        with ops._span("sn_train_reduce_jobs_f32"):
            check(lib().sn_train_reduce_jobs_f32(arr, len(chunk), stream()), "sn_train_reduce_jobs_f32")


def direct_grad(p):
    """The buffer a parameter's gradient may be accumulated into IN the adjoint kernels (`optim.FlatAdam` marks its parameters: their
    .grad are views of one zeroed flat buffer, so `+=` in the kernel is exactly autograd's AccumulateGrad without the ~100 tiny add
    launches per step); None: hand the gradient to autograd as usual.  (autograd still runs the parameter's post-accumulate hooks after
    an adjoint that returned None for it — the bucketed all-reduce of FlatAdam needs no extra notification.)"""
    if p is None or not getattr(p, "_sn_direct_grad", False):
        return None
    g = p.grad
    return g if (g is not None and g.is_contiguous() and g.dtype == torch.float32 and g.device == p.device) else None


class BNState:
    """Batch statistics of one train-mode BatchNorm site: state[(mean, var, rstd, scale, shift)][g][C]; count[g]."""

    def __init__(self, G, Cc, device):
        self.G, self.C = G, Cc
        self.state = torch.empty(5, G, Cc, dtype=torch.float32, device=device)
        self.count = torch.empty(G, dtype=torch.float32, device=device)

    mean = property(lambda s: s.state[0])
    scale = property(lambda s: s.state[3])
    shift = property(lambda s: s.state[4])


# ----------------------------------------------------------------------------- raw launches
def linear_fwd(x, R, G, W, b, nvalid, K, in_state=None, in_relu=False, out_relu=False, bn=None):
    """x [G*R, d_in] -> y [G*R, d_out] (+ the BatchNorm state of `bn` over y's valid rows, running statistics updated)."""
    W = _w(W)
    d_out, d_in = W.shape
    y = torch.empty(G * R, d_out, dtype=torch.float32, device=x.device)
    stat = None
    nblk = int(lib().sn_train_linear_blocks(R, G))
    if bn is not None:
        stat = torch.empty(G * (2 * nblk * d_out + nblk), dtype=torch.float32, device=x.device)
    a = _LinArgs(ptr(x), x.stride(0), R, G, d_in, d_out, ptr(W), W.stride(0), ptr(None if b is None else b.detach()), ptr(nvalid), int(K),
                 ptr(None if in_state is None else in_state.scale), ptr(None if in_state is None else in_state.shift),
                 int(in_relu), int(out_relu), ptr(y), d_out, ptr(stat))
    st = None
    if bn is not None and R > 0 and FUSE_FINISH:
        # the BatchNorm finish inside the link's launch (its last-arriving workgroup): no second launch
        st = BNState(G, d_out, x.device)
        mom = 0.1 if bn.momentum is None else float(bn.momentum)
        track = bn.track_running_stats and bn.running_mean is not None
        a.fin_gamma, a.fin_beta = ptr(None if bn.weight is None else bn.weight.detach()), ptr(None if bn.bias is None else bn.bias.detach())
        a.fin_eps, a.fin_momentum = float(bn.eps), mom
        a.fin_running_mean, a.fin_running_var = ptr(bn.running_mean if track else None), ptr(bn.running_var if track else None)
        a.fin_state, a.fin_count = ptr(st.state), ptr(st.count)
        with ops._span("sn_train_linear_f32"):
            check(lib().sn_train_linear_f32(C.byref(a), stream()), "sn_train_linear_f32")
        if track and bn.num_batches_tracked is not None:
            ops._count_batch(bn, G)
        return y, st
    with ops._span("sn_train_linear_f32"):
        check(lib().sn_train_linear_f32(C.byref(a), stream()), "sn_train_linear_f32")
    if bn is not None and R <= 0:
        # no rows (an edge encoder over a batch without edges): nn.BatchNorm1d in training mode returns the empty tensor, leaves the running
        # statistics alone and counts the batch — nothing to finish (the kernel above did not run: its moment partials are zeros)
        st = BNState(G, d_out, x.device)
        st.state.zero_()
        st.count.zero_()
        if bn.track_running_stats and bn.running_mean is not None and bn.num_batches_tracked is not None:
            ops._count_batch(bn, G)
    elif bn is not None:
        st = BNState(G, d_out, x.device)
        mom = 0.1 if bn.momentum is None else float(bn.momentum)
        track = bn.track_running_stats and bn.running_mean is not None
        with ops._span("sn_train_bn_finish_f32"):
            check(lib().sn_train_bn_finish_f32(ptr(stat), nblk, G, d_out, ptr(None if bn.weight is None else bn.weight.detach()),
                                               ptr(None if bn.bias is None else bn.bias.detach()), float(bn.eps), mom,
                                               ptr(bn.running_mean if track else None), ptr(bn.running_var if track else None),
                                               ptr(st.state), ptr(st.count), stream()), "sn_train_bn_finish_f32")
        if track and bn.num_batches_tracked is not None:
            ops._count_batch(bn, G)
    return y, st


def linear_bwd(dy, R, G, W, nvalid, K, x, zo=None, coef=None, mask=None, x_state=None, x_relu=False, want_sums=False,
               want_dx=True, want_db=True, dW_acc=None, db_acc=None, dot_x=None, gx_into=None, finish_bn=None, dot_acc=None, merge=None):
    """One pass: gx (the input gradient, masked by the operand's ReLU), its column-sum partials, dW / db.  See signnet_hip.h.
    dW_acc / db_acc: accumulate the weight / bias gradient into these buffers (returned dW / db are then None).
    dot_x: also sum gx . dot_x over all rows (per-workgroup partials, left on `linear_bwd.dot_part` for the caller to add up).
    gx_into: ADD the input gradient to this buffer instead of allocating one (several Linears reading one operand).
    With dW_acc (and db_acc when there is a bias) the reductions behind the kernel are ONE launch (sn_train_post_link_f32), which can also
    take: finish_bn — the producer's BatchNorm (x_state) whose backward finish these column sums feed, its d gamma / d beta accumulated in
    place (parameters of a FlatAdam): the (a, b, c) coefficients are left on `linear_bwd.coef`; dot_acc — the buffer the eps gradient
    (dot_x) is added to (`linear_bwd.dot_part` is then None: nothing left for eps_grad to do)."""
    W = _w(W)
    d_out, d_in = W.shape
    dev = dy.device
    nblk = int(lib().sn_train_linear_bwd_blocks(R, G))
    gx = (gx_into if gx_into is not None else torch.empty(G * R, d_in, dtype=torch.float32, device=dev)) if want_dx else None
    alloc = torch.empty if R > 0 else torch.zeros        # (no rows: no launch — the partials the reductions read are zeros)
    sums = alloc(G * nblk * 2 * d_in, dtype=torch.float32, device=dev) if want_sums else None
    stride = d_out * d_in + (d_out if want_db else 0)
    dwp = alloc(G * nblk * stride, dtype=torch.float32, device=dev)
    a = _BwdArgs(R, G, ptr(nvalid), int(K), d_in, d_out, ptr(dy), dy.stride(0), ptr(zo), 0 if zo is None else zo.stride(0),
                 ptr(None if coef is None else coef[0]), ptr(None if coef is None else coef[1]), ptr(None if coef is None else coef[2]),
                 ptr(None if mask is None else mask[0]), ptr(None if mask is None else mask[1]),
                 ptr(x), x.stride(0), ptr(None if x_state is None else x_state.scale), ptr(None if x_state is None else x_state.shift),
                 int(x_relu), ptr(x_state.mean if (want_sums and x_state is not None) else None), ptr(W), W.stride(0),
                 ptr(gx), d_in, ptr(sums), ptr(dwp), int(want_db), int(gx_into is not None), ptr(dot_x),
                 0 if dot_x is None else dot_x.stride(0), None)
    if merge is not None:
        # merge = (sums_part, nblk, BNState, gamma, dgamma_acc, dbeta_acc): this link's BatchNorm-backward coefficients are merged from
        # the column-sum partials by the link itself (coef must be None); d gamma / d beta are added in place by its workgroup 0
        m_sums, m_nblk, m_st, m_gamma, m_dg, m_db = merge
        a.merge_sums, a.merge_nblk = ptr(m_sums), int(m_nblk)
        a.merge_state, a.merge_count = ptr(m_st.state), ptr(m_st.count)
        a.merge_gamma = ptr(None if m_gamma is None else m_gamma.detach())
        a.merge_dgamma, a.merge_dbeta, a.merge_accumulate = ptr(m_dg), ptr(m_db), 1
    linear_bwd.dot_part = None
    if dot_x is not None:
        linear_bwd.dot_part = alloc(G * nblk, dtype=torch.float64, device=dev)
        a.dot_part = ptr(linear_bwd.dot_part)
    nw = d_out * d_in
    linear_bwd.coef = None
    acc_w = dW_acc is not None and (not want_db or db_acc is not None)
    fuse_bn = None
    if acc_w and finish_bn is not None and want_sums and x_state is not None:
        dg, dbt = direct_grad(finish_bn.weight), direct_grad(finish_bn.bias)
        if dg is not None and dbt is not None:
            fuse_bn = (dg, dbt)
    fuse_dot = acc_w and dot_acc is not None and dot_x is not None
    in_launch = FUSE_FINISH and R > 0 and (fuse_bn is not None or fuse_dot)
    if in_launch:            # the finishes behind the link run in the link's own launch (its last-arriving workgroup)
        if fuse_bn is not None:
            cf = torch.empty(3, x_state.G, x_state.C, dtype=torch.float32, device=dev)
            a.fin_state, a.fin_count, a.fin_gamma = ptr(x_state.state), ptr(x_state.count), ptr(finish_bn.weight.detach())
            a.fin_coef, a.fin_dgamma, a.fin_dbeta, a.fin_accumulate = ptr(cf), ptr(fuse_bn[0]), ptr(fuse_bn[1]), 1
            linear_bwd.coef = cf
        if fuse_dot:
            a.fin_dot_out = ptr(dot_acc)
    with ops._span("sn_train_linear_bwd_f32"):
        check(lib().sn_train_linear_bwd_f32(C.byref(a), stream()), "sn_train_linear_bwd_f32")
    if acc_w and DEFER_DW and (in_launch or fuse_bn is None):
        # the dW / db partials — and the eps gradient's — wait for the end of the backward pass (one launch each for all links)
        _defer_reduce(dwp, 0, G * nblk, stride, nw, dW_acc)
        if want_db:
            _defer_reduce(dwp, nw, G * nblk, stride, d_out, db_acc)
        if fuse_dot:
            if not in_launch:
                _defer_dot(linear_bwd.dot_part, dot_acc)
            linear_bwd._dot_keep, linear_bwd.dot_part = linear_bwd.dot_part, None      # (eps_grad has nothing left to do)
        return gx, sums, nblk, None, None
    if acc_w:
        if in_launch:
            fuse_bn, fuse_dot = None, False        # (already done; only the dW reduction is left)
            if linear_bwd.dot_part is not None and dot_acc is not None:
                linear_bwd._dot_keep, linear_bwd.dot_part = linear_bwd.dot_part, None
        if fuse_bn is not None or fuse_dot or want_db:
            q = _PostArgs(ptr(dwp), G * nblk, stride, nw, ptr(dW_acc), d_out if want_db else 0, ptr(db_acc) if want_db else None,
                          None, 0, 0, 0, None, None, None, None, None, None, 1, None, 0, None)
            if fuse_bn is not None:
                cf = torch.empty(3, x_state.G, x_state.C, dtype=torch.float32, device=dev)
                q.sums_part, q.nblk, q.G, q.C = ptr(sums), nblk, x_state.G, x_state.C
                q.state, q.count, q.gamma = ptr(x_state.state), ptr(x_state.count), ptr(finish_bn.weight.detach())
                q.coef, q.dgamma, q.dbeta = ptr(cf), ptr(fuse_bn[0]), ptr(fuse_bn[1])
                linear_bwd.coef = cf
            if fuse_dot:
                q.dot_part, q.dot_n, q.dot_out = ptr(linear_bwd.dot_part), G * nblk, ptr(dot_acc)
            with ops._span("sn_train_post_link_f32"):
                check(lib().sn_train_post_link_f32(C.byref(q), stream()), "sn_train_post_link_f32")
            if fuse_dot:
                linear_bwd._dot_keep, linear_bwd.dot_part = linear_bwd.dot_part, None      # (kept alive until the next call)
            return gx, sums, nblk, None, None
        with ops._span("sn_train_reduce_parts_f32"):
            check(lib().sn_train_reduce_parts_f32(ptr(dwp), G * nblk, stride, nw, ptr(dW_acc), 1, stream()), "sn_train_reduce_parts_f32")
            if want_db:
                check(lib().sn_train_reduce_parts_f32(dwp.data_ptr() + 4 * nw, G * nblk, stride, d_out, ptr(db_acc), 1, stream()),
                      "sn_train_reduce_parts_f32")
        return gx, sums, nblk, None, None
    out = torch.empty(stride, dtype=torch.float32, device=dev)
    with ops._span("sn_train_reduce_parts_f32"):
        check(lib().sn_train_reduce_parts_f32(ptr(dwp), G * nblk, stride, stride, ptr(out), 0, stream()), "sn_train_reduce_parts_f32")
    dW = out[:nw].view(d_out, d_in)
    db = out[nw:] if want_db else None
    return gx, sums, nblk, dW, db


def eps_grad(eps):
    """The eps gradient of an aggregation from `linear_bwd.dot_part` (float64 partials): one launch, added straight into eps.grad
    when the optimiser owns it (returns None then), else a [1] tensor for autograd."""
    part = linear_bwd.dot_part
    if part is None:                  # the link's post launch has already added it to eps.grad
        return None
    acc = direct_grad(eps)
    out = acc if acc is not None else torch.empty(1, dtype=torch.float32, device=part.device)
    with ops._span("sn_train_dot_finish_f64"):
        check(lib().sn_train_dot_finish_f64(ptr(part), part.numel(), ptr(out), int(acc is not None), stream()), "sn_train_dot_finish_f64")
    return None if acc is not None else out


def bn_bwd_sums(dy, z, R, G, nvalid, K, st, relu):
    Cc = z.shape[-1]
    nblk = int(lib().sn_train_bn_bwd_blocks(R, G))
    sums = torch.empty(G * nblk * 2 * Cc, dtype=torch.float32, device=z.device)
    with ops._span("sn_train_bn_bwd_sums_f32"):
        check(lib().sn_train_bn_bwd_sums_f32(ptr(dy), dy.stride(0), ptr(z), z.stride(0), R, G, Cc, ptr(nvalid), int(K), ptr(st.state),
                                             int(relu), ptr(sums), stream()), "sn_train_bn_bwd_sums_f32")
    return sums, nblk


def bn_bwd_finish(sums, nblk, st, gamma, dg_acc=None, db_acc=None):
    """coef [3, G, C] of dz = a*g - b - c*z, and (d gamma, d beta) summed over the groups (accumulated into dg_acc / db_acc when
    both are given: None is returned for them then)."""
    coef = torch.empty(3, st.G, st.C, dtype=torch.float32, device=sums.device)
    acc = dg_acc is not None and db_acc is not None
    dgb = None if acc else torch.empty(2, st.C, dtype=torch.float32, device=sums.device)
    with ops._span("sn_train_bn_bwd_finish_f32"):
        check(lib().sn_train_bn_bwd_finish_f32(ptr(sums), nblk, st.G, st.C, ptr(st.state), ptr(st.count),
                                               ptr(None if gamma is None else gamma.detach()), ptr(coef),
                                               ptr(dg_acc if acc else dgb[0]), ptr(db_acc if acc else dgb[1]), int(acc),
                                               stream()), "sn_train_bn_bwd_finish_f32")
    return (coef, None, None) if acc else (coef, dgb[0], dgb[1])


def bn_bwd(dy, z, R, G, nvalid, K, st, relu, gamma, dg_acc=None, db_acc=None):
    """bn_bwd_sums + bn_bwd_finish; ONE launch (the sums kernel's last-arriving workgroup finishes) for widths up to 128."""
    Cc = z.shape[-1]
    if not (FUSE_FINISH and R > 0 and Cc <= 128):
        sums, nblk = bn_bwd_sums(dy, z, R, G, nvalid, K, st, relu)
        return bn_bwd_finish(sums, nblk, st, gamma, dg_acc, db_acc)
    nblk = int(lib().sn_train_bn_bwd_blocks(R, G))
    sums = torch.empty(G * nblk * 2 * Cc, dtype=torch.float32, device=z.device)
    coef = torch.empty(3, st.G, st.C, dtype=torch.float32, device=z.device)
    acc = dg_acc is not None and db_acc is not None
    dgb = None if acc else torch.empty(2, st.C, dtype=torch.float32, device=z.device)
    with ops._span("sn_train_bn_bwd_f32"):
        check(lib().sn_train_bn_bwd_f32(ptr(dy), dy.stride(0), ptr(z), z.stride(0), R, G, Cc, ptr(nvalid), int(K), ptr(st.state),
                                        ptr(st.count), int(relu), ptr(None if gamma is None else gamma.detach()), ptr(sums), ptr(coef),
                                        ptr(dg_acc if acc else dgb[0]), ptr(db_acc if acc else dgb[1]), int(acc), stream()),
              "sn_train_bn_bwd_f32")
    return (coef, None, None) if acc else (coef, dgb[0], dgb[1])


def bn_apply(z, R, G, nvalid, K, st, relu, residual):
    """y = [relu](z * scale[g] + shift[g]) [+ residual] on valid rows, 0 elsewhere."""
    y = torch.empty_like(z)
    Cc = z.shape[-1]
    with ops._span("sn_train_bn_apply_f32"):
        check(lib().sn_train_bn_apply_f32(ptr(z), z.stride(0), R, G, Cc, ptr(nvalid), int(K), ptr(st.state), int(relu),
                                          ptr(residual), 0 if residual is None else residual.stride(0), ptr(y), Cc, stream()),
              "sn_train_bn_apply_f32")
    return y


# ----------------------------------------------------------------------------- Functions
def _mlp2_forward(x, R, G, lin1, bn1, lin2, bn2, nvalid, K, relu_out, res):
    W1, b1, W2, b2 = lin1.weight, lin1.bias, lin2.weight, lin2.bias
    z1, st1 = linear_fwd(x, R, G, W1, b1, nvalid, K, bn=bn1)
    z2, st2 = linear_fwd(z1, R, G, W2, b2, nvalid, K, in_state=st1, in_relu=True, bn=bn2)
    y = bn_apply(z2, R, G, nvalid, K, st2, relu_out, res) if bn2 is not None else z2
    return y, z1, z2, st1, st2


def _can_merge(bn, R, C):
    """The consumer-side merge needs the BatchNorm's affine gradients accumulated in place (a FlatAdam's parameters) and rows to walk."""
    return (MERGE_COEF and R > 0 and C <= 128 and bn is not None and bn.weight is not None and
            direct_grad(bn.weight) is not None and direct_grad(bn.bias) is not None)


def _mlp2_backward(dy, x, z1, z2, st1, st2, R, G, lin1, bn1, lin2, bn2, nvalid, K, relu_out, want_dx, dot_x=None, dot_acc=None):
    """-> (dx, dW1, db1, dg1, dbe1, dW2, db2, dg2, dbe2); parameter gradients already accumulated in-kernel come back as None."""
    dg2 = dbe2 = coef2 = mask2 = merge2 = None
    if st2 is not None:
        if _can_merge(bn2, R, z2.shape[-1]):
            sums2, nb2 = bn_bwd_sums(dy, z2, R, G, nvalid, K, st2, relu_out)
            merge2 = (sums2, nb2, st2, bn2.weight, direct_grad(bn2.weight), direct_grad(bn2.bias))
        else:
            coef2, dg2, dbe2 = bn_bwd(dy, z2, R, G, nvalid, K, st2, relu_out, bn2.weight, direct_grad(bn2.weight), direct_grad(bn2.bias))
        mask2 = (st2.scale, st2.shift) if relu_out else None
    merge1_ok = _can_merge(bn1, R, z1.shape[-1]) and direct_grad(lin2.weight) is not None and (lin2.bias is None or direct_grad(lin2.bias) is not None)
    gz1, sums1, nb1, dW2, db2 = linear_bwd(dy, R, G, lin2.weight, nvalid, K, z1, zo=z2 if st2 is not None else None, coef=coef2, mask=mask2,
                                           x_state=st1, x_relu=True, want_sums=True, want_db=lin2.bias is not None,
                                           dW_acc=direct_grad(lin2.weight), db_acc=direct_grad(lin2.bias),
                                           finish_bn=None if merge1_ok else (bn1 if bn1.weight is not None else None), merge=merge2)
    coef1 = dg1 = dbe1 = merge1 = None
    if merge1_ok:                             # lin1's own launch merges sums1 (no finish in front of it)
        merge1 = (sums1, nb1, st1, bn1.weight, direct_grad(bn1.weight), direct_grad(bn1.bias))
    elif linear_bwd.coef is not None:         # the finish ran inside the link's post launch (d gamma / d beta accumulated in place)
        coef1 = linear_bwd.coef
    else:
        coef1, dg1, dbe1 = bn_bwd_finish(sums1, nb1, st1, bn1.weight, direct_grad(bn1.weight), direct_grad(bn1.bias))
    dx, _, _, dW1, db1 = linear_bwd(gz1, R, G, lin1.weight, nvalid, K, x, zo=z1, coef=coef1, want_dx=want_dx, want_db=lin1.bias is not None,
                                    dW_acc=direct_grad(lin1.weight), db_acc=direct_grad(lin1.bias), dot_x=dot_x, dot_acc=dot_acc, merge=merge1)
    if bn1.weight is None:
        dg1 = dbe1 = None
    if bn2 is None or bn2.weight is None:
        dg2 = dbe2 = None
    return dx, dW1, db1, dg1, dbe1, dW2, db2, dg2, dbe2


class _Mlp2Bn(Function):
    @staticmethod
    def forward(ctx, x, W1, b1, g1, be1, W2, b2, g2, be2, residual, lin1, bn1, lin2, bn2, nvalid, K, G, relu_out):
        x = _c(x)
        R = x.shape[0] // G
        res = None if residual is None else _c(residual)
        y, z1, z2, st1, st2 = _mlp2_forward(x, R, G, lin1, bn1, lin2, bn2, nvalid, K, relu_out, res)
        ctx.save_for_backward(x, z1, z2)
        ctx.meta = (R, G, nvalid, K, relu_out, st1, st2, residual is not None, lin1, bn1, lin2, bn2)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, z1, z2 = ctx.saved_tensors
        R, G, nvalid, K, relu_out, st1, st2, has_res, lin1, bn1, lin2, bn2 = ctx.meta
        dy = _c(dy)
        grads = _mlp2_backward(dy, x, z1, z2, st1, st2, R, G, lin1, bn1, lin2, bn2, nvalid, K, relu_out, ctx.needs_input_grad[0])
        return grads + (dy if has_res else None, None, None, None, None, None, None, None, None)


class _LinBn(Function):
    """[relu](bn(lin(x))) [+ residual]: one Linear -> train-mode BatchNorm link (rho's output projection, the read-out's first layer)."""

    @staticmethod
    def forward(ctx, x, W, b, gam, bet, residual, lin, bn, nvalid, K, relu):
        x = _c(x)
        R = x.shape[0]
        res = None if residual is None else _c(residual)
        z, st = linear_fwd(x, R, 1, W, b, nvalid, K, bn=bn)
        y = bn_apply(z, R, 1, nvalid, K, st, relu, res)
        ctx.save_for_backward(x, z)
        ctx.meta = (R, nvalid, K, relu, st, residual is not None, lin, bn)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, z = ctx.saved_tensors
        R, nvalid, K, relu, st, has_res, lin, bn = ctx.meta
        dy = _c(dy)
        coef = dg = dbe = merge = None
        if _can_merge(bn, R, z.shape[-1]):
            sums, nb = bn_bwd_sums(dy, z, R, 1, nvalid, K, st, relu)
            merge = (sums, nb, st, bn.weight, direct_grad(bn.weight), direct_grad(bn.bias))
        else:
            coef, dg, dbe = bn_bwd(dy, z, R, 1, nvalid, K, st, relu, bn.weight, direct_grad(bn.weight), direct_grad(bn.bias))
        dx, _, _, dW, db = linear_bwd(dy, R, 1, lin.weight, nvalid, K, x, zo=z, coef=coef, mask=(st.scale, st.shift) if relu else None,
                                      want_dx=ctx.needs_input_grad[0], want_db=lin.bias is not None,
                                      dW_acc=direct_grad(lin.weight), db_acc=direct_grad(lin.bias), merge=merge)
        if bn.weight is None:
            dg = dbe = None
        return dx, dW, db, dg, dbe, (dy if has_res else None), None, None, None, None, None


def lin_bn(x, lin, bn, nvalid=None, K=0, relu=True, residual=None):
    return _LinBn.apply(x, lin.weight, lin.bias, bn.weight, bn.bias, residual, lin, bn, nvalid, K, relu)


class _GinLayer(Function):
    """x -> a = (1+eps) x + sum_nbr x -> mlp2_bn(a) + x: one GNN3d layer (sign_net.py:36-43) for G stacked sign passes.  The adjoint of
    the aggregation takes the residual's gradient in the same pass, and the eps gradient comes out of the first Linear's backward."""

    @staticmethod
    def forward(ctx, x, eps, W1, b1, g1, be1, W2, b2, g2, be2, lin1, bn1, lin2, bn2, plan, rplan, nvalid, K, G):
        x = _c(x)
        d = x.shape[1]
        R = x.shape[0] // G
        a = ops.gin_aggregate(x.view(-1, K * d), plan, eps.detach()).view(-1, d)
        y, z1, z2, st1, st2 = _mlp2_forward(a, R, G, lin1, bn1, lin2, bn2, nvalid, K, True, x)
        ctx.save_for_backward(x, a, z1, z2, eps)
        ctx.eps_param = eps
        ctx.meta = (R, G, nvalid, K, st1, st2, lin1, bn1, lin2, bn2, rplan)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, a, z1, z2, eps = ctx.saved_tensors
        R, G, nvalid, K, st1, st2, lin1, bn1, lin2, bn2, rplan = ctx.meta
        dy = _c(dy)
        d = x.shape[1]
        want_eps = ctx.needs_input_grad[1]
        grads = _mlp2_backward(dy, a, z1, z2, st1, st2, R, G, lin1, bn1, lin2, bn2, nvalid, K, True, True, dot_x=x if want_eps else None,
                               dot_acc=direct_grad(ctx.eps_param) if want_eps else None)
        da = grads[0]
        deps = eps_grad(ctx.eps_param) if want_eps else None
        dx = torch.empty_like(x)
        with ops._span("sn_gin_aggregate_add_f32"):
            check(lib().sn_gin_aggregate_add_f32(ptr(da), ptr(dy), ptr(dx), x.shape[0] // K, K * d, ptr(rplan.rowptr), ptr(rplan.col),
                                                 ptr(eps.detach()), stream()), "sn_gin_aggregate_add_f32")
        return (dx, deps) + grads[1:] + (None,) * 9


def gin_layer(x, conv_eps, lin1, bn1, lin2, bn2, plan, rplan, nvalid, K, G=1):
    """One phi layer (l >= 1) on [G*N*K, d] rows: relu(bn2(lin2(relu(bn1(lin1(GIN-aggregate(x))))))) + x."""
    return _GinLayer.apply(x, conv_eps, lin1.weight, lin1.bias, bn1.weight, bn1.bias, lin2.weight, lin2.bias, bn2.weight, bn2.bias,
                           lin1, bn1, lin2, bn2, plan, rplan, nvalid, K, G)


class _SignSum(Function):
    """phi(x) + phi(-x) of the two stacked sign passes [2*M, d] -> [M, d] (sign_net.py:113).  The adjoint hands the SAME gradient to
    both passes: one duplicating copy instead of what autograd builds for `x[:M] + x[M:]` (two zero-filled [2M, d] buffers, two slice
    copies and an add)."""

    @staticmethod
    def forward(ctx, x2, nvalid, K):
        x2 = _c(x2)
        M = x2.shape[0] // 2
        ctx.meta = (M,)
        return ops.masked_affine(x2[:M], nvalid, K, residual=x2[M:])

    @staticmethod
    def backward(ctx, dy):
        return _c(dy).repeat(2, 1), None, None           # (dy is zero on invalid rows: the consumer masks)


def sign_sum(x2, nvalid, K):
    return _SignSum.apply(x2, nvalid, K)


class _GineLayer(Function):
    """h, e -> u = (1+eps) h + sum relu(h_j + e_ji) -> mlp2_bn(u) + h: one layer of GNN.forward (model.py:52-60, pyg_gnn_wrapper.py:19-28).
    lidx >= 0: `e` is the [L, E, C] block of autograd.embedding_sum_layers and this layer reads plane lidx; its adjoint writes the
    plane's gradient into the block's shared buffer (e._sn_gbuf) and only layer 0 — the last to run backward — hands the buffer to
    autograd, so the L edge encoders' table gradients are ONE launch pair (no slice / zero-fill / add per layer)."""

    @staticmethod
    def forward(ctx, h, e, eps, W1, b1, g1, be1, W2, b2, g2, be2, lin1, bn1, lin2, bn2, plan, rplan, lidx, gbuf):
        h = _c(h)
        ev = _c(e) if lidx < 0 else e[lidx]
        u = ops.gine_aggregate(h, ev, plan, eps.detach())
        y, z1, z2, st1, st2 = _mlp2_forward(u, h.shape[0], 1, lin1, bn1, lin2, bn2, None, 0, True, h)
        ctx.save_for_backward(h, ev, u, z1, z2, eps)
        ctx.eps_param = eps
        ctx.meta = (st1, st2, lin1, bn1, lin2, bn2, rplan, lidx, gbuf)
        return y

    @staticmethod
    def backward(ctx, dy):
        h, e, u, z1, z2, eps = ctx.saved_tensors
        st1, st2, lin1, bn1, lin2, bn2, rplan, lidx, gbuf = ctx.meta
        dy = _c(dy)
        N, d = h.shape
        want_eps = ctx.needs_input_grad[2]
        grads = _mlp2_backward(dy, u, z1, z2, st1, st2, N, 1, lin1, bn1, lin2, bn2, None, 0, True, True, dot_x=h if want_eps else None,
                               dot_acc=direct_grad(ctx.eps_param) if want_eps else None)
        du = grads[0]
        deps = eps_grad(ctx.eps_param) if want_eps else None
        dh = torch.empty_like(h)
        dee = torch.empty_like(e) if lidx < 0 else gbuf[lidx]
        with ops._span("sn_gine_aggregate_bwd_add_f32"):
            check(lib().sn_gine_aggregate_bwd_add_f32(ptr(h), ptr(e), ptr(du), ptr(dy), N, d, ptr(rplan.rowptr), ptr(rplan.col),
                                                      ptr(rplan.eperm), ptr(eps.detach()), ptr(dh), ptr(dee), stream()),
                  "sn_gine_aggregate_bwd_add_f32")
        if lidx >= 0:
            dee = gbuf if lidx == 0 else None
        return (dh, dee, deps) + grads[1:] + (None,) * 8


def gine_layer(h, e, conv_eps, lin1, bn1, lin2, bn2, plan, rplan, layer=-1):
    """layer >= 0: e is the [L, E, C] block of autograd.embedding_sum_layers (see _GineLayer)."""
    gbuf = getattr(e, "_sn_gbuf", None) if layer >= 0 else None
    if layer >= 0 and gbuf is None:
        raise ValueError("gine_layer: a layer index needs the block of embedding_sum_layers")
    return _GineLayer.apply(h, e, conv_eps, lin1.weight, lin1.bias, bn1.weight, bn1.bias, lin2.weight, lin2.bias, bn2.weight, bn2.bias,
                            lin1, bn1, lin2, bn2, plan, rplan, layer, gbuf)


def mlp2_bn(x, lin1, bn1, lin2, bn2, nvalid=None, K=0, G=1, residual=None, relu_out=True):
    """[relu](bn2(lin2(relu(bn1(lin1(x)))))) [+ residual]; bn2 None: the second Linear's output as is (no ReLU, no residual).
    Gradients flowing in must be zero on invalid rows (the convention of autograd.py: every producer masks them)."""
    if bn2 is None and (residual is not None):
        raise ValueError("mlp2_bn: a residual needs the second BatchNorm")
    return _Mlp2Bn.apply(x, lin1.weight, lin1.bias, bn1.weight, bn1.bias, lin2.weight, lin2.bias,
                         None if bn2 is None else bn2.weight, None if bn2 is None else bn2.bias, residual, lin1, bn1, lin2, bn2,
                         nvalid, K, G, relu_out and bn2 is not None)


class _Linear(Function):
    @staticmethod
    def forward(ctx, x, W, b, nvalid, K, relu, owner):
        x = _c(x)
        R = x.shape[0]
        y, _ = linear_fwd(x, R, 1, W, b, nvalid, K, out_relu=relu)
        ctx.save_for_backward(x, W, y if relu else None)
        ctx.meta = (R, nvalid, K, relu, b is not None, owner)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W, y = ctx.saved_tensors
        R, nvalid, K, relu, has_b, owner = ctx.meta
        dy = _c(dy)
        mask = None
        if relu:           # g = dy * [y > 0]: the mask mechanism with scale 1, shift 0 on the saved output
            one = torch.ones(1, y.shape[1], dtype=torch.float32, device=y.device)
            mask = (one, torch.zeros_like(one))
        Wp, bp = (owner.weight, owner.bias) if owner is not None else (None, None)
        dx, _, _, dW, db = linear_bwd(dy, R, 1, W, nvalid, K, x, zo=y if relu else None, mask=mask,
                                      want_dx=ctx.needs_input_grad[0], want_db=has_b, dW_acc=direct_grad(Wp), db_acc=direct_grad(bp))
        return dx, dW, db, None, None, None, None


class _QKV(Function):
    """q, k, v = x Wq^T, x Wk^T, x Wv^T (MultiHeadAttention's three bias-free projections of one operand, transformer_module.py:84-86):
    the adjoint ADDS the three input gradients in the kernels' epilogues (gx +=) instead of two elementwise launches."""

    @staticmethod
    def forward(ctx, x, Wq, Wk, Wv, mq, mk, mv, nvalid, K):
        x = _c(x)
        R = x.shape[0]
        outs = tuple(linear_fwd(x, R, 1, m.weight, None, nvalid, K)[0] for m in (mq, mk, mv))
        ctx.save_for_backward(x)
        ctx.meta = (R, nvalid, K, mq, mk, mv)
        return outs

    @staticmethod
    def backward(ctx, dq, dk, dv):
        (x,) = ctx.saved_tensors
        R, nvalid, K, mq, mk, mv = ctx.meta
        gx, dWs = None, []
        for i, (g, m) in enumerate(((dq, mq), (dk, mk), (dv, mv))):
            r = linear_bwd(_c(g), R, 1, m.weight, nvalid, K, x, want_dx=ctx.needs_input_grad[0], want_db=False,
                           dW_acc=direct_grad(m.weight), gx_into=gx)
            gx = r[0]
            dWs.append(r[3])
        return (gx, *dWs, None, None, None, None, None)


def qkv(x, mq, mk, mv, nvalid=None, K=0):
    if any(m.bias is not None for m in (mq, mk, mv)):
        raise ValueError("qkv: the attention projections have no bias")
    return _QKV.apply(x, mq.weight, mk.weight, mv.weight, mq, mk, mv, nvalid, K)


def linear(x, W, b=None, nvalid=None, K=0, relu=False, owner=None):
    """y = [relu](x W^T + b) on valid rows, 0 elsewhere.  owner: the nn.Linear holding W / b (enables in-kernel gradient accumulation)."""
    return _Linear.apply(x, W, b, nvalid, K, relu, owner)


def linear_module(x, lin, nvalid=None, K=0, relu=False):
    return _Linear.apply(x, lin.weight, lin.bias, nvalid, K, relu, lin)


# ----------------------------------------------------------------------------- the 1 -> 1 -> d MaskedMLP on a scalar input, closed form
def _bn_run(bn):
    track = bn.track_running_stats and bn.running_mean is not None
    mom = 0.1 if bn.momentum is None else float(bn.momentum)
    return mom, (bn.running_mean if track else None), (bn.running_var if track else None), track


def _smlp_args(a, M, G, negate, nvalid, K, lin1, bn1, lin2, bn2, relu_b, sst, cst):
    d = lin2.weight.shape[0]
    det = lambda t: None if t is None else t.detach()
    return _SMlpArgs(ptr(a), M, G, int(negate), ptr(nvalid), int(K), d, ptr(det(lin1.weight)), ptr(det(bn1.weight)), ptr(det(bn1.bias)),
                     float(bn1.eps), ptr(det(lin2.weight)), ptr(det(lin2.bias)), ptr(det(bn2.weight)), ptr(det(bn2.bias)), float(bn2.eps),
                     int(relu_b), ptr(sst), ptr(cst))


def scalar_mlp_stats(a, lin1, bn1, lin2, bn2, nvalid, K, G=1, negate_second=False):
    """Only the side effects of the module call (running statistics of both BatchNorms) and the closed-form batch state:
    GINESignNetPyG evaluates eigen_encoder2 and discards the value (core/sign_net.py:111-112)."""
    a = _c(a.detach()).view(-1)
    M, d = a.numel(), lin2.weight.shape[0]
    sst = torch.empty(G, 8, dtype=torch.float64, device=a.device)
    cst = torch.empty(G, 2, d, dtype=torch.float32, device=a.device)
    args = _smlp_args(a, M, G, negate_second, nvalid, K, lin1, bn1, lin2, bn2, True, sst, cst)
    ma, rma, rva, ta = _bn_run(bn1)
    mb, rmb, rvb, tb = _bn_run(bn2)
    work = torch.empty(int(lib().sn_train_scalar_mlp_work_doubles(M, G, d)), dtype=torch.float64, device=a.device)
    with ops._span("sn_train_scalar_mlp_stats_f32"):
        check(lib().sn_train_scalar_mlp_stats_f32(C.byref(args), ma, ptr(rma), ptr(rva), mb, ptr(rmb), ptr(rvb), ptr(work), stream()),
              "sn_train_scalar_mlp_stats_f32")
    for bn, t in ((bn1, ta), (bn2, tb)):
        if t and bn.num_batches_tracked is not None:
            ops._count_batch(bn, G)
    return a, sst, cst


class _ScalarMlp(Function):
    @staticmethod
    def forward(ctx, a, w1, ga, ba, w2, b2, gb, bb, lin1, bn1, lin2, bn2, nvalid, K, G, negate_second):
        a_, sst, cst = scalar_mlp_stats(a, lin1, bn1, lin2, bn2, nvalid, K, G, negate_second)
        M, d = a_.numel(), lin2.weight.shape[0]
        y = torch.empty(G * M, d, dtype=torch.float32, device=a.device)
        args = _smlp_args(a_, M, G, negate_second, nvalid, K, lin1, bn1, lin2, bn2, True, sst, cst)
        with ops._span("sn_train_scalar_mlp_apply_f32"):
            check(lib().sn_train_scalar_mlp_apply_f32(C.byref(args), ptr(y), stream()), "sn_train_scalar_mlp_apply_f32")
        ctx.save_for_backward(a_, sst, cst)
        ctx.meta = (lin1, bn1, lin2, bn2, nvalid, K, G, negate_second, a.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        a_, sst, cst = ctx.saved_tensors
        lin1, bn1, lin2, bn2, nvalid, K, G, negate_second, ashape = ctx.meta
        dy = _c(dy)
        M, d = a_.numel(), lin2.weight.shape[0]
        dev = dy.device
        args = _smlp_args(a_, M, G, negate_second, nvalid, K, lin1, bn1, lin2, bn2, True, sst, cst)
        nblk = int(lib().sn_train_bn_bwd_blocks(M, G))
        part = torch.empty(G * nblk * 2 * d, dtype=torch.float32, device=dev)
        trow = torch.empty(G * M, dtype=torch.float32, device=dev)
        da = torch.empty(M, dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        targets = [direct_grad(p) for p in (lin1.weight, bn1.weight, bn1.bias, lin2.weight, bn2.weight, bn2.bias)]
        acc = all(t is not None for t in targets)
        if not acc:
            small = torch.empty(3 + 3 * d, dtype=torch.float32, device=dev)
            targets = [small[0:1], small[1:2], small[2:3], small[3:3 + d], small[3 + d:3 + 2 * d], small[3 + 2 * d:]]
        work = torch.empty(int(lib().sn_train_scalar_mlp_work_doubles(M, G, d)), dtype=torch.float64, device=dev)
        with ops._span("sn_train_scalar_mlp_bwd_f32"):
            check(lib().sn_train_scalar_mlp_bwd_f32(C.byref(args), ptr(dy), ptr(part), ptr(trow), *[ptr(t) for t in targets], ptr(da),
                                                    int(acc), ptr(work), stream()), "sn_train_scalar_mlp_bwd_f32")
        if acc:
            gw1 = gga = gba = gw2 = ggb = gbb = None
        else:
            gw1, gga, gba = targets[0].view_as(lin1.weight), targets[1], targets[2]
            gw2, ggb, gbb = targets[3].view_as(lin2.weight), targets[4], targets[5]
        gb2 = None if lin2.bias is None else torch.zeros_like(lin2.bias)      # in front of a batch-statistics BatchNorm: exactly 0
        return (None if da is None else da.view(ashape), gw1, gga, gba, gw2, gb2, ggb, gbb,
                None, None, None, None, None, None, None, None)


def scalar_mlp(a, lin1, bn1, lin2, bn2, nvalid, K, G=1, negate_second=False):
    """relu(bn2(lin2(relu(bn1(lin1(a)))))) for a scalar input a [M] with lin1 = Linear(1, 1, bias=False): -> [G*M, d]; G = 2 with
    negate_second evaluates a and -a (the two sign passes) with shared weights and separate batch statistics."""
    if lin1.weight.numel() != 1 or lin1.bias is not None or lin2.weight.shape[1] != 1:
        raise ValueError("scalar_mlp: needs Linear(1, 1, bias=False) -> Linear(1, d)")
    return _ScalarMlp.apply(a, lin1.weight, bn1.weight, bn1.bias, lin2.weight, lin2.bias, bn2.weight, bn2.bias,
                            lin1, bn1, lin2, bn2, nvalid, K, G, negate_second)
