"""torch.autograd.Function wrappers: forward = the layer-at-a-time HIP entry points, backward = their hand-written
adjoints (csrc/backward.hip).  SURVEY.md §8 f1 — what the reference gets from torch.autograd over ATen / PyG /
torch_scatter kernels (loss.backward(), Alchemy/main_alchemy.py:108, GINESignNetPyG/core/train.py:62-63).

torch only records the graph and owns the buffers; no arithmetic of the path runs in ATen except autograd's own gradient
accumulation (the adds where a tensor feeds several ops, and the += into .grad).  Conventions as in ops.py:
row matrices [R, C] with R = N*K slot rows (valid iff slot < nvalid[node]) or R = N / E / B plain rows (nvalid None).
Gradients flowing into an op are zero on invalid rows because every producer masks them.
"""
from __future__ import annotations

import ctypes as C

import torch
from torch.autograd import Function

from . import ops
from ._lib import check, lib, ptr, stream


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


# ----------------------------------------------------------------------------- raw adjoint launches
def linear_wgrad(x, dy, nvalid, K, want_bias=True):
    d_in, d_out = x.shape[-1], dy.shape[-1]
    R = x.numel() // d_in
    buf = torch.empty(d_out * d_in + d_out, dtype=torch.float32, device=x.device)     # dW | db: one reduction produces both
    dW = buf[:d_out * d_in].view(d_out, d_in)
    db = buf[d_out * d_in:]
    scratch = torch.empty(int(lib().sn_linear_wgrad_scratch_floats(R, d_in, d_out)), dtype=torch.float32, device=x.device)
    with ops._span("sn_linear_wgrad_f32"):
        check(lib().sn_linear_wgrad_f32(ptr(x), d_in, ptr(dy), d_out, R, d_in, d_out, ptr(nvalid), int(K), ptr(dW), ptr(db),
                                        ptr(scratch), stream()), "sn_linear_wgrad_f32")
    return dW, (db if want_bias else None)


def relu_bwd(y, dy, nvalid, K):
    Cc = y.shape[-1]
    dx = torch.empty_like(y)
    check(lib().sn_relu_bwd_f32(ptr(y), ptr(dy), y.numel() // Cc, Cc, ptr(nvalid), int(K), ptr(dx), stream()), "sn_relu_bwd_f32")
    return dx


def dot(a, b):
    out = torch.empty(1, dtype=torch.float32, device=a.device)
    scratch = torch.empty(256, dtype=torch.float32, device=a.device)
    check(lib().sn_dot_f32(ptr(a), ptr(b), a.numel(), ptr(out), ptr(scratch), stream()), "sn_dot_f32")
    return out


# ----------------------------------------------------------------------------- Linear (+ mask, optional ReLU)
def _linear_adjoint(x, W, dy, nvalid, K, has_b, want_dx):
    dx = None
    if want_dx:
        plt = ops.PackedLinear(ops.pack_weight_t(W.detach()), W.shape[1], W.shape[0], None)      # W^T packed in place
        dx = ops.masked_linear(dy, plt, nvalid, K, use_bias=False).view(x.shape)
    dW, db = linear_wgrad(x, dy, nvalid, K, has_b)
    return dx, dW, db


def _bn_act_adjoint(z, dy, mean, rstd, scale, shift, count, nvalid, K, relu):
    Cc = z.shape[-1]
    R = z.numel() // Cc
    sums = torch.empty(2 * Cc, dtype=torch.float32, device=z.device)
    dz = torch.empty_like(z)
    scratch = torch.empty(int(lib().sn_bn_act_bwd_scratch_floats(R, Cc)), dtype=torch.float32, device=z.device)
    with ops._span("sn_bn_act_bwd_f32"):
        check(lib().sn_bn_act_bwd_f32(ptr(z), Cc, ptr(dy), Cc, R, Cc, ptr(nvalid), int(K), ptr(mean), ptr(rstd), ptr(scale),
                                      ptr(shift), int(relu), ptr(count), ptr(sums), ptr(dz), Cc, ptr(scratch), stream()),
              "sn_bn_act_bwd_f32")
    return dz, sums


class _Linear(Function):
    @staticmethod
    def forward(ctx, x, W, b, nvalid, K, relu):
        x = _c(x)
        pl = ops.PackedLinear(ops.pack_weight(W.detach()), W.shape[0], W.shape[1], None if b is None else _c(b.detach()))
        y = ops.masked_linear(x, pl, nvalid, K, relu=relu)
        ctx.save_for_backward(x, W, y if relu else None)
        ctx.meta = (nvalid, K, relu, b is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W, y = ctx.saved_tensors
        nvalid, K, relu, has_b = ctx.meta
        dy = _c(dy)
        if relu:
            dy = relu_bwd(y, dy, nvalid, K)
        dx, dW, db = _linear_adjoint(x, W, dy, nvalid, K, has_b, ctx.needs_input_grad[0])
        return dx, dW, db, None, None, None


def linear(x, W, b=None, nvalid=None, K=0, relu=False):
    """y = [relu](x @ W^T + b) on valid rows, 0 elsewhere (nn.Linear + the reference's mask)."""
    if W.stride(-1) != 1 or (W.shape[0] > 1 and W.stride(0) < W.shape[1]):
        W = W.contiguous()                       # a transposed / strided view of a parameter: autograd routes the gradient back
    return _Linear.apply(x, W, b, nvalid, K, relu)


# ----------------------------------------------------------------------------- train-mode BatchNorm (+ReLU) (+residual)
class _BnAct(Function):
    @staticmethod
    def forward(ctx, z, gamma, beta, residual, bn, nvalid, K, relu):
        z = _c(z)
        mean, _, rstd, scale, shift, count = ops.bn_train_stats(z, bn, nvalid, K)      # gamma / beta are bn.weight / bn.bias
        res = None if residual is None else _c(residual)
        y = ops.masked_affine(z, nvalid, K, scale=scale, shift=shift, relu=relu, residual=res)
        ctx.save_for_backward(z, mean, rstd, scale, shift, count)
        ctx.meta = (nvalid, K, relu, gamma is not None, residual is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        z, mean, rstd, scale, shift, count = ctx.saved_tensors
        nvalid, K, relu, affine, has_res = ctx.meta
        dy = _c(dy)
        Cc = z.shape[-1]
        dz, sums = _bn_act_adjoint(z, dy, mean, rstd, scale, shift, count, nvalid, K, relu)
        dres = None
        if has_res:
            dres = dy if nvalid is None else ops.masked_affine(dy, nvalid, K)      # the output is 0 on invalid rows
        return dz, sums[Cc:] if affine else None, sums[:Cc] if affine else None, dres, None, None, None, None


class _LinearBnAct(Function):
    """[relu](BatchNorm1d_train(x @ W^T + b)) [+ residual] — the composition of _Linear and _BnAct with the Linear and the batch
    statistics in ONE forward launch (sn_linear_bn_train_f32); the backward is the two adjoints back to back."""

    @staticmethod
    def forward(ctx, x, W, b, gamma, beta, residual, bn, nvalid, K, relu):
        x = _c(x)
        pl = ops.PackedLinear(ops.pack_weight(W.detach()), W.shape[0], W.shape[1], None if b is None else _c(b.detach()))
        z, mean, _, rstd, scale, shift, count = ops.linear_bn_train(x, pl, bn, nvalid, K)
        res = None if residual is None else _c(residual)
        y = ops.masked_affine(z, nvalid, K, scale=scale, shift=shift, relu=relu, residual=res)
        ctx.save_for_backward(x, W, z, mean, rstd, scale, shift, count)
        ctx.meta = (nvalid, K, relu, b is not None, gamma is not None, residual is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W, z, mean, rstd, scale, shift, count = ctx.saved_tensors
        nvalid, K, relu, has_b, affine, has_res = ctx.meta
        dy = _c(dy)
        Cc = z.shape[-1]
        dz, sums = _bn_act_adjoint(z, dy, mean, rstd, scale, shift, count, nvalid, K, relu)
        dx, dW, db = _linear_adjoint(x, W, dz, nvalid, K, has_b, ctx.needs_input_grad[0])
        dres = None
        if has_res:
            dres = dy if nvalid is None else ops.masked_affine(dy, nvalid, K)      # the output is 0 on invalid rows
        return dx, dW, db, sums[Cc:] if affine else None, sums[:Cc] if affine else None, dres, None, None, None, None


def linear_bn_act(x, W, b, bn, nvalid=None, K=0, relu=True, residual=None):
    if W.stride(-1) != 1 or (W.shape[0] > 1 and W.stride(0) < W.shape[1]):
        W = W.contiguous()
    return _LinearBnAct.apply(x, W, b, bn.weight, bn.bias, residual, bn, nvalid, K, relu)


def bn_act(z, bn, nvalid=None, K=0, relu=True, residual=None):
    """[relu](BatchNorm1d(z) with batch statistics over the valid rows) [+ residual]; updates bn's running statistics."""
    return _BnAct.apply(z, bn.weight, bn.bias, residual, bn, nvalid, K, relu)


# ----------------------------------------------------------------------------- masked add (phi(x) + phi(-x), residuals)
class _MaskedAdd(Function):
    @staticmethod
    def forward(ctx, a, b, nvalid, K):
        ctx.meta = (nvalid, K)
        return ops.masked_affine(_c(a), nvalid, K, residual=_c(b))

    @staticmethod
    def backward(ctx, dy):
        nvalid, K = ctx.meta
        g = _c(dy) if nvalid is None else ops.masked_affine(_c(dy), nvalid, K)
        return g, g, None, None


def masked_add(a, b, nvalid=None, K=0):
    return _MaskedAdd.apply(a, b, nvalid, K)


# ----------------------------------------------------------------------------- GIN / GINE aggregation
class _GinAgg(Function):
    @staticmethod
    def forward(ctx, x, eps, plan, rplan, negate):
        x = _c(x)
        ctx.save_for_backward(x, eps)
        ctx.meta = (rplan, negate)
        return ops.gin_aggregate(x, plan, eps.detach(), negate=negate)

    @staticmethod
    def backward(ctx, g):
        x, eps = ctx.saved_tensors
        rplan, negate = ctx.meta
        g = _c(g)
        dx = ops.gin_aggregate(g, rplan, eps.detach(), negate=negate) if ctx.needs_input_grad[0] else None
        deps = None
        if ctx.needs_input_grad[1]:
            deps = dot(g, x)
            if negate:
                deps = -deps
        return dx, deps, None, None, None


def gin_aggregate(x, eps, plan, rplan, negate=False):
    """(1+eps) x_i + sum_{j->i} x_j over the node axis of x [N, F] (F = K*C flattened), optionally of -x."""
    return _GinAgg.apply(x, eps, plan, rplan, negate)


class _GineAgg(Function):
    @staticmethod
    def forward(ctx, h, ee, eps, plan, rplan):
        h, ee = _c(h), _c(ee)
        ctx.save_for_backward(h, ee, eps)
        ctx.rplan = rplan
        return ops.gine_aggregate(h, ee, plan, eps.detach())

    @staticmethod
    def backward(ctx, g):
        h, ee, eps = ctx.saved_tensors
        rp = ctx.rplan
        g = _c(g)
        dh, dee = torch.empty_like(h), torch.empty_like(ee)
        with ops._span("sn_gine_aggregate_bwd_f32"):
            check(lib().sn_gine_aggregate_bwd_f32(ptr(h), ptr(ee), ptr(g), h.shape[0], h.shape[1], ptr(rp.rowptr), ptr(rp.col),
                                                  ptr(rp.eperm), ptr(eps.detach()), ptr(dh), ptr(dee), stream()),
                  "sn_gine_aggregate_bwd_f32")
        return dh, dee, (dot(g, h) if ctx.needs_input_grad[2] else None), None, None


def gine_aggregate(h, ee, eps, plan, rplan):
    return _GineAgg.apply(h, ee, eps, plan, rplan)


# ----------------------------------------------------------------------------- set attention / LayerNorm / slot sum
class _Attention(Function):
    @staticmethod
    def forward(ctx, q, k, v, N, K, heads, nvalid, prob_mask):
        q, k, v = _c(q), _c(k), _c(v)
        ctx.save_for_backward(q, k, v)
        ctx.meta = (N, K, heads, nvalid, prob_mask)
        return ops.set_attention(q, k, v, N, K, heads, nvalid, prob_mask)

    @staticmethod
    def backward(ctx, g):
        q, k, v = ctx.saved_tensors
        N, K, heads, nvalid, prob_mask = ctx.meta
        g = _c(g)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        with ops._span("sn_set_attention_bwd_f32"):
            check(lib().sn_set_attention_bwd_f32(ptr(q), ptr(k), ptr(v), ptr(g), N, K, heads, q.shape[-1] // heads, ptr(nvalid),
                                                 ptr(prob_mask), ptr(dq), ptr(dk), ptr(dv), stream()), "sn_set_attention_bwd_f32")
        return dq, dk, dv, None, None, None, None, None


def set_attention(q, k, v, N, K, heads, nvalid=None, prob_mask=None):
    """prob_mask: ops.attention_dropout_mask(...) for the train-mode attention dropout, or None."""
    return _Attention.apply(q, k, v, N, K, heads, nvalid, prob_mask)


class _LayerNorm(Function):
    @staticmethod
    def forward(ctx, x, residual, gamma, beta, eps, nvalid, K):
        x = _c(x)
        res = None if residual is None else _c(residual)
        ctx.save_for_backward(x, res, gamma)
        ctx.meta = (eps, nvalid, K)
        ctx.affine = (gamma, beta)
        return ops.masked_layernorm(x, res, gamma.detach(), beta.detach(), eps, nvalid, K)

    @staticmethod
    def backward(ctx, g):
        x, res, gamma = ctx.saved_tensors
        eps, nvalid, K = ctx.meta
        g = _c(g)
        Cc = x.shape[-1]
        R = x.numel() // Cc
        du = torch.empty_like(x)
        from .train_stage import direct_grad
        tg, tb = direct_grad(ctx.affine[0]), direct_grad(ctx.affine[1])
        acc = tg is not None and tb is not None
        dgamma = tg if acc else torch.empty(Cc, dtype=torch.float32, device=x.device)
        dbeta = tb if acc else torch.empty_like(dgamma)
        scratch = torch.empty(int(lib().sn_layernorm_bwd_scratch_floats(R, Cc)), dtype=torch.float32, device=x.device)
        fn = lib().sn_masked_layernorm_bwd_acc_f32 if acc else lib().sn_masked_layernorm_bwd_f32
        with ops._span("sn_masked_layernorm_bwd_f32"):
            check(fn(ptr(x), ptr(res), ptr(g), R, Cc, ptr(gamma.detach()), float(eps), ptr(nvalid),
                     int(K), ptr(du), ptr(dgamma), ptr(dbeta), ptr(scratch), stream()), "sn_masked_layernorm_bwd_f32")
        return du, (du if res is not None else None), (None if acc else dgamma), (None if acc else dbeta), None, None, None


def masked_layernorm(x, residual, gamma, beta, eps, nvalid=None, K=0):
    return _LayerNorm.apply(x, residual, gamma, beta, eps, nvalid, K)


class _SlotSum(Function):
    @staticmethod
    def forward(ctx, x, N, K, nvalid):
        ctx.meta = (N, K, nvalid, x.shape)
        return ops.slot_sum(_c(x), N, K)

    @staticmethod
    def backward(ctx, g):
        N, K, nvalid, shape = ctx.meta
        g = _c(g)
        dx = torch.empty(N * K, g.shape[-1], dtype=torch.float32, device=g.device)
        check(lib().sn_slot_broadcast_f32(ptr(g), N, K, g.shape[-1], ptr(nvalid), ptr(dx), stream()), "sn_slot_broadcast_f32")
        return dx.view(shape), None, None, None


def slot_sum(x, N, K, nvalid=None):
    return _SlotSum.apply(x, N, K, nvalid)


# ----------------------------------------------------------------------------- embeddings / pooling
class _EmbeddingSum(Function):
    @staticmethod
    def forward(ctx, idx, status, *tables):
        if idx.dim() == 1:
            idx = idx.unsqueeze(1)
        idx = idx.contiguous()
        ctx.idx = idx
        ctx.shapes = [t.shape for t in tables]
        ctx.tables = tables
        return ops.embedding_sum(idx, [t.detach() for t in tables], status)

    @staticmethod
    def backward(ctx, g):
        idx = ctx.idx
        g = _c(g)
        R, nf = idx.shape
        # the kernel ADDS into its target (dT[v] += ...): a table whose .grad the optimiser owns (optim.FlatAdam) is accumulated in place —
        # no zero-filled [vocab, d] temporary and no elementwise add per table
        from .train_stage import direct_grad
        direct = [direct_grad(t) if f < nf else None for f, t in enumerate(ctx.tables)]
        grads = [(None if direct[f] is not None else torch.zeros(s, dtype=torch.float32, device=g.device)) if f < nf else None
                 for f, s in enumerate(ctx.shapes)]
        arr = (C.c_void_p * nf)(*[(direct[f] if direct[f] is not None else grads[f]).data_ptr() for f in range(nf)])
        rows = (C.c_int64 * nf)(*[ctx.shapes[f][0] for f in range(nf)])
        # (out-of-range indices were reported by the forward; the backward skips them — deterministic, no atomics)
        scratch = torch.empty(int(lib().sn_embedding_bwd_scratch_floats(R, nf, rows, g.shape[-1])), dtype=torch.float32, device=g.device)
        check(lib().sn_embedding_sum_bwd_f32(ptr(idx), nf, nf, R, arr, rows, g.shape[-1], ptr(g), None, ptr(scratch), stream()),
              "sn_embedding_sum_bwd_f32")
        return (None, None, *grads)


def embedding_sum(idx, tables, status=None):
    """sum_f tables[f][idx[:, f]] (DiscreteEncoder); tables beyond idx's feature columns get no gradient.  `status`: see
    ops.embedding_sum (None = the op checks the index range itself, one host sync)."""
    return _EmbeddingSum.apply(idx, status, *tables)


class _EmbeddingSumLayers(Function):
    """out[l] = sum_f tables[l][f][idx[:, f]] for L encoders over ONE index block (the per-layer edge encoders of GNN.forward,
    model.py:52-60); the adjoint is one launch pair for all L planes (sn_embedding_sum_bwd_layers_f32)."""

    @staticmethod
    def forward(ctx, idx, status, L, *tables):
        if idx.dim() == 1:
            idx = idx.unsqueeze(1)
        idx = idx.contiguous()
        R, nf = idx.shape
        nt = len(tables) // L
        if nf > nt:
            raise ValueError("embedding_sum_layers: more feature columns than embedding tables")
        Cc = tables[0].shape[1]
        out = torch.empty(L, R, Cc, dtype=torch.float32, device=idx.device)
        rows = (C.c_int64 * nf)(*[tables[f].shape[0] for f in range(nf)])
        tabs = [[ops._f32c(tables[l * nt + f].detach(), "embedding table") for f in range(nf)] for l in range(L)]
        if any(t.shape != tables[f].shape for ts in tabs for f, t in enumerate(ts)):
            raise ValueError("embedding_sum_layers: the layers' tables must have one shape per feature column")
        if nf <= 4 and L <= 16:          # all planes in one launch
            arr = (C.c_void_p * (L * nf))(*[t.data_ptr() for ts in tabs for t in ts])
            with ops._span("sn_embedding_sum_layers_f32"):
                check(lib().sn_embedding_sum_layers_f32(ptr(idx), nf, nf, R, L, arr, rows, Cc, ptr(out), ptr(status), stream()),
                      "sn_embedding_sum_layers_f32")
        else:
            for l in range(L):
                arr = (C.c_void_p * nf)(*[t.data_ptr() for t in tabs[l]])
                with ops._span("sn_embedding_sum_f32"):
                    check(lib().sn_embedding_sum_f32(ptr(idx), nf, nf, R, arr, rows, Cc, out[l].data_ptr(), ptr(status), stream()),
                          "sn_embedding_sum_f32")
        ctx.idx, ctx.L, ctx.nt, ctx.tables = idx, L, nt, tables
        return out

    @staticmethod
    def backward(ctx, g):
        idx, L, nt, tables = ctx.idx, ctx.L, ctx.nt, ctx.tables
        g = _c(g)
        R, nf = idx.shape
        Cc = g.shape[-1]
        from .train_stage import direct_grad
        grads, ptrs = [], []
        for l in range(L):
            for f in range(nt):
                t = tables[l * nt + f]
                if f >= nf:
                    grads.append(None)
                    continue
                d = direct_grad(t)
                grads.append(None if d is not None else torch.zeros(t.shape, dtype=torch.float32, device=g.device))
                ptrs.append((d if d is not None else grads[-1]).data_ptr())
        arr = (C.c_void_p * (L * nf))(*ptrs)
        rows = (C.c_int64 * nf)(*[tables[f].shape[0] for f in range(nf)])
        scratch = torch.empty(int(lib().sn_embedding_bwd_layers_scratch_floats(R, L, Cc)), dtype=torch.float32, device=g.device)
        with ops._span("sn_embedding_sum_bwd_layers_f32"):
            check(lib().sn_embedding_sum_bwd_layers_f32(ptr(idx), nf, nf, R, L, arr, rows, Cc, ptr(g), None, ptr(scratch), stream()),
                  "sn_embedding_sum_bwd_layers_f32")
        return (None, None, None, *grads)


def embedding_sum_layers(idx, layer_tables, status):
    """[L, R, C]: plane l = sum_f layer_tables[l][f][idx[:, f]].  The block carries a gradient buffer (`_sn_gbuf`) that
    train_stage.gine_layer(..., layer=l) fills plane by plane (see _GineLayer)."""
    L = len(layer_tables)
    flat = [t for tabs in layer_tables for t in tabs]
    out = _EmbeddingSumLayers.apply(idx, status, L, *flat)
    out._sn_gbuf = torch.empty_like(out)
    return out


class _SegmentPool(Function):
    @staticmethod
    def forward(ctx, h, plan, mode):
        ctx.meta = (plan, mode, h.shape)
        return ops.segment_pool(_c(h), plan, mode)

    @staticmethod
    def backward(ctx, g):
        plan, mode, shape = ctx.meta
        g = _c(g)
        dx = torch.empty(shape, dtype=torch.float32, device=g.device)
        check(lib().sn_segment_broadcast_f32(ptr(g), plan.B, g.shape[-1], ptr(plan.graph_ptr), 1 if mode == "mean" else 0, ptr(dx),
                                             stream()), "sn_segment_broadcast_f32")
        return dx, None, None


def segment_pool(h, plan, mode="add"):
    return _SegmentPool.apply(h, plan, mode)


class _SegmentBcastAdd(Function):
    """y = act(x1 + x2[graph(row)]): the `x1 + x2` of the DeepSets / IGN layers (x2 = a per-graph row broadcast over the graph's
    nodes; LearningFilters/models.py:74-77, ign.py:330-335).  d x1 = dy', d x2 = segment sum of dy' (dy' = dy masked by the ReLU)."""

    @staticmethod
    def forward(ctx, x1, x2, plan, relu):
        x2 = _c(x2)
        xb = torch.empty_like(x1)
        check(lib().sn_segment_broadcast_f32(ptr(x2), plan.B, x2.shape[-1], ptr(plan.graph_ptr), 0, ptr(xb), stream()),
              "sn_segment_broadcast_f32")
        y = ops.pointwise(_c(x1), act="relu_sum" if relu else "none", residual=xb)
        ctx.meta = (plan, relu)
        ctx.save_for_backward(y if relu else None)
        return y

    @staticmethod
    def backward(ctx, g):
        plan, relu = ctx.meta
        (y,) = ctx.saved_tensors
        g = _c(g)
        if relu:
            g = relu_bwd(y, g, None, 0)
        return g, ops.segment_pool(g, plan, "add"), None, None


def segment_bcast_add(x1, x2, plan, relu=False):
    return _SegmentBcastAdd.apply(x1, x2, plan, relu)


class _DenseAttention(Function):
    @staticmethod
    def forward(ctx, q, k, v, heads):
        q, k, v = _c(q), _c(k), _c(v)
        out, lse = ops.dense_attention(q, k, v, heads, want_lse=True)
        ctx.save_for_backward(q, k, v, out, lse)
        ctx.heads = heads
        return out

    @staticmethod
    def backward(ctx, g):
        q, k, v, out, lse = ctx.saved_tensors
        g = _c(g)
        Bt, L, d = q.shape
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        delta = torch.empty_like(lse)
        with ops._span("sn_dense_attention_bwd_f32"):
            check(lib().sn_dense_attention_bwd_f32(ptr(q), ptr(k), ptr(v), ptr(out), ptr(lse), ptr(g), Bt, L, ctx.heads, d // ctx.heads,
                                                   ptr(dq), ptr(dk), ptr(dv), ptr(delta), stream()), "sn_dense_attention_bwd_f32")
        return dq, dk, dv, None


def dense_attention(q, k, v, heads):
    """softmax(q k^T / sqrt(dk)) v per head over whole sequences; q, k, v [Bt, L, heads*dk]."""
    return _DenseAttention.apply(q, k, v, heads)


# ----------------------------------------------------------------------------- PNA / sparse graph Transformer
class _GatherRows(Function):
    """h[idx] with idx = one row of the edge list: node rows copied onto edges (pna_layer.py:38-44).  `plan`: the CSR that groups the
    edges by that row (the batch plan for dst, the plan of the flipped edge list for src): the adjoint is a CSR walk, no atomics."""

    @staticmethod
    def forward(ctx, h, idx, plan):
        ctx.plan, ctx.shape = plan, h.shape
        return h.index_select(0, idx)

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        plan = ctx.plan
        N, Cc = ctx.shape
        out = torch.empty(N, Cc, dtype=torch.float32, device=g.device)
        check(lib().sn_edge_rows_sum_f32(ptr(g), Cc, Cc, N, ptr(plan.rowptr), ptr(plan.eperm), ptr(out), Cc, stream()), "sn_edge_rows_sum_f32")
        return out, None, None


def gather_rows(h, idx, plan):
    return _GatherRows.apply(h, idx, plan)


class _PnaAggregate(Function):
    @staticmethod
    def forward(ctx, msg, hself, plan, avg_log):
        msg, hself = _c(msg), _c(hself)
        ctx.save_for_backward(msg)
        ctx.meta = (plan, float(avg_log))
        return ops.pna_aggregate(msg, hself, plan, avg_log)

    @staticmethod
    def backward(ctx, g):
        (msg,) = ctx.saved_tensors
        plan, avg_log = ctx.meta
        g = _c(g)
        Cc = msg.shape[1]
        dmsg = torch.zeros_like(msg)                  # (every edge has a destination: all rows are written; zeros guard malformed plans)
        dself = torch.empty(plan.N, Cc, dtype=torch.float32, device=g.device)
        check(lib().sn_pna_aggregate_bwd_f32(ptr(msg), Cc, Cc, plan.N, ptr(plan.rowptr), ptr(plan.eperm), avg_log, ptr(g), 13 * Cc, ptr(dmsg),
                                             ptr(dself), stream()), "sn_pna_aggregate_bwd_f32")
        return dmsg, dself, None, None


def pna_aggregate(msg, hself, plan, avg_log):
    return _PnaAggregate.apply(msg, hself, plan, avg_log)


class _EdgeAttention(Function):
    @staticmethod
    def forward(ctx, Q, K, V, Ee, plan, rplan, heads):
        Q, K, V, Ee = _c(Q), _c(K), _c(V), _c(Ee)
        out = ops.edge_attention(Q, K, V, Ee, plan, heads)
        ctx.save_for_backward(Q, K, V, Ee, out)
        ctx.meta = (plan, rplan, heads)
        return out

    @staticmethod
    def backward(ctx, g):
        Q, K, V, Ee, out = ctx.saved_tensors
        plan, rp, heads = ctx.meta
        g = _c(g)
        E = Ee.shape[0]
        dQ, dK, dV, dE = torch.empty_like(Q), torch.empty_like(K), torch.empty_like(V), torch.zeros_like(Ee)
        scratch = torch.empty(2 * max(E, 1) * heads, dtype=torch.float32, device=g.device)
        with ops._span("sn_edge_attention_bwd_f32"):
            check(lib().sn_edge_attention_bwd_f32(ptr(Q), ptr(K), ptr(V), ptr(Ee), ptr(out), ptr(g), plan.N, E, heads, Q.shape[1] // heads,
                                                  ptr(plan.rowptr), ptr(plan.col), ptr(plan.eperm), ptr(rp.rowptr), ptr(rp.col), ptr(rp.eperm),
                                                  ptr(dQ), ptr(dK), ptr(dV), ptr(dE), ptr(scratch), stream()), "sn_edge_attention_bwd_f32")
        return dQ, dK, dV, dE, None, None, None


def edge_attention(Q, K, V, Ee, plan, rplan, heads):
    return _EdgeAttention.apply(Q, K, V, Ee, plan, rplan, heads)


class _GatAggregate(Function):
    """dgl GATConv after its fc (gat_net.py:62-66): edge softmax of leaky_relu(el[src] + er[dst]) + weighted sum + bias + ReLU."""

    @staticmethod
    def forward(ctx, feat, attn_l, attn_r, bias, plan, rplan, heads, slope, relu):
        feat = _c(feat)
        out, lse = ops.gat_aggregate(feat, attn_l, attn_r, bias, plan, heads, slope, relu=relu, want_lse=True)
        ctx.save_for_backward(feat, attn_l, attn_r, bias, out, lse)
        ctx.meta = (plan, rplan, heads, slope, relu)
        return out

    @staticmethod
    def backward(ctx, g):
        feat, attn_l, attn_r, bias, out, lse = ctx.saved_tensors
        plan, rp, H, slope, relu = ctx.meta
        g = _c(g)
        N, d = feat.shape
        Cc = d // H
        E = plan.col.numel()
        al, ar = _c(attn_l.reshape(-1)), _c(attn_r.reshape(-1))
        dfeat, gob = torch.empty_like(feat), torch.empty_like(feat)
        dlr = torch.empty(2, N, H, dtype=torch.float32, device=g.device)                 # d el | d er
        scratch = torch.empty(2 * max(E, 1) * H, dtype=torch.float32, device=g.device)
        with ops._span("sn_gat_aggregate_bwd_f32"):
            check(lib().sn_gat_aggregate_bwd_f32(ptr(feat), ptr(al), ptr(ar), ptr(None if bias is None else _c(bias.reshape(-1))), ptr(out), ptr(lse),
                                                 ptr(g), N, E, H, Cc, float(slope), int(relu), ptr(plan.rowptr), ptr(plan.col), ptr(plan.eperm),
                                                 ptr(rp.rowptr), ptr(rp.col), ptr(rp.eperm), ptr(dfeat), ptr(gob), ptr(dlr[0]), ptr(dlr[1]),
                                                 ptr(scratch), stream()), "sn_gat_aggregate_bwd_f32")
        # attn_l / attn_r gradients: the diagonal [h, h*C:(h+1)*C] blocks of (d el | d er)^T feat — one weight-gradient reduction; the
        # bias gradient is the column sum of the masked cotangent (the same reduction's bias output)
        lr = torch.cat([dlr[0], dlr[1]], dim=1)                                          # [N, 2H]
        dW, _ = linear_wgrad(feat, lr, None, 0, want_bias=False)                         # [2H, H*C]
        blocks = dW.view(2, H, H, Cc)
        idx = torch.arange(H, device=g.device)
        dal = blocks[0, idx, idx].reshape(attn_l.shape)
        dar = blocks[1, idx, idx].reshape(attn_r.shape)
        db = None
        if bias is not None:
            _, db = linear_wgrad(lr, gob, None, 0, want_bias=True)
            db = db.reshape(bias.shape)
        return dfeat, dal, dar, db, None, None, None, None, None


def gat_aggregate(feat, attn_l, attn_r, bias, plan, rplan, heads, slope=0.2, relu=True):
    return _GatAggregate.apply(feat, attn_l, attn_r, bias, plan, rplan, heads, slope, relu)


class _ActResidual(Function):
    """act(x [* rowscale]) + residual with act in none / relu / leaky: FCLayer's LeakyReLU + the layer residual (pna_layer.py:126-134),
    graph_norm's row scaling (:75-76)."""

    @staticmethod
    def forward(ctx, x, residual, rowscale, act, slope):
        x = _c(x)
        ctx.save_for_backward(x if act != "none" and rowscale is None else None, rowscale)
        ctx.meta = (act, slope, residual is not None)
        if act != "none" and rowscale is not None:
            raise ValueError("act_residual: a row scale combines with act = none only")
        return ops.pointwise(x, rowscale=rowscale, act=act, slope=slope, residual=None if residual is None else _c(residual))

    @staticmethod
    def backward(ctx, g):
        x, rowscale = ctx.saved_tensors
        act, slope, has_res = ctx.meta
        g = _c(g)
        Cc = g.shape[-1]
        dx = torch.empty_like(g)
        check(lib().sn_act_bwd_f32(ptr(x), ptr(g), g.numel() // Cc, Cc, ptr(rowscale), {"none": 0, "relu": 1, "leaky": 2}[act], float(slope),
                                   ptr(dx), stream()), "sn_act_bwd_f32")
        return dx, (g if has_res else None), None, None, None


def act_residual(x, residual=None, rowscale=None, act="none", slope=0.01):
    return _ActResidual.apply(x, residual, rowscale, act, slope)


# ----------------------------------------------------------------------------- GatedGCN edge-gated aggregation
class _Gated(Function):
    @staticmethod
    def forward(ctx, Ah, Bh, Dh, Eh, Ce, plan, rplan):
        h, e, den = ops.gated_aggregate(Ah, Bh, Dh, Eh, Ce, plan, want_den=True)
        ctx.save_for_backward(_c(Ah), _c(Bh), e, h, den)
        ctx.meta = (plan, rplan)
        return h, e

    @staticmethod
    def backward(ctx, dh, de):
        Ah, Bh, e, h, den = ctx.saved_tensors
        plan, rp = ctx.meta
        dh = _c(dh) if dh is not None else torch.zeros_like(h)
        de = _c(de) if de is not None else None
        N, Cc = Ah.shape
        dB, dD, dE = torch.empty_like(Ah), torch.empty_like(Ah), torch.empty_like(Ah)
        de_new, scratch = torch.empty_like(e), torch.empty_like(Ah)
        with ops._span("sn_gated_aggregate_bwd_f32"):
            check(lib().sn_gated_aggregate_bwd_f32(ptr(Ah), ptr(Bh), ptr(e), ptr(h), ptr(den), ptr(dh), ptr(de), N, Cc, ptr(plan.rowptr),
                                                   ptr(plan.col), ptr(plan.eperm), ptr(rp.rowptr), ptr(rp.col), ptr(rp.eperm), ptr(dB),
                                                   ptr(dD), ptr(dE), ptr(de_new), ptr(scratch), stream()), "sn_gated_aggregate_bwd_f32")
        return dh, dB, dD, dE, de_new, None, None


def gated_aggregate(Ah, Bh, Dh, Eh, Ce, plan, rplan):
    """(h, e) of GatedGCN's message passing (gatedgcn_layer.py:51-56); differentiable in all five inputs."""
    return _Gated.apply(Ah, Bh, Dh, Eh, Ce, plan, rplan)
