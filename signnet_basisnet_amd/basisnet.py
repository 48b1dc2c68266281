"""BasisNet / LearningFilters modules (drop-in `nn.Module` surface, HIP forward).

Mirrors LearningFilters/ign.py:9-39 (IGN2to1 with layer_2_to_1 :88-128 and layer_1_to_1 :174-214),
signbasisnet.py:11-41 (SignPlus, IGNBasisInv) and models.py:58-113 (EqDeepSetsEncoder) — same constructors,
same forward contracts:
    IGNBasisInv(mult_lst, in_channels, hidden_channels).forward(proj [b,1,n,n], mult) -> [b, mult, n]
    EqDeepSetsEncoder(...).forward(x [set, F] or [b, set, F]) -> [..., out]
    SignPlus(model).forward(v) = model(v) + model(-v)
All arithmetic runs in libsignnet_hip.so (the 2->1 contractions in sn_ign_contract_2to1_f32, every Linear in
sn_masked_linear_f32, means / BatchNorm statistics in the segment / column-statistics kernels).  No CPU path.
In train mode with gradients enabled every module here builds its forward from the differentiable device ops of autograd.py
(linear, segment mean / broadcast-add, batch-statistic BatchNorm), so `loss.backward()` fills the parameter gradients the
reference's 2 000-epoch loop uses (training.py:132-143) — no ATen arithmetic; torch only routes views, concatenations and
gradient accumulation.  Under torch.no_grad() train mode gives the same forward value (running statistics updated).

Difference from the reference worth knowing: the reference's equivariant-layer
coefficients are `nn.Parameter(...).to(device)`, i.e. NOT registered parameters when the device differs from
the default; here `coeffs` / `bias` are ordinary registered Parameters with the reference's initialisation.
"""
from __future__ import annotations

import ctypes as C
import math

import torch
import torch.nn as nn

from . import ops
from ._lib import check, lib, ptr, stream

BN_EPS = 1e-5


class layer_2_to_1(nn.Module):
    def __init__(self, input_depth, output_depth):
        super().__init__()
        if input_depth != 1:
            raise ValueError("HIP IGN2to1: the 2->1 layer takes the single-channel projector stack (in_channels = 1)")
        self.basis_dimension = 5
        self.coeffs = nn.Parameter(torch.randn(input_depth, output_depth, 5) * math.sqrt(2.0) / (input_depth + output_depth))
        self.bias = nn.Parameter(torch.zeros(1, output_depth, 1))


class layer_1_to_1(nn.Module):
    def __init__(self, input_depth, output_depth):
        super().__init__()
        self.basis_dimension = 2
        self.coeffs = nn.Parameter(torch.randn(input_depth, output_depth, 2) * math.sqrt(2.0) / (input_depth + output_depth))
        self.bias = nn.Parameter(torch.zeros(1, output_depth, 1))


class _IgnMlpParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("w0", "b0", "s0", "t0", "w1a", "w1b", "b1", "s1", "t1", "w2a", "w2b", "b2", "s2", "t2",
                                          "fc1_w", "fc1_b", "fc2_w", "fc2_b")]


_DS_MAX = 8


class _DeepSetsTailParams(C.Structure):
    _fields_ = [("n_layers", C.c_int), ("use_bn", C.c_int), ("eps", C.c_float), ("split0", C.c_int), ("width", C.c_int * _DS_MAX)] + \
               [(n, C.c_void_p * _DS_MAX) for n in ("w1", "b1", "w2", "b2", "gamma", "beta")]


def _lin(W, b):
    W = W.detach().contiguous()
    return ops.PackedLinear(ops.pack_weight(W), W.shape[0], W.shape[1], None if b is None else b.detach().reshape(-1).contiguous())


class IGN2to1(nn.Module):
    """batch x 1 x n x n -> batch x out x n   (ign.py:9-39).  `num_layers` is ignored as in the reference."""

    def __init__(self, in_channels, hidden_channels, out_channels, num_layers=1, device=None, use_bn=True):
        super().__init__()
        if not use_bn:
            raise ValueError("HIP IGN2to1 supports use_bn=True (the reference default)")
        self.bns = nn.ModuleList([nn.BatchNorm1d(hidden_channels) for _ in range(4)])   # 4 created, 3 used (ign.py:12-25)
        self.equi_layers = nn.ModuleList([layer_2_to_1(in_channels, hidden_channels),
                                          layer_1_to_1(hidden_channels, hidden_channels),
                                          layer_1_to_1(hidden_channels, hidden_channels)])
        self.fc1 = nn.Linear(hidden_channels, hidden_channels)
        self.fc2 = nn.Linear(hidden_channels, out_channels)
        self._prep = None
        self.fused_head = True       # eval: sn_ign_mlp_f32 where it applies (False: the layer-at-a-time entry points)
        # fires also when a PARENT module's (IGNBasisInv, a wrapper) load_state_dict recurses into this one
        self.register_load_state_dict_post_hook(lambda m, keys=None: setattr(m, "_prep", None))

    def train(self, mode=True):
        self._prep = None
        return super().train(mode)

    def _apply(self, fn, *a, **k):
        self._prep = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._prep = None
        return super().load_state_dict(*a, **k)

    def _prepare(self, train=False):
        from .dgl_deepsigns import _BNSite
        e0, e1, e2 = self.equi_layers
        P = {}
        # einsum('dsb,ndbi->nsi'): out[s] = sum_k coeffs[0,s,k] * ops[k]  -> weight [S, 5]
        P["l0"] = _lin(e0.coeffs.detach()[0], e0.bias)
        for name, e in (("l1", e1), ("l2", e2)):
            c = e.coeffs.detach()                                              # [D, S, 2]
            W = torch.cat([c[:, :, 0].t(), c[:, :, 1].t()], dim=1)             # [S, 2D]: [identity | mean] blocks
            P[name] = _lin(W, e.bias)
            # eval: the two blocks apart — W_a on the rows, W_b on the per-matrix mean, which enters as a bias per block of n rows
            P[name + "_a"] = _lin(c[:, :, 0].t(), e.bias)
            P[name + "_b"] = _lin(c[:, :, 1].t(), None)
        P["bn"] = [_BNSite(self.bns[i], train) for i in range(3)]
        P["fc1"] = _lin(self.fc1.weight, self.fc1.bias)
        P["fc2"] = _lin(self.fc2.weight, self.fc2.bias)
        if not train:
            # the one-launch head (sn_ign_mlp_f32): raw row-major matrices and the folded BatchNorms, kept alive by the cache
            f = lambda t: t.detach().float().contiguous()
            keep = [f(e0.coeffs[0]), f(e0.bias.reshape(-1)), P["bn"][0].scale, P["bn"][0].shift]
            for i, e in ((1, e1), (2, e2)):
                keep += [f(e.coeffs[:, :, 0].t()), f(e.coeffs[:, :, 1].t()), f(e.bias.reshape(-1)), P["bn"][i].scale, P["bn"][i].shift]
            keep += [f(self.fc1.weight), f(self.fc1.bias), f(self.fc2.weight), f(self.fc2.bias)]
            P["mlp_keep"] = keep
            P["mlp"] = _IgnMlpParams(*[t.data_ptr() for t in keep])
        return P

    def _relu_bn(self, x, pl, site, train):
        """equivariant layer (a Linear over the contraction basis) -> ReLU -> BatchNorm1d(hidden) on [b, hidden, n]
        (ign.py:31-33: activation BEFORE the norm).  train: batch statistics over the b*n rows + running-stat update."""
        if not train:
            return ops.masked_linear(x, pl, relu_pre=True, scale=site.scale, shift=site.shift)
        y = ops.masked_linear(x, pl, relu=True)
        sc, sh = site.affine(y, True)
        return ops.masked_affine(y, scale=sc, shift=sh)

    def forward(self, x):
        ops.require_cuda(x)
        return self.forward_contractions(ops.ign_contract_2to1(x.contiguous()))           # [b, n, 5]

    def forward_contractions(self, o):
        """Everything after the 2->1 contractions: o [b, n, 5] = contractions_2_to_1 of the input stack (ign.py:344-374), from
        sn_ign_contract_2to1_f32 (the projectors) or sn_ign_contract_eigvecs_f32 (the eigenvectors alone)."""
        train = self.training
        if train and torch.is_grad_enabled():
            return self._forward_autograd(o)
        if train:
            P = self._prepare(True)
        else:
            if self._prep is None:
                self._prep = self._prepare()
            P = self._prep
        b, n = o.shape[0], o.shape[1]
        H, O = self.fc1.weight.shape[0], self.fc2.weight.shape[0]
        if not train and self.fused_head and lib().sn_ign_mlp_supported(int(n), int(H), int(O)):
            # eval: the whole head in ONE launch, a workgroup per matrix, the rows' channels in registers; y comes out as [b, O, n]
            o = o.contiguous()
            y = torch.empty(b, O, n, dtype=torch.float32, device=o.device)
            with ops._span("sn_ign_mlp_f32"):
                check(lib().sn_ign_mlp_f32(ptr(o), b, n, H, O, C.byref(P["mlp"]), ptr(y), stream()), "sn_ign_mlp_f32")
            return y
        h = self._relu_bn(o.reshape(b * n, 5), P["l0"], P["bn"][0], train)
        seg = torch.arange(0, b * n + 1, n, dtype=torch.int32, device=o.device)             # rows of matrix i: [i*n, (i+1)*n)
        segplan = _SegPlan(b, seg)
        for li, name in ((1, "l1"), (2, "l2")):
            m = ops.segment_pool(h, segplan, "mean")                                       # [b, H]   sum_n h / n (ign.py:405-414)
            if not train:
                # Linear over cat[h, mean broadcast] = W_a h + (W_b mean) per matrix: the [b*n, 2H] concatenation is never formed
                site = P["bn"][li]
                h = ops.linear_block_bias(h, P[name + "_a"], ops.masked_linear(m, P[name + "_b"]), n, relu_pre=True,
                                          scale=site.scale, shift=site.shift)
                continue
            cat = torch.cat([h.view(b, n, -1), m.unsqueeze(1).expand(b, n, m.shape[1])], dim=-1).contiguous()
            h = self._relu_bn(cat.view(b * n, -1), P[name], P["bn"][li], train)
        h = ops.masked_linear(h, P["fc1"], relu=True)
        y = ops.masked_linear(h, P["fc2"])                                                 # [b*n, out]
        return y.view(b, n, -1).transpose(2, 1).contiguous()                               # [b, out, n]


    def _forward_autograd(self, o):
        """The same network from differentiable ops (train mode): equivariant layer = Linear over the contraction basis
        (identity block on the rows + mean block on the per-matrix mean, broadcast back), ReLU, batch-statistic BatchNorm."""
        from . import autograd as AG
        e0, e1, e2 = self.equi_layers
        b, n = o.shape[0], o.shape[1]
        seg = _SegPlan(b, torch.arange(0, b * n + 1, n, dtype=torch.int32, device=o.device))
        h = AG.linear(o.reshape(b * n, 5), e0.coeffs[0], e0.bias.reshape(-1), relu=True)
        h = AG.bn_act(h, self.bns[0], relu=False)
        for i, e in ((1, e1), (2, e2)):
            x1 = AG.linear(h, e.coeffs[:, :, 0].t(), e.bias.reshape(-1))
            x2 = AG.linear(AG.segment_pool(h, seg, "mean"), e.coeffs[:, :, 1].t(), None)
            h = AG.bn_act(AG.segment_bcast_add(x1, x2, seg, relu=True), self.bns[i], relu=False)
        h = AG.linear(h, self.fc1.weight, self.fc1.bias, relu=True)
        y = AG.linear(h, self.fc2.weight, self.fc2.bias)
        return y.view(b, n, -1).transpose(2, 1).contiguous()


class _SegPlan:
    """Just enough of a GraphPlan for ops.segment_pool: equal-length row segments."""

    def __init__(self, B, graph_ptr):
        self.B, self.graph_ptr = B, graph_ptr


class IGNBasisInv(nn.Module):
    """One IGN2to1 per eigenvalue multiplicity (signbasisnet.py:23-41)."""

    def __init__(self, mult_lst, in_channels, hidden_channels=16, num_layers=2):
        super().__init__()
        self.encs = nn.ModuleList()
        self.mult_to_idx = {}
        for i, mult in enumerate(mult_lst):
            self.encs.append(IGN2to1(1, hidden_channels, mult, num_layers=num_layers))
            self.mult_to_idx[mult] = i

    def forward(self, proj, mult):
        return self.encs[self.mult_to_idx[mult]](proj)

    def forward_contractions(self, o, mult):
        """forward(proj, mult) from the 2->1 contractions o [b, N, 5] of `proj` (constants of the graph: computed once)."""
        return self.encs[self.mult_to_idx[mult]].forward_contractions(o)

    def forward_eigvecs(self, eigvecs, plan, contractions=None):
        """Extension (not in the reference): every multiplicity group evaluated from the eigenvectors alone — the 2->1
        contractions of P = V V^T are computed from V (4*N*mult bytes per eigenspace instead of the 4*N^2 of the projector;
        SURVEY.md §8 a20).  Returns {mult: [b, mult, N]}, the values `forward(same_size_projs[mult], mult)` gives (same maths,
        different fp32 summation order in the contractions)."""
        o = ops.ign_contract_eigvecs(eigvecs, plan) if contractions is None else contractions
        return {m: self.encs[self.mult_to_idx[m]].forward_contractions(plan.group(o, m)).contiguous() for m in plan.mults}


class IGNShared(nn.Module):
    """IGN BasisNet with one shared IGN2to1(1, hidden, 1) and a Linear(1, mult) per multiplicity (signbasisnet.py:43-64)."""

    def __init__(self, mult_lst, in_channels, hidden_channels=16, num_layers=2):
        super().__init__()
        self.enc = IGN2to1(1, hidden_channels, 1, num_layers=num_layers)
        self.fcs = nn.ModuleList()
        self.mult_to_idx = {}
        for i, mult in enumerate(mult_lst):
            self.fcs.append(nn.Linear(1, mult))
            self.mult_to_idx[mult] = i

    def forward(self, proj, mult):
        return self._head(self.enc(proj), mult)

    def forward_contractions(self, o, mult):
        return self._head(self.enc.forward_contractions(o), mult)

    def _head(self, x, mult):
        fc = self.fcs[self.mult_to_idx[mult]]                                # x: [b, 1, n]
        b, n = x.shape[0], x.shape[2]
        if self.training and torch.is_grad_enabled():
            from . import autograd as AG
            return AG.linear(x.reshape(b * n, 1), fc.weight, fc.bias).view(b, n, -1).transpose(2, 1).contiguous()
        y = ops.masked_linear(x.reshape(b * n, 1), _lin(fc.weight, fc.bias))   # transpose(2,1) . Linear(1, mult)   (:60-62)
        return y.view(b, n, -1).transpose(2, 1).contiguous()                # [b, mult, n]


class EqDeepSetsEncoder(nn.Module):
    """Equivariant DeepSets over the second-to-last axis (models.py:58-113).  BatchNorm here has
    track_running_stats=False, i.e. it always normalises with the statistics of the current input."""

    def __init__(self, in_channels, hidden_channels=32, out_channels=1, num_layers=3, use_bn=False, use_ln=False, dropout=0.0,
                 activation="relu"):
        super().__init__()
        if use_ln or activation != "relu":
            raise ValueError("HIP EqDeepSetsEncoder: relu / no LayerNorm only (what the reference instantiates)")
        if dropout:
            raise NotImplementedError("HIP EqDeepSetsEncoder: dropout > 0 is not built (the reference never sets it)")
        self.lins1, self.lins2 = nn.ModuleList(), nn.ModuleList()
        if use_bn:
            self.bns = nn.ModuleList()
        dims = [in_channels] + [hidden_channels] * (num_layers - 1) + [out_channels]
        for i in range(num_layers):
            self.lins1.append(nn.Linear(dims[i], dims[i + 1]))
            self.lins2.append(nn.Linear(dims[i], dims[i + 1]))
            if use_bn and i < num_layers - 1:
                self.bns.append(nn.BatchNorm1d(dims[i + 1], track_running_stats=False))
        self.use_bn = use_bn

    fused_tail = True        # eval, one set of <= 1024 rows, widths <= 32 behind the first layer: sn_deepsets_tail_f32

    def _tail_params(self):
        """Eval cache: the parameter block of sn_deepsets_tail_f32 (None when a layer behind the first is wider than 32)."""
        if getattr(self, "_tail_cache", None) is None:
            L = len(self.lins1)
            widths = [self.lins1[0].weight.shape[0]] + [self.lins1[i].weight.shape[0] for i in range(1, L)]
            if L > _DS_MAX or max(widths) > 32:
                self._tail_cache = (None,)
            else:
                f = lambda t: t.detach().float().contiguous()
                P, keep = _DeepSetsTailParams(), []
                P.n_layers, P.use_bn = L, int(self.use_bn)
                P.eps = float(self.bns[0].eps) if self.use_bn else 1e-5
                for i, w in enumerate(widths):
                    P.width[i] = int(w)
                for i in range(1, L):
                    ts = [f(self.lins1[i].weight), f(self.lins1[i].bias), f(self.lins2[i].weight), f(self.lins2[i].bias)]
                    keep += ts
                    P.w1[i], P.b1[i], P.w2[i], P.b2[i] = (t.data_ptr() for t in ts)
                    if self.use_bn:
                        bn = self.bns[i - 1]
                        gb = [f(bn.weight), f(bn.bias)]
                        keep += gb
                        P.gamma[i - 1], P.beta[i - 1] = gb[0].data_ptr(), gb[1].data_ptr()
                P.split0 = 1
                W01 = torch.cat([f(self.lins1[0].weight), f(self.lins2[0].weight)], 0).contiguous()
                b01 = torch.cat([f(self.lins1[0].bias), f(self.lins2[0].bias)], 0).contiguous()
                self._tail_cache = (P, keep, ops.PackedLinear(ops.pack_weight(W01), W01.shape[0], W01.shape[1], b01))
        return None if self._tail_cache[0] is None else self._tail_cache

    def _packed(self):
        """Eval cache: (lin1, lin2) of every layer as packed Linears (dropped by train(), .to(), load_state_dict())."""
        if getattr(self, "_pk_cache", None) is None:
            self._pk_cache = [(_lin(l1.weight, l1.bias), _lin(l2.weight, l2.bias)) for l1, l2 in zip(self.lins1, self.lins2)]
        return self._pk_cache

    def train(self, mode=True):
        self._pk_cache = self._tail_cache = None
        return super().train(mode)

    def _apply(self, fn, *a, **k):
        self._pk_cache = self._tail_cache = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._pk_cache = self._tail_cache = None
        return super().load_state_dict(*a, **k)

    def _load_from_state_dict(self, *a, **k):
        self._pk_cache = self._tail_cache = None
        return super()._load_from_state_dict(*a, **k)

    def forward(self, x, *args):
        ops.require_cuda(x)
        shp = x.shape
        three_d = x.dim() == 3
        if x.dim() not in (2, 3):
            raise ValueError("invalid x dimension")
        b, n = (shp[0], shp[1]) if three_d else (1, shp[0])
        h = x.contiguous().float().view(b * n, shp[-1])
        seg = _SegPlan(b, torch.arange(0, b * n + 1, n, dtype=torch.int32, device=x.device))
        L = len(self.lins1)
        if self.training and torch.is_grad_enabled():
            from . import autograd as AG
            for i in range(L):
                l1, l2 = self.lins1[i], self.lins2[i]
                last = i == L - 1
                x2 = AG.linear(AG.segment_pool(h, seg, "mean"), l2.weight, l2.bias)          # lin2(x.mean(dim=-2))
                h = AG.segment_bcast_add(AG.linear(h, l1.weight, l1.bias), x2, seg, relu=not last)
                if self.use_bn and not last:
                    h = AG.bn_act(h, self.bns[i], relu=False)
            return h.view(*shp[:-1], -1)
        if not self.training:
            # eval: lin1(x) + lin2(mean) as ONE Linear with a bias per block of n rows (the mean term), packed weights cached —
            # no [b*n, 2 F] concatenation, no per-forward packing
            P = self._packed()
            tail = self._tail_params() if (b == 1 and L >= 2 and self.fused_tail) else None
            if tail is not None and n * max(tail[0].width[i] for i in range(L - 1)) > 16384:
                tail = None
            if tail is not None:
                # one set: x [W1 ; W2]^T as ONE GEMM (the set mean commutes with lin2: mean(x) W2^T = mean(x W2^T)), then everything else in
                # ONE launch (sn_deepsets_tail_f32)
                z = ops.masked_linear(h, tail[2])
                y = torch.empty(n, self.lins1[-1].weight.shape[0], dtype=torch.float32, device=z.device)
                with ops._span("sn_deepsets_tail_f32"):
                    check(lib().sn_deepsets_tail_f32(ptr(z), n, C.byref(tail[0]), ptr(y), stream()), "sn_deepsets_tail_f32")
                return y.view(*shp[:-1], -1)
            for i in range(L):
                last = i == L - 1
                m = ops.segment_pool(h, seg, "mean")                                          # x.mean(dim=-2)
                h = ops.linear_block_bias(h, P[i][0], ops.masked_linear(m, P[i][1]), n, relu=not last)
                if self.use_bn and not last:
                    bn = self.bns[i]
                    mean, var, _ = ops.masked_colstats(h)                                     # stats over all b*n rows
                    sc, sh = ops.bn_fold_stats(bn.weight.detach(), bn.bias.detach(), mean, var, bn.eps)
                    h = ops.masked_affine(h, scale=sc, shift=sh)
            return h.view(*shp[:-1], -1)
        for i in range(L):
            l1, l2 = self.lins1[i], self.lins2[i]
            m = ops.segment_pool(h, seg, "mean")                                              # x.mean(dim=-2)
            cat = torch.cat([h.view(b, n, -1), m.unsqueeze(1).expand(b, n, m.shape[1])], dim=-1).contiguous()
            W = torch.cat([l1.weight.detach(), l2.weight.detach()], dim=1).contiguous()       # x1 + x2 as one Linear
            bias = ops.masked_affine(l1.bias.detach().view(1, -1).contiguous(), residual=l2.bias.detach().view(1, -1).contiguous())
            pl = ops.PackedLinear(ops.pack_weight(W), W.shape[0], W.shape[1], bias.view(-1))
            last = i == L - 1
            h = ops.masked_linear(cat.view(b * n, -1), pl, relu=not last)
            if self.use_bn and not last:
                bn = self.bns[i]
                mean, var, _ = ops.masked_colstats(h)                                         # stats over all b*n rows
                sc, sh = ops.bn_fold_stats(bn.weight.detach(), bn.bias.detach(), mean, var, bn.eps)
                h = ops.masked_affine(h, scale=sc, shift=sh)
        return h.view(*shp[:-1], -1)


class SignPlus(nn.Module):
    """model(v) + model(-v); with side features x (negate v, do not negate x): model(cat(v, x)) + model(cat(-v, x))
    (signbasisnet.py:11-20).  The concatenation is index plumbing (torch.cat); the negation and the sum are device ops."""

    def __init__(self, model):
        super().__init__()
        self.model = model

    def forward(self, v, *args, x=None):
        v = v.contiguous().float()
        neg = ops.masked_affine(v.view(-1, v.shape[-1]), scale=torch.full((v.shape[-1],), -1.0, device=v.device),
                                shift=torch.zeros(v.shape[-1], device=v.device)).view(v.shape)
        if x is not None:
            x = x.float()
            v, neg = torch.cat((v, x), dim=-1), torch.cat((neg, x), dim=-1)
        a, b = self.model(v), self.model(neg)
        if a.requires_grad or b.requires_grad:
            from . import autograd as AG
            return AG.masked_add(a.reshape(-1, a.shape[-1]), b.reshape(-1, b.shape[-1])).view(a.shape)
        return ops.masked_affine(a.contiguous().view(-1, a.shape[-1]), residual=b.contiguous().view(-1, b.shape[-1])).view(a.shape)


def group_eigenspaces(eigvals, eigvecs, decimals=5, return_plan=False):
    """The module-level preprocessing of LearningFilters/training.py:47-73 on the device (run once per graph): eigenvalues
    rounded to `decimals`, eigenvectors grouped by rounded value, P = V_s V_s^T per eigenspace, stacked by multiplicity.
    Returns {mult: [b, 1, N, N]} (views of one [n_spaces, N, N] device buffer, ascending multiplicity — the reference's
    `same_size_projs`); with return_plan also the ops.EigenspacePlan (for IGNBasisInv.forward_eigvecs)."""
    plan = ops.eigenspace_group(eigvals, decimals)
    stack = ops.eigenspace_projectors(eigvecs, plan)
    N = plan.N
    groups = {m: plan.group(stack, m).view(-1, 1, N, N) for m in plan.mults}
    return (groups, plan) if return_plan else groups
