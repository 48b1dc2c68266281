"""Host side of the fused whole-stage kernels (csrc/fused_*.hip): parameter descriptors
(ctypes mirrors of the structs in include/signnet_hip.h) built once per eval-mode model."""
from __future__ import annotations

import ctypes as C

import torch

from . import ops
from ._lib import check, lib, ptr, stream

PHI_MAX_LAYERS = 16


class _PhiLayer(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("w1s", "w2s", "eps")]


class _PhiParams(C.Structure):
    _fields_ = [("d", C.c_int), ("n_layers", C.c_int), ("hid0", C.c_int), ("reserved", C.c_int)] + \
               [(n, C.c_void_p) for n in ("l0_w1", "l0_bn0_scale", "l0_bn0_shift", "l0_w2", "l0_bias2", "l0_bn_scale",
                                          "l0_bn_shift", "l0_eps")] + \
               [("layers", _PhiLayer * (PHI_MAX_LAYERS - 1))]


class PhiPlan:
    """Packed parameters of a GNN3d (phi) stack for sn_phi_fused_f32.  Keeps the device tensors alive."""

    def __init__(self, phi_module):
        convs, norms = phi_module.convs, phi_module.norms
        L = len(convs)
        d = convs[0].nn.layers[1].weight.shape[0]
        if not (0 < d <= 128 and d % 4 == 0 and 1 <= L <= PHI_MAX_LAYERS):
            raise ValueError("fused phi supports hidden width <= 128 (a multiple of 4) and <= 16 layers")
        dp = 16 * ((d + 15) // 16)
        self.d, self.L, self.dp = d, L, dp
        keep = self._keep = []

        def hold(t):
            keep.append(t)
            return t.data_ptr()

        P = _PhiParams()
        P.d, P.n_layers = d, L
        c0 = convs[0]
        w1 = c0.nn.layers[0].weight.detach()          # [hid0, 1]
        hid0 = w1.shape[0]
        if hid0 not in (1, d):
            raise ValueError("fused phi: first hidden width must be 1 or d")
        P.hid0 = hid0
        hp = 16 * ((hid0 + 15) // 16)
        P.l0_w1 = hold(ops.pad_vec(w1[:, 0], hp))
        s, h = ops.bn_fold(c0.nn.norms[0].bn, hp)
        P.l0_bn0_scale, P.l0_bn0_shift = hold(s), hold(h)
        w2 = c0.nn.layers[1].weight.detach()          # [d, hid0]
        b2 = c0.nn.layers[1].bias
        s, h = ops.bn_fold(norms[0].bn, dp)
        if hid0 == 1:
            P.l0_w2 = hold(ops.pad_vec(w2[:, 0], dp))
            P.l0_bias2 = hold(ops.pad_vec(b2, dp)) if b2 is not None else None
            P.l0_bn_scale, P.l0_bn_shift = hold(s), hold(h)
        else:                                          # a [d, d] Linear: split-packed with its epilogue (bias, BN scale, BN shift)
            P.l0_w2 = hold(ops.pack_split(w2, None if b2 is None else b2.detach(), s, h))
        P.l0_eps = hold(c0.layer.eps.detach())
        for l in range(1, L):
            c, Lp = convs[l], P.layers[l - 1]
            s, h = ops.bn_fold(c.nn.norms[0].bn, dp)
            Lp.w1s = hold(ops.pack_split(c.nn.layers[0].weight.detach(), s, h, None))
            b2 = c.nn.layers[1].bias
            s, h = ops.bn_fold(norms[l].bn, dp)
            Lp.w2s = hold(ops.pack_split(c.nn.layers[1].weight.detach(), None if b2 is None else b2.detach(), s, h))
            Lp.eps = hold(c.layer.eps.detach())
        self.params = P

    def run(self, plan: ops.GraphPlan, eigen_vectors, K: int, out=None, zero_invalid=True):
        """phi(x)+phi(-x) -> [N, K, d]; rows of invalid slots are left untouched (zero if `out` is None and
        zero_invalid; the fused rho stage never reads them, so the forward skips the 24 MB memset)."""
        ev = eigen_vectors
        if ev.dtype != torch.float32 or not ev.is_contiguous():
            raise ValueError("eigen_vectors must be contiguous float32")
        if out is None:
            alloc = torch.zeros if zero_invalid else torch.empty
            out = alloc(plan.N, K, self.d, dtype=torch.float32, device=ev.device)
        with ops._span("sn_phi_fused_f32"):
            check(lib().sn_phi_fused_f32(C.byref(self.params), ptr(ev), ptr(plan.graph_ptr), ptr(plan.evoff),
                                         ptr(plan.rowptr), ptr(plan.col), C.byref(plan.bins.cstruct), plan.kmax, K,
                                         ptr(out), stream()), "sn_phi_fused_f32")
        return out


RHO_MAX_LAYERS = 8


class _RhoLayer(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("wq", "wk", "wv", "wfc", "ln1_g", "ln1_b", "w1", "w2", "ln2_g", "ln2_b")]


class _RhoParams(C.Structure):
    _fields_ = [("d", C.c_int), ("n_layers", C.c_int), ("heads", C.c_int), ("has_pos", C.c_int),
                ("ln_eps", C.c_float), ("head_pad", C.c_int)] + \
               [(n, C.c_void_p) for n in ("pe_w1", "pe_bn0_scale", "pe_bn0_shift", "pe_w2", "pe_bn1_scale", "pe_bn1_shift")] + \
               [("layers", _RhoLayer * RHO_MAX_LAYERS)]


class RhoPlan:
    """Packed parameters of a SetTransformer (rho) for sn_rho_fused_f32."""

    def __init__(self, rho_module, eigen_encoder, heads, ln_eps):
        tls = rho_module.transformer_layers
        d = rho_module.out[0].weight.shape[1]
        if not (0 < d <= 128 and len(tls) <= RHO_MAX_LAYERS and heads == 4 and d % heads == 0):
            raise ValueError("fused rho supports hidden width <= 128 (divisible by 4 heads) and <= 8 layers")
        dk = d // heads
        self._keep = []
        hp_want = 0 if dk % 16 == 0 else 16 * ((dk + 15) // 16)
        self.d, self.head_pad = d, hp_want
        self.params = self._build(rho_module, eigen_encoder, heads, ln_eps, 0)
        # a second, head-padded packing for batches whose nodes have <= 16 slots (chosen per call in run())
        self.params_hp = self._build(rho_module, eigen_encoder, heads, ln_eps, hp_want) if hp_want else None

    def _build(self, rho_module, eigen_encoder, heads, ln_eps, hp):
        tls = rho_module.transformer_layers
        d = self.d
        dk = d // heads
        # hp > 0: head width not a multiple of 16 (e.g. the Alchemy config's 108/4 = 27): pack every head into hp = 16 or 32 channels
        # (zero padded), permuting the q/k/v output rows and fc's input columns, so that the kernel's register attention
        # (one 16-channel MFMA chunk never straddles two heads) applies; all other tensors are zero padded to heads*hp
        dp = heads * hp if hp else 16 * ((d + 15) // 16)
        keep = self._keep

        def hold(t):
            keep.append(t)
            return t.data_ptr()

        dev = rho_module.out[0].weight.device
        if hp:
            hidx = torch.tensor([h * hp + j for h in range(heads) for j in range(dk)], device=dev)     # padded slot of channel h*dk+j

        def wpad(W, head_rows=False, head_cols=False):
            """Weight as the kernel streams it: [dp, dp] zero padded, optionally with head-permuted rows / columns."""
            W = W.detach()
            if not hp:
                return W
            out = torch.zeros(dp, dp, dtype=torch.float32, device=dev)
            rows = hidx if head_rows else torch.arange(W.shape[0], device=dev)
            cols = hidx if head_cols else torch.arange(W.shape[1], device=dev)
            out[rows[:, None], cols[None, :]] = W
            return out

        def vpad(v):
            return ops.pad_vec(v.detach(), dp)

        P = _RhoParams()
        P.d, P.n_layers, P.heads, P.ln_eps, P.head_pad = d, len(tls), heads, float(ln_eps), hp
        P.has_pos = 1 if eigen_encoder is not None else 0
        if eigen_encoder is not None:
            ee = eigen_encoder
            P.pe_w1 = hold(ops.pad_vec(ee.layers[0].weight.detach(), 4))
            s, h = ops.bn_fold(ee.norms[0].bn, 4)
            P.pe_bn0_scale, P.pe_bn0_shift = hold(s), hold(h)
            P.pe_w2 = hold(ops.pad_vec(ee.layers[1].weight.detach()[:, 0], dp))
            s, h = ops.bn_fold(ee.norms[1].bn, dp)
            P.pe_bn1_scale, P.pe_bn1_shift = hold(s), hold(h)
        for l, tl in enumerate(tls):
            a, f, Lp = tl.slf_attn, tl.pos_ffn, P.layers[l]
            Lp.wq = hold(ops.pack_split(wpad(a.w_qs.weight, head_rows=True)))
            Lp.wk = hold(ops.pack_split(wpad(a.w_ks.weight, head_rows=True)))
            Lp.wv = hold(ops.pack_split(wpad(a.w_vs.weight, head_rows=True)))
            Lp.wfc = hold(ops.pack_split(wpad(a.fc.weight, head_cols=True)))
            Lp.ln1_g, Lp.ln1_b = hold(vpad(a.norm.ln.weight)), hold(vpad(a.norm.ln.bias))
            Lp.w1 = hold(ops.pack_split(wpad(f.w_1.weight), vpad(f.w_1.bias) if hp else f.w_1.bias.detach()))
            Lp.w2 = hold(ops.pack_split(wpad(f.w_2.weight), vpad(f.w_2.bias) if hp else f.w_2.bias.detach()))
            Lp.ln2_g, Lp.ln2_b = hold(vpad(f.norm.ln.weight)), hold(vpad(f.norm.ln.bias))
        return P

    def run(self, plan: ops.GraphPlan, x, eigen_values, K: int):
        """x [N*K, d] -> sum over valid slots of the encoder output, [N, d]."""
        out = torch.empty(plan.N, self.d, dtype=torch.float32, device=x.device)
        kcap = min(plan.kmax, K) if plan.kmax > 0 else K
        params = self.params_hp if (self.params_hp is not None and kcap <= 16) else self.params
        with ops._span("sn_rho_fused_f32"):
            check(lib().sn_rho_fused_f32(C.byref(params), ptr(x), ptr(eigen_values), ptr(plan.graph_ptr), plan.B,
                                         plan.N, C.byref(plan.bins.cstruct), plan.kmax, K, ptr(out), stream()),
                  "sn_rho_fused_f32")
        return out


GNN_MAX_LAYERS = 16
GNN_MAX_NODES = 64
GNN_MAX_EDGES = 192      # in-edges of one graph the fused GINE stage stages in LDS (GNN_EMAX, csrc/fused_gnn.hip)


class _GnnLayer(C.Structure):
    _fields_ = [("etab", C.c_void_p * 10)] + \
               [(n, C.c_void_p) for n in ("ew", "e_scale", "e_shift", "w1s", "w2s", "eps")]


class _GnnParams(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("d", "n_layers", "n_out", "reserved", "node_discrete", "node_nf", "edge_discrete",
                                       "edge_nf", "node_vocab", "edge_vocab")] + \
               [("ntab", C.c_void_p * 10)] + \
               [(n, C.c_void_p) for n in ("nw", "n_scale", "n_shift", "rho_out_w", "lin_a", "lin_b", "head_w1", "head_w2")] + \
               [("layers", _GnnLayer * GNN_MAX_LAYERS)]


class GnnPlan:
    """Packed parameters of rho.out + the GINE `GNN` module for sn_gnn_fused_f32."""

    def __init__(self, rho_out, gnn, node_feat, edge_feat):
        d = gnn.linear.weight.shape[0]
        L = len(gnn.convs)
        n_out = gnn.output_encoder.layers[1].weight.shape[0]
        if not (0 < d <= 128 and L <= GNN_MAX_LAYERS and 1 <= n_out <= 16 and gnn.pooling == "add"):
            raise ValueError("fused gnn supports hidden width <= 128, <= 16 layers, n_out <= 16, add pooling")
        if (node_feat is not None and node_feat > 16) or (edge_feat is not None and edge_feat > 16):
            raise ValueError("fused gnn supports at most 16 continuous node/edge features")
        dp = 16 * ((d + 15) // 16)
        self.d, self.n_out = d, n_out
        self.node_discrete, self.edge_discrete = node_feat is None, edge_feat is None
        keep = self._keep = []

        def hold(t):
            keep.append(t)
            return t.data_ptr()

        P = _GnnParams()
        P.d, P.n_layers, P.n_out = d, L, n_out
        P.node_discrete, P.edge_discrete = int(self.node_discrete), int(self.edge_discrete)
        P.node_nf = 1 if self.node_discrete else int(node_feat)     # discrete column count is set per call
        P.edge_nf = 1 if self.edge_discrete else int(edge_feat)
        P.node_vocab = P.edge_vocab = 0
        if self.node_discrete:
            P.node_vocab = gnn.input_encoder.embeddings[0].weight.shape[0]
            for f, e in enumerate(gnn.input_encoder.embeddings):
                P.ntab[f] = hold(e.weight.detach())
        else:
            w = gnn.input_encoder.layers[0].weight.detach()            # [d, F]
            wp = torch.zeros(dp, w.shape[1], dtype=torch.float32, device=w.device)
            wp[:d].copy_(w)
            P.nw = hold(wp)
            s, h = ops.bn_fold(gnn.input_encoder.norms[0], dp)
            P.n_scale, P.n_shift = hold(s), hold(h)
        # rho.out (Linear(no bias) + eval BatchNorm, sign_net.py:71) and the pos half of `linear` (model.py:39-40) are two affine maps
        # with nothing in between: folded once into  W' = W_pos . diag(scale) . W_out,  b' = W_pos . shift + b  on the device
        # (fp32 MFMA GEMM kernels; rounding differs from the two-step evaluation by ~1e-7 relative), which removes one whole
        # dependent GEMM stage and a barrier from the per-graph latency chain of the kernel
        W = gnn.linear.weight.detach()                                  # [d, 2d]: x part | pos part
        Wp = W[:, d:].contiguous()
        s, h = ops.bn_fold(rho_out[1], d)
        A = ops.masked_affine(Wp, scale=s, shift=torch.zeros_like(s))   # W_pos . diag(scale)
        Wt = rho_out[0].weight.detach().t().contiguous()                # y = x . Wt^T = x . W_out
        Wf = ops.masked_linear(A, ops.PackedLinear(ops.pack_weight(Wt), d, d, None))
        bf = ops.masked_linear(h.view(1, d).contiguous(), ops.PackedLinear(ops.pack_weight(Wp), d, d, gnn.linear.bias.detach().contiguous()))
        P.rho_out_w = None
        P.lin_a = hold(ops.pack_split(W[:, :d]))
        P.lin_b = hold(ops.pack_split(Wf, bf.view(-1)))
        oe = gnn.output_encoder
        s, h = ops.bn_fold(oe.norms[0], dp)
        P.head_w1 = hold(ops.pack_split(oe.layers[0].weight.detach(), s, h))
        P.head_w2 = hold(ops.pack_split(oe.layers[1].weight.detach(), oe.layers[1].bias.detach()))
        for l, (enc, conv, norm) in enumerate(zip(gnn.edge_encoders, gnn.convs, gnn.norms)):
            Lp = P.layers[l]
            if self.edge_discrete:
                P.edge_vocab = enc.embeddings[0].weight.shape[0]
                for f, e in enumerate(enc.embeddings):
                    Lp.etab[f] = hold(e.weight.detach())
            else:
                w = enc.layers[0].weight.detach()                      # [d, F_e]
                wp = torch.zeros(dp, w.shape[1], dtype=torch.float32, device=w.device)
                wp[:d].copy_(w)
                Lp.ew = hold(wp)
                s, h = ops.bn_fold(enc.norms[0], dp)
                Lp.e_scale, Lp.e_shift = hold(s), hold(h)
            s, h = ops.bn_fold(conv.nn.norms[0], dp)
            Lp.w1s = hold(ops.pack_split(conv.nn.layers[0].weight.detach(), s, h))
            s, h = ops.bn_fold(norm, dp)
            Lp.w2s = hold(ops.pack_split(conv.nn.layers[1].weight.detach(), s, h))
            Lp.eps = hold(conv.layer.eps.detach())
        self.params = P

    def run(self, plan: ops.GraphPlan, x, edge_attr, rho_sum, flags_host=None):
        """-> model output [B, n_out] (graphs with more than 64 nodes set plan.status[3]).  flags_host: optional pinned
        int32 host tensor (>= plan.flags.numel()) that the kernel's last workgroup fills with plan.flags."""
        P = self.params
        if self.node_discrete:
            if x.dtype != torch.int64:
                raise ValueError("discrete node features must be int64")
            x = x.reshape(plan.N, -1).contiguous()
            P.node_nf = x.shape[1]
        else:
            if x.dtype != torch.float32:
                raise ValueError("continuous node features must be float32")
            x = x.reshape(plan.N, -1).contiguous()
        if self.edge_discrete:
            if edge_attr.dtype != torch.int64:
                raise ValueError("discrete edge features must be int64")
            edge_attr = (edge_attr.reshape(plan.E, -1) if plan.E else edge_attr.reshape(0, 1)).contiguous()   # (no edges: E = 0)
            P.edge_nf = edge_attr.shape[1] if plan.E else 1
        else:
            edge_attr = (edge_attr.reshape(plan.E, -1) if plan.E else edge_attr.reshape(0, max(1, int(P.edge_nf)))).contiguous()
        if plan.E == 0:      # a batch without edges (single-node graphs): nothing is read, but the entry point wants a pointer
            edge_attr = torch.zeros(1, max(1, int(P.edge_nf)), dtype=edge_attr.dtype, device=rho_sum.device)
        if P.node_nf > (10 if self.node_discrete else 16) or P.edge_nf > (10 if self.edge_discrete else 16):
            raise ValueError("too many feature columns for the fused gnn kernel")
        y = torch.empty(plan.B, self.n_out, dtype=torch.float32, device=rho_sum.device)
        with ops._span("sn_gnn_fused_f32"):
            check(lib().sn_gnn_fused_f32(C.byref(P), ptr(x), x.shape[1], ptr(edge_attr),
                                         edge_attr.shape[1] if edge_attr.dim() > 1 else 1, ptr(rho_sum),
                                         ptr(plan.graph_ptr), plan.B, ptr(plan.rowptr), ptr(plan.col),
                                         ptr(plan.eperm), ptr(plan.status), ptr(y),
                                         ptr(plan.flags), plan.flags.numel(), ptr(flags_host), stream()),
                  "sn_gnn_fused_f32")
        return y
