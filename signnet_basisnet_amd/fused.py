"""Host side of the fused whole-stage kernels (csrc/fused_*.hip): parameter descriptors
(ctypes mirrors of the structs in include/signnet_hip.h) built once per eval-mode model."""
from __future__ import annotations

import ctypes as C

import torch

from . import ops
from ._lib import check, lib, ptr, stream

PHI_MAX_LAYERS = 16


class _PhiLayer(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("w1p", "bn0_scale", "bn0_shift", "w2p", "bias2", "bn_scale", "bn_shift", "eps")]


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
        if not (0 < d <= 128 and 1 <= L <= PHI_MAX_LAYERS):
            raise ValueError("fused phi supports hidden width <= 128 and <= 16 layers")
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
        P.l0_w2 = hold(ops.pad_vec(w2[:, 0], dp)) if hid0 == 1 else hold(ops.pack_weight(w2))
        b2 = c0.nn.layers[1].bias
        P.l0_bias2 = hold(ops.pad_vec(b2, dp)) if b2 is not None else None
        s, h = ops.bn_fold(norms[0].bn, dp)
        P.l0_bn_scale, P.l0_bn_shift = hold(s), hold(h)
        P.l0_eps = hold(c0.layer.eps.detach())
        for l in range(1, L):
            c, Lp = convs[l], P.layers[l - 1]
            Lp.w1p = hold(ops.pack_weight(c.nn.layers[0].weight.detach()))
            s, h = ops.bn_fold(c.nn.norms[0].bn, dp)
            Lp.bn0_scale, Lp.bn0_shift = hold(s), hold(h)
            Lp.w2p = hold(ops.pack_weight(c.nn.layers[1].weight.detach()))
            b2 = c.nn.layers[1].bias
            Lp.bias2 = hold(ops.pad_vec(b2, dp)) if b2 is not None else None
            s, h = ops.bn_fold(norms[l].bn, dp)
            Lp.bn_scale, Lp.bn_shift = hold(s), hold(h)
            Lp.eps = hold(c.layer.eps.detach())
        self.params = P

    def run(self, plan: ops.GraphPlan, bins: ops.Bins, eigen_vectors, K: int, out=None):
        """phi(x)+phi(-x) -> [N, K, d]; rows of invalid slots are left untouched (zero if `out` is None)."""
        ev = eigen_vectors
        if ev.dtype != torch.float32 or not ev.is_contiguous():
            raise ValueError("eigen_vectors must be contiguous float32")
        if out is None:
            out = torch.zeros(plan.N, K, self.d, dtype=torch.float32, device=ev.device)
        with ops._span("sn_phi_fused_f32"):
            check(lib().sn_phi_fused_f32(C.byref(self.params), ptr(ev), ptr(plan.graph_ptr), ptr(plan.node_graph),
                                         ptr(plan.evoff), ptr(plan.rowptr), ptr(plan.col), ptr(bins.node),
                                         ptr(bins.slot), ptr(bins.meta), bins.max_bins, K, ptr(out), stream()),
                  "sn_phi_fused_f32")
        return out
