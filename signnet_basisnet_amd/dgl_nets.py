"""The DGL tree's GIN base network on the HIP layer kernels (first item: the consumer of the sign-invariant
positional encoding in the tree that owns main_ZINC_graph_regression.py).

Mirrors GraphPrediction/nets/ZINC_graph_regression/gin_net.py:19-139 (`GINNet`) for the configuration the shipped
GIN_ZINC_LapPE_signinv_GIN.json selects — pe_init = 'lap_pe', lap_lspe = False: `h = embedding_h(h) + embedding_p(p)`
(:80-92), L x dgl GINConv(MLP(hidden, hidden, out, 2, use_bn), 'sum') (:58-66, 99-100), mean / sum readout (:126-133),
MLPReadout (layers/mlp_readout_layer.py:9-24) — with the same constructor (`net_params`), forward contract
`model(g, h, p, e, snorm_n) -> (scores, g)` and state_dict keys, and `model.sign_inv_net` (GINDeepSigns / MaskedGINDeepSigns)
attached as in the reference.  Eval, train-mode value and — with gradients enabled — the differentiable path (autograd.py).
The LSPE / random-walk variants (p_out, Whp, lapeig loss) are not built and raise.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops
from .dgl_deepsigns import MLP, _BNSite, _GINConv, _pack, _prep_mlp, _run_mlp, get_sign_inv_net


class MLPReadout(nn.Module):
    """layers/mlp_readout_layer.py:9-24: L halving Linear+ReLU layers, then Linear to the output."""

    def __init__(self, input_dim, output_dim, L=2):
        super().__init__()
        dims = [input_dim // 2 ** l for l in range(L + 1)]
        self.FC_layers = nn.ModuleList([nn.Linear(dims[l], dims[l + 1], bias=True) for l in range(L)] +
                                       [nn.Linear(dims[L], output_dim, bias=True)])
        self.L = L


class _PackCache:
    """Eval-mode cache of packed Linears / folded BatchNorms (rebuilt after train(), .to(), load_state_dict(); call
    `invalidate()` after changing parameters in place while in eval mode).  Train mode packs per call: parameters move."""

    def invalidate(self):
        self._cache = {}

    def train(self, mode=True):
        self._cache = {}
        return super().train(mode)

    def _apply(self, fn, *a, **k):
        self._cache = {}
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._cache = {}
        return super().load_state_dict(*a, **k)

    def _load_from_state_dict(self, *a, **k):
        # reached also when a PARENT module's load_state_dict recurses into this one (the override above is not)
        self._cache = {}
        return super()._load_from_state_dict(*a, **k)

    def _pk(self, lin):
        if self.training:
            return _pack(lin)
        c = self.__dict__.setdefault("_cache", {})
        if id(lin) not in c:
            c[id(lin)] = _pack(lin)
        return c[id(lin)]

    def _bn(self, bn, train):
        if train:
            return _BNSite(bn, True)
        c = self.__dict__.setdefault("_cache", {})
        if id(bn) not in c:
            c[id(bn)] = _BNSite(bn, False)
        return c[id(bn)]

    def _abde(self, L):
        """The four node Linears of a GatedGCN layer packed as one [4d, d] Linear (eval cache)."""
        c = self.__dict__.setdefault("_cache", {})
        key = ("abde", id(L))
        if key not in c:
            W = torch.cat([getattr(L, n).weight.detach() for n in "ABDE"], 0).contiguous()
            b = torch.cat([getattr(L, n).bias.detach() for n in "ABDE"], 0).contiguous()
            c[key] = ops.PackedLinear(ops.pack_weight(W), W.shape[0], W.shape[1], b)
        return c[key]

    def _mlp(self, mlp, train):
        if train:
            return _prep_mlp(mlp, True)
        c = self.__dict__.setdefault("_cache", {})
        if id(mlp) not in c:
            c[id(mlp)] = _prep_mlp(mlp, False)
        return c[id(mlp)]


class GINNet(_PackCache, nn.Module):
    def __init__(self, net_params):
        super().__init__()
        p = net_params
        hidden, out_dim = p["hidden_dim"], p["out_dim"]
        self.n_layers, self.readout, self.batch_norm = p["L"], p["readout"], p["batch_norm"]
        self.residual, self.edge_feat, self.device = p["residual"], p["edge_feat"], p["device"]
        self.pe_init, self.lap_method, self.lap_lspe = p["pe_init"], p["lap_method"], p["lap_lspe"]
        self.use_lapeig_loss, self.lambda_loss, self.alpha_loss = p["use_lapeig_loss"], p["lambda_loss"], p["alpha_loss"]
        self.pos_enc_dim = p["pos_enc_dim"]
        if self.pe_init == "rand_walk" or self.lap_lspe or self.use_lapeig_loss:
            raise NotImplementedError("HIP GINNet covers pe_init='lap_pe' / lap_lspe=False (the sign-invariant PE configs)")
        if self.readout == "max":
            raise NotImplementedError("HIP GINNet: readout 'sum' or 'mean'")
        if p.get("in_feat_dropout", 0.0) or p.get("dropout", 0.0):
            raise NotImplementedError("HIP GINNet: dropout 0.0 (as in the shipped configs)")
        if self.pe_init == "lap_pe":
            self.embedding_p = nn.Linear(self.pos_enc_dim, hidden)
        self.embedding_h = nn.Embedding(p["num_atom_type"], hidden)
        self.embedding_e = nn.Embedding(p["num_bond_type"], hidden) if self.edge_feat else nn.Linear(1, hidden)   # unused by GIN
        self.layers = nn.ModuleList(
            [_GINConv(MLP(hidden, hidden, hidden, 2, use_bn=self.batch_norm, dropout=0.0, activation="relu")) for _ in range(self.n_layers - 1)] +
            [_GINConv(MLP(hidden, hidden, out_dim, 2, use_bn=self.batch_norm, dropout=0.0, activation="relu"))])
        self.MLP_layer = MLPReadout(out_dim, 1)
        self.g = None
        if self.lap_method == "sign_inv":
            self.sign_inv_net = get_sign_inv_net(net_params)

    # ------------------------------------------------------------------
    def _plan(self, g, N):
        src, dst = g.edges()
        bnn = g.batch_num_nodes().to(src.device)
        B = int(bnn.numel())
        batch = torch.repeat_interleave(torch.arange(B, device=src.device), bnn)      # index plumbing only
        if batch.numel() != N:
            raise ValueError("batch_num_nodes does not sum to the number of feature rows")
        return batch.long(), torch.stack([src.long(), dst.long()]), B

    def forward(self, g, h, p, e, snorm_n=None):
        ops.require_cuda(h)
        if p is None or self.pe_init != "lap_pe":
            raise NotImplementedError("HIP GINNet needs the positional encoding p (pe_init='lap_pe')")
        N = h.shape[0]
        batch, ei, B = self._plan(g, N)
        plan = ops.build_plan(batch, ei, B, 0)
        hidx = h.long().reshape(N)
        p = p.contiguous().float()
        train = self.training
        if train and torch.is_grad_enabled() and any(q.requires_grad for q in self.parameters()):
            y = self._forward_grad(plan, batch, ei, B, hidx, p)
        else:
            with torch.no_grad():
                x = ops.embedding_sum(hidx, [self.embedding_h.weight])
                x = ops.masked_linear(p, self._pk(self.embedding_p), residual=x)                                              # h + embedding_p(p)   (:87-92)
                for conv in self.layers:
                    a = ops.gin_aggregate(x, plan, conv.eps)
                    x = _run_mlp(self._mlp(conv.apply_func, train), a, train=train)
                hg = ops.segment_pool(x, plan, "mean" if self.readout != "sum" else "add")
                fcs = self.MLP_layer.FC_layers
                for i, fc in enumerate(fcs):
                    hg = ops.masked_linear(hg, self._pk(fc), relu=i < len(fcs) - 1)
                y = hg
        self.g = g
        return y, g

    def _forward_grad(self, plan, batch, ei, B, hidx, p):
        from . import autograd as AG
        rplan = ops.build_plan(batch, ei.flip(0).contiguous(), B, 0)
        x = AG.embedding_sum(hidx, [self.embedding_h.weight])
        x = AG.masked_add(AG.linear(p, self.embedding_p.weight, self.embedding_p.bias), x)
        for conv in self.layers:
            mlp = conv.apply_func
            a = AG.gin_aggregate(x, conv.eps, plan, rplan)
            n = len(mlp.lins)
            for i, lin in enumerate(mlp.lins):
                a = AG.linear(a, lin.weight, lin.bias, relu=i < n - 1)
                if mlp.use_bn and i < n - 1:
                    a = AG.bn_act(a, mlp.bns[i], relu=False)
            x = a
        hg = AG.segment_pool(x, plan, "mean" if self.readout != "sum" else "add")
        fcs = self.MLP_layer.FC_layers
        for i, fc in enumerate(fcs):
            hg = AG.linear(hg, fc.weight, fc.bias, relu=i < len(fcs) - 1)
        return hg

    def loss(self, scores, targets):
        """gin_net.py:141-143 (use_lapeig_loss = False): the L1 task loss (a torch reduction over B scalars)."""
        return (scores - targets).abs().mean()


# ---------------------------------------------------------------------------------------------------------------
class GatedGCNLayer(nn.Module):
    """layers/gatedgcn_layer.py:12-81 (parameters only; the arithmetic is in GatedGCNNet.forward)."""

    def __init__(self, input_dim, output_dim, dropout, batch_norm, residual=False, graph_norm=True):
        super().__init__()
        if dropout or graph_norm:
            raise NotImplementedError("HIP GatedGCNLayer: dropout 0.0 and graph_norm=False (what gatedgcn_net.py builds for lap_pe)")
        self.in_channels, self.out_channels = input_dim, output_dim
        self.batch_norm, self.residual = batch_norm, residual and input_dim == output_dim
        for n in "ABCDE":
            setattr(self, n, nn.Linear(input_dim, output_dim, bias=True))
        self.bn_node_h = nn.BatchNorm1d(output_dim)
        self.bn_node_e = nn.BatchNorm1d(output_dim)


class GatedGCNNet(_PackCache, nn.Module):
    """nets/ZINC_graph_regression/gatedgcn_net.py:18-148 for pe_init = 'lap_pe', lap_lspe = False (the sign-invariant PE configs
    GatedGCN_ZINC_LapPE_signinv_GIN[_mask].json): embedding_h / embedding_p with `add` or `concat` + pe_proj (:93-103), edge
    embedding, L GatedGCN layers on sn_gated_aggregate_f32, mean / sum readout, MLPReadout.  Same constructor, forward contract
    `model(g, h, p, e, snorm_n) -> (scores, g)`, state_dict keys and attached `sign_inv_net` as the reference."""

    def __init__(self, net_params):
        super().__init__()
        p = net_params
        hidden, out_dim = p["hidden_dim"], p["out_dim"]
        self.n_layers, self.readout, self.batch_norm = p["L"], p["readout"], p["batch_norm"]
        self.residual, self.edge_feat, self.device = p["residual"], p["edge_feat"], p["device"]
        self.pe_init, self.lap_method, self.lap_lspe = p["pe_init"], p["lap_method"], p["lap_lspe"]
        self.use_lapeig_loss, self.lambda_loss, self.alpha_loss = p["use_lapeig_loss"], p["lambda_loss"], p["alpha_loss"]
        self.pos_enc_dim, self.pe_aggregate = p["pos_enc_dim"], p["pe_aggregate"]
        if self.pe_init != "lap_pe" or self.lap_lspe or self.use_lapeig_loss:
            raise NotImplementedError("HIP GatedGCNNet covers pe_init='lap_pe' / lap_lspe=False (the sign-invariant PE configs)")
        if self.readout == "max" or not self.edge_feat or not self.batch_norm:
            raise NotImplementedError("HIP GatedGCNNet: readout sum/mean, edge_feat=True, batch_norm=True")
        if p.get("in_feat_dropout", 0.0) or p.get("dropout", 0.0):
            raise NotImplementedError("HIP GatedGCNNet: dropout 0.0 (as in the shipped configs)")
        self.embedding_p = nn.Linear(self.pos_enc_dim, hidden)
        self.embedding_h = nn.Embedding(p["num_atom_type"], hidden)
        self.embedding_e = nn.Embedding(p["num_bond_type"], hidden)
        self.layers = nn.ModuleList([GatedGCNLayer(hidden, hidden, 0.0, True, residual=self.residual, graph_norm=False)
                                     for _ in range(self.n_layers - 1)] +
                                    [GatedGCNLayer(hidden, out_dim, 0.0, True, residual=self.residual, graph_norm=False)])
        self.MLP_layer = MLPReadout(out_dim, 1)
        self.g = None
        if self.lap_method == "sign_inv":
            self.sign_inv_net = get_sign_inv_net(net_params)
        if self.pe_aggregate == "concat":
            self.pe_proj = nn.Linear(2 * hidden, hidden)

    _plan = GINNet._plan

    def forward(self, g, h, p, e, snorm_n=None):
        ops.require_cuda(h)
        if p is None:
            raise NotImplementedError("HIP GatedGCNNet needs the positional encoding p")
        N = h.shape[0]
        batch, ei, B = self._plan(g, N)
        plan = ops.build_plan(batch, ei, B, 0)
        hidx, eidx = h.long().reshape(N), e.long().reshape(-1)
        p = p.contiguous().float()
        train = self.training
        if train and torch.is_grad_enabled() and any(q.requires_grad for q in self.parameters()):
            y = self._forward_grad(plan, batch, ei, B, hidx, p, eidx)
        else:
            with torch.no_grad():
                y = self._forward_value(plan, hidx, p, eidx, train)
        self.g = g
        return y, g

    def _forward_value(self, plan, hidx, p, eidx, train):
        x = ops.embedding_sum(hidx, [self.embedding_h.weight])
        if self.pe_aggregate == "concat":
            pp = ops.masked_linear(p, self._pk(self.embedding_p))
            x = ops.masked_linear(torch.cat([x, pp], dim=1), self._pk(self.pe_proj))                  # (:96-98)
        else:
            x = ops.masked_linear(p, self._pk(self.embedding_p), residual=x)                         # (:100-101)
        e = ops.embedding_sum(eidx, [self.embedding_e.weight])
        for L in self.layers:
            if not train:
                # eval: A, B, D, E as ONE GEMM ([4d, d] weight, column blocks of the result go to the gather kernel by stride),
                # BatchNorm (folded) + ReLU + residual of both h and e fused into the gather pass: 3 launches per layer
                d = L.out_channels
                Y = ops.masked_linear(x, self._abde(L))
                Ce = ops.masked_linear(e, self._pk(L.C))
                sh, se = self._bn(L.bn_node_h, False), self._bn(L.bn_node_e, False)
                x, e = ops.gated_aggregate(Y[:, 0:d], Y[:, d:2 * d], Y[:, 2 * d:3 * d], Y[:, 3 * d:4 * d], Ce, plan,
                                           epilogue=(sh.scale, sh.shift, se.scale, se.shift, x if L.residual else None,
                                                     e if L.residual else None))
                continue
            Ah, Bh, Dh, Eh = (ops.masked_linear(x, self._pk(getattr(L, n))) for n in "ABDE")
            Ce = ops.masked_linear(e, self._pk(L.C))
            h2, e2 = ops.gated_aggregate(Ah, Bh, Dh, Eh, Ce, plan)
            sh, se = self._bn(L.bn_node_h, train), self._bn(L.bn_node_e, train)
            sc, sf = sh.affine(h2, train)
            x = ops.masked_affine(h2, scale=sc, shift=sf, relu=True, residual=x if L.residual else None)
            sc, sf = se.affine(e2, train)
            e = ops.masked_affine(e2, scale=sc, shift=sf, relu=True, residual=e if L.residual else None)
        hg = ops.segment_pool(x, plan, "mean" if self.readout != "sum" else "add")
        fcs = self.MLP_layer.FC_layers
        for i, fc in enumerate(fcs):
            hg = ops.masked_linear(hg, self._pk(fc), relu=i < len(fcs) - 1)
        return hg

    def _forward_grad(self, plan, batch, ei, B, hidx, p, eidx):
        from . import autograd as AG
        rplan = ops.build_plan(batch, ei.flip(0).contiguous(), B, 0)
        x = AG.embedding_sum(hidx, [self.embedding_h.weight])
        pp = AG.linear(p, self.embedding_p.weight, self.embedding_p.bias)
        if self.pe_aggregate == "concat":
            x = AG.linear(torch.cat([x, pp], dim=1), self.pe_proj.weight, self.pe_proj.bias)
        else:
            x = AG.masked_add(pp, x)
        e = AG.embedding_sum(eidx, [self.embedding_e.weight])
        for L in self.layers:
            Ah, Bh, Dh, Eh = (AG.linear(x, getattr(L, n).weight, getattr(L, n).bias) for n in "ABDE")
            Ce = AG.linear(e, L.C.weight, L.C.bias)
            h2, e2 = AG.gated_aggregate(Ah, Bh, Dh, Eh, Ce, plan, rplan)
            x = AG.bn_act(h2, L.bn_node_h, relu=True, residual=x if L.residual else None)
            e = AG.bn_act(e2, L.bn_node_e, relu=True, residual=e if L.residual else None)
        hg = AG.segment_pool(x, plan, "mean" if self.readout != "sum" else "add")
        fcs = self.MLP_layer.FC_layers
        for i, fc in enumerate(fcs):
            hg = AG.linear(hg, fc.weight, fc.bias, relu=i < len(fcs) - 1)
        return hg

    def loss(self, scores, targets):
        """gatedgcn_net.py:150-152 (use_lapeig_loss = False): the L1 task loss."""
        return (scores - targets).abs().mean()
