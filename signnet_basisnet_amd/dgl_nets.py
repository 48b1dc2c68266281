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
from .dgl_deepsigns import MLP, _GINConv, _prep_mlp, _run_mlp, get_sign_inv_net


class MLPReadout(nn.Module):
    """layers/mlp_readout_layer.py:9-24: L halving Linear+ReLU layers, then Linear to the output."""

    def __init__(self, input_dim, output_dim, L=2):
        super().__init__()
        dims = [input_dim // 2 ** l for l in range(L + 1)]
        self.FC_layers = nn.ModuleList([nn.Linear(dims[l], dims[l + 1], bias=True) for l in range(L)] +
                                       [nn.Linear(dims[L], output_dim, bias=True)])
        self.L = L


class GINNet(nn.Module):
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
                pl = ops.PackedLinear(ops.pack_weight(self.embedding_p.weight.detach()), *self.embedding_p.weight.shape,
                                      self.embedding_p.bias.detach().contiguous())
                x = ops.masked_linear(p, pl, residual=x)                                              # h + embedding_p(p)   (:87-92)
                for conv in self.layers:
                    a = ops.gin_aggregate(x, plan, conv.eps)
                    x = _run_mlp(_prep_mlp(conv.apply_func, train), a, train=train)
                hg = ops.segment_pool(x, plan, "mean" if self.readout != "sum" else "add")
                fcs = self.MLP_layer.FC_layers
                for i, fc in enumerate(fcs):
                    plf = ops.PackedLinear(ops.pack_weight(fc.weight.detach()), *fc.weight.shape, fc.bias.detach().contiguous())
                    hg = ops.masked_linear(hg, plf, relu=i < len(fcs) - 1)
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
