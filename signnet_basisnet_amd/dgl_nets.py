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

import ctypes as C

import torch
import torch.nn as nn

from . import ops
from ._lib import check, lib, ptr, stream
from .dgl_deepsigns import MLP, cached_plan, _await_side, _max_nodes, _max_in_edges, _node_counts, _BNSite, _GINConv, _pack, _prep_mlp, _run_mlp, get_sign_inv_net


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


class _FusedGin:
    """Packed eval-mode parameters of a GINNet for sn_gin_net_fused_f32: embedding_h + embedding_p, every GIN layer, the readout and
    MLPReadout in ONE launch — one workgroup per graph on the stage kernel of the PyG tree's GINE net (csrc/fused_gnn.hip, DGL mode).
    `ok` False: shapes the kernel does not take (width > 128, more than 16 layers, MLPs that are not the reference's
    Linear-ReLU-BN-Linear, a readout other than the 3-Linear MLPReadout to one score) — the layer path serves those."""

    def __init__(self, net):
        from .fused import _GnnParams, GNN_MAX_LAYERS
        from .dgl_deepsigns import _fold_bn_before, _pad_mat
        self.ok = False
        convs, fcs = list(net.layers), list(net.MLP_layer.FC_layers)
        hid, kp = net.embedding_h.weight.shape[1], net.embedding_p.weight.shape[1]
        widths = [hid] + [w for c in convs for w in (c.apply_func.lins[0].weight.shape[0], c.apply_func.lins[-1].weight.shape[0])] + \
                 [fc.weight.shape[0] for fc in fcs[:-1]]
        dmax = max(widths + [kp])
        if dmax > 128 or len(convs) > GNN_MAX_LAYERS or len(convs) < 1 or len(fcs) != 3 or fcs[2].weight.shape[0] != 1:
            return
        if any(len(c.apply_func.lins) != 2 for c in convs):
            return
        dp = 64 if dmax <= 64 else (96 if dmax <= 96 else 128)
        dev = net.embedding_h.weight.device
        keep = self._keep = []

        def hold(t):
            keep.append(t)
            return t.data_ptr()

        ones = torch.ones(dp, dtype=torch.float32, device=dev)
        P = _GnnParams()
        P.d, P.n_layers, P.n_out = dp, len(convs), 1
        P.node_discrete, P.node_nf, P.edge_discrete, P.edge_nf = 1, 1, 0, 0
        P.node_vocab, P.edge_vocab = net.embedding_h.weight.shape[0], 0
        E = torch.zeros(net.embedding_h.weight.shape[0], dp, dtype=torch.float32, device=dev)
        E[:, :hid].copy_(net.embedding_h.weight.detach())
        P.ntab[0] = hold(E)
        P.rho_out_w = None
        P.lin_a = hold(ops.pack_split(torch.eye(dp, dtype=torch.float32, device=dev)))
        P.lin_b = hold(ops.pack_split(_pad_mat(net.embedding_p.weight, dp), ops.pad_vec(net.embedding_p.bias, dp)))
        for l, conv in enumerate(convs):
            m = conv.apply_func
            site = net._bn(m.bns[0], False) if m.use_bn else None        # the BatchNorm between the ReLU and the second Linear: folded
            W2, b2 = _fold_bn_before(m.lins[1], site)
            Lp = P.layers[l]
            Lp.w1s = hold(ops.pack_split(_pad_mat(m.lins[0].weight, dp), ones, ops.pad_vec(m.lins[0].bias, dp)))
            Lp.w2s = hold(ops.pack_split(_pad_mat(W2, dp), ones, ops.pad_vec(b2, dp)))
            Lp.eps = hold(conv.eps.detach().float().reshape(1).contiguous())
        P.head_w1 = hold(ops.pack_split(_pad_mat(fcs[0].weight, dp), ones, ops.pad_vec(fcs[0].bias, dp)))
        self.head_mid = hold(ops.pack_split(_pad_mat(fcs[1].weight, dp), ones, ops.pad_vec(fcs[1].bias, dp)))
        P.head_w2 = hold(ops.pack_split(_pad_mat(fcs[2].weight, dp), ops.pad_vec(fcs[2].bias, dp)))
        self.params, self.kp, self.pool_mean = P, kp, 0 if net.readout == "sum" else 1
        self.ok = True

    def run(self, plan, hidx, p):
        """atom types [N] int64, positional encoding [N, k] -> scores [B, 1]"""
        y = torch.empty(plan.B, 1, dtype=torch.float32, device=p.device)
        with ops._span("sn_gin_net_fused_f32"):
            check(lib().sn_gin_net_fused_f32(C.byref(self.params), self.head_mid, self.pool_mean, ptr(hidx), ptr(p), p.shape[1], self.kp,
                                             ptr(plan.graph_ptr), plan.B, ptr(plan.rowptr), ptr(plan.col), ptr(plan.eperm),
                                             ptr(plan.status), ptr(y), ptr(plan.status), 8, stream()), "sn_gin_net_fused_f32")
        return y


class GINNet(_PackCache, nn.Module):
    fused_stages = True      # eval: embeddings, every layer and the readout in ONE launch (sn_gin_net_fused_f32); False: the layer path

    def _fused_gin(self, g):
        """The packed stage-kernel parameters if this batch can take the one-launch path, else None (as GatedGCNNet._fused_gated: graph
        sizes come with the batch object; a batch beyond the kernel's limits that still reaches it gets NaN scores + check_last())."""
        if self.training or not self.fused_stages or self.pe_init != "lap_pe":
            return None
        c = self.__dict__.setdefault("_cache", {})
        if "fused_gin" not in c:
            c["fused_gin"] = _FusedGin(self)
        fz = c["fused_gin"]
        if not fz.ok:
            return None
        me = _max_in_edges(g)
        return fz if 0 < _max_nodes(g) <= 64 and (me is None or me <= 192) else None

    def check_last(self):
        """Raise what the last one-launch eval forward flagged on the device (its scores are NaN in that case): an atom type outside
        the embedding table (IndexError, as nn.Embedding), a malformed batch, or a graph the stage kernel could not hold.  One host
        sync; also run by train().  The layer path (fused_stages = False, train mode) raises immediately instead."""
        plan, self._last_plan = getattr(self, "_last_plan", None), None
        if plan is not None:
            st = plan.status.tolist()
            if st[3] or st[5]:
                plan.status[3:6].zero_()         # (the plan is cached on the graph object: a later forward of the same graph starts clean)
            if st[0]:
                raise ValueError(f"{type(self).__name__}: the last batch is malformed (sn_batch_plan status {st[0]})")
            if (st[3] & 4) or st[5]:
                raise IndexError(ops.EMBEDDING_INDEX_ERROR)
            if st[3] & 8:
                raise RuntimeError(f"{type(self).__name__}: a graph of the last batch has no nodes; its score is NaN — set fused_stages = False "
                                   "for such batches")
            if st[3] & 3:
                raise RuntimeError(f"{type(self).__name__}: a graph of the last batch has more than 64 nodes or 192 in-edges; its score is "
                                   "NaN — set fused_stages = False for such batches")

    def train(self, mode=True):
        self.check_last()
        return super().train(mode)

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
        # (the node total comes from the graph object's cached host counts: repeat_interleave without output_size would read the sum
        #  back from the device — a host wait per batch, and not recordable in a HIP graph)
        if _node_counts(g)[1] != N:
            raise ValueError("batch_num_nodes does not sum to the number of feature rows")
        batch = torch.repeat_interleave(torch.arange(B, device=src.device), bnn, output_size=N)      # index plumbing only
        return batch.long(), torch.stack([src.long(), dst.long()]), B

    def _gin_padded(self):
        """Eval cache: the net's parameters with every channel axis zero-padded to a multiple of 4 (pad rows / columns of the weights,
        pad entries of biases and folded BatchNorms are zero, so pad channels stay exactly 0 through ReLU and the aggregations)."""
        c = self.__dict__.setdefault("_cache", {})
        if "gin_padded" in c:
            return c["gin_padded"]
        dev = self.embedding_h.weight.device
        up = lambda k: (k + 3) // 4 * 4

        def padw(W, b, rows, cols):
            Wp = torch.zeros(rows, cols, dtype=torch.float32, device=dev)
            Wp[:W.shape[0], :W.shape[1]] = W.detach().float()
            bp = torch.zeros(rows, dtype=torch.float32, device=dev)
            if b is not None:
                bp[:b.shape[0]] = b.detach().float()
            return ops.PackedLinear(ops.pack_weight(Wp), rows, cols, bp)

        def padv(v, n_):
            out = torch.zeros(n_, dtype=torch.float32, device=dev)
            out[:v.shape[0]] = v
            return out

        class _Site:          # a folded BatchNorm on the padded channels (what _run_mlp reads)
            def __init__(self, sc, sh):
                self.scale, self.shift = sc, sh

        hid = self.embedding_h.weight.shape[1]
        E = torch.zeros(self.embedding_h.weight.shape[0], up(hid), dtype=torch.float32, device=dev)
        E[:, :hid] = self.embedding_h.weight.detach().float()
        P = {"emb_h": E, "emb_p": padw(self.embedding_p.weight, self.embedding_p.bias, up(hid), self.embedding_p.weight.shape[1])}
        layers, keep = [], []
        for conv in self.layers:
            m = conv.apply_func
            chain = []
            for i, lin in enumerate(m.lins):
                site = None
                if m.use_bn and i < len(m.lins) - 1:
                    s0 = self._bn(m.bns[i], False)
                    site = _Site(padv(s0.scale, up(lin.weight.shape[0])), padv(s0.shift, up(lin.weight.shape[0])))
                chain.append((padw(lin.weight, lin.bias, up(lin.weight.shape[0]), up(lin.weight.shape[1])), site))
            # the whole MLP as one sn_mlp_chain_f32 launch: Linear -> ReLU -> [BatchNorm folded into the next Linear] ... -> Linear
            fused = None
            widths = [up(m.lins[0].weight.shape[1])] + [up(l.weight.shape[0]) for l in m.lins]
            rp = max(48, 16 * ((max(widths) + 15) // 16))
            if rp <= 128 and len(m.lins) <= 16:
                from .dgl_deepsigns import _fold_bn_before, _pad_mat
                import ctypes as C_
                ws = []
                for i, lin in enumerate(m.lins):
                    site = self._bn(m.bns[i - 1], False) if (m.use_bn and i > 0) else None
                    Wf, bf = _fold_bn_before(lin, site)
                    ws.append(ops.pack_split(_pad_mat(Wf, rp), ops.pad_vec(bf, rp), None, None))
                keep.extend(ws)
                fused = ((C_.c_void_p * len(ws))(*[w.data_ptr() for w in ws]), len(ws), rp, up(m.lins[-1].weight.shape[0]))
            layers.append((conv.eps, chain, fused))
        P["layers"] = layers
        P["keep"] = keep
        fc0 = self.MLP_layer.FC_layers[0]
        P["fc0"] = padw(fc0.weight, fc0.bias, fc0.weight.shape[0], up(fc0.weight.shape[1]))
        c["gin_padded"] = P
        return P

    def forward(self, g, h, p, e, snorm_n=None):
        _await_side(g)          # a sign_inv_net in overlap mode hands p over with an event kept on the graph
        ops.require_cuda(h)
        if p is None or self.pe_init != "lap_pe":
            raise NotImplementedError("HIP GINNet needs the positional encoding p (pe_init='lap_pe')")
        N = h.shape[0]
        hidx = h.long().reshape(N)
        p = p.contiguous().float()
        train = self.training
        if train and torch.is_grad_enabled() and any(q.requires_grad for q in self.parameters()):
            batch, ei, B = self._plan(g, N)
            plan = ops.build_plan(batch, ei, B, 0)
            y = self._forward_grad(plan, batch, ei, B, hidx, p)
        else:
            plan = cached_plan(g, N)       # the sign-invariant net's plan of this graph object, if there is one: ONE sn_batch_plan per batch
            with torch.no_grad():
                fz = self._fused_gin(g)
                if fz is not None:
                    # no host sync on this path: a bad atom type / an oversize graph is flagged in the plan's status block, the score is
                    # NaN, check_last() raises
                    self._last_plan = plan
                    self.g = g
                    return fz.run(plan, hidx.contiguous(), p), g
                if not train and self.embedding_h.weight.shape[1] % 4:
                    # eval, hidden width not a multiple of 4 (GIN_ZINC_LapPE_signinv_GIN.json: 95): every row would be misaligned and every
                    # Linear on the scalar kernel; run on zero-padded channels instead (`_gin_padded`: 95 -> 96, all rows 16-byte aligned)
                    P = self._gin_padded()
                    x = ops.embedding_sum(hidx, [P["emb_h"]])
                    x = ops.masked_linear(p, P["emb_p"], residual=x)
                    for eps, chain, fused in P["layers"]:
                        a = ops.gin_aggregate(x, plan, eps)
                        if fused is None:
                            x = _run_mlp(chain, a)
                            continue
                        ws, n_l, rp, d_out = fused                     # the layer's MLP in one launch (sn_mlp_chain_f32)
                        x = torch.empty(a.shape[0], d_out, dtype=torch.float32, device=a.device)
                        with ops._span("sn_mlp_chain_f32"):
                            check(lib().sn_mlp_chain_f32(ptr(a), a.shape[1], a.shape[0], a.shape[1], None, 0, ws, n_l, rp, ptr(x), d_out, d_out,
                                                         stream()), "sn_mlp_chain_f32")
                    hg = ops.segment_pool(x, plan, "mean" if self.readout != "sum" else "add")
                    fcs = self.MLP_layer.FC_layers
                    hg = ops.masked_linear(hg, P["fc0"], relu=len(fcs) > 1)
                    for i, fc in enumerate(fcs[1:], 1):
                        hg = ops.masked_linear(hg, self._pk(fc), relu=i < len(fcs) - 1)
                    self.g = g
                    return hg, g
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


GATED_MAX_LAYERS = 32


class _GatedLayerC(C.Structure):
    _fields_ = [("wabde", C.c_void_p), ("wc", C.c_void_p), ("h_scale", C.c_void_p), ("h_shift", C.c_void_p), ("residual", C.c_int),
                ("reserved", C.c_int)]


class _GatedParamsC(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("d", "d_out", "n_layers", "readout_mean", "ro_d1", "ro_d2")] + \
               [(n, C.c_void_p) for n in ("ro_w0", "ro_b0", "ro_w1", "ro_b1", "ro_w2", "ro_b2")] + \
               [("layers", _GatedLayerC * GATED_MAX_LAYERS)]


class _FusedGated:
    """Packed eval-mode parameters of a GatedGCNNet's layer stack + readout for sn_gatedgcn_fused_f32 (one launch).  `ok` False:
    shapes the stage kernel does not take (hidden not a multiple of 4 in [4, 96], a readout other than the 3-Linear MLPReadout,
    differing layer widths) — the layer path serves those."""

    def __init__(self, net):
        self.ok = False
        Ls = list(net.layers)
        d = Ls[0].in_channels
        fcs = list(net.MLP_layer.FC_layers)
        if not (4 <= d <= 96 and d % 4 == 0 and len(Ls) <= GATED_MAX_LAYERS and len(fcs) == 3 and fcs[2].weight.shape[0] == 1):
            return
        dp = max(48, 16 * ((d + 15) // 16))
        if any(L.in_channels != d for L in Ls) or any(L.out_channels != d for L in Ls[:-1]) or Ls[-1].out_channels > dp:
            return
        if fcs[0].weight.shape[0] > 128 or fcs[1].weight.shape[0] > 128 or fcs[0].weight.shape[1] != Ls[-1].out_channels:
            return
        keep = self._keep = []

        def hold(t):
            keep.append(t)
            return t.data_ptr()

        dev = Ls[0].A.weight.device
        P = _GatedParamsC()
        P.d, P.d_out, P.n_layers, P.readout_mean = d, Ls[-1].out_channels, len(Ls), 0 if net.readout == "sum" else 1
        P.ro_d1, P.ro_d2 = fcs[0].weight.shape[0], fcs[1].weight.shape[0]
        for i, fc in enumerate(fcs):
            setattr(P, f"ro_w{i}", hold(fc.weight.detach().float().contiguous()))
            setattr(P, f"ro_b{i}", hold(fc.bias.detach().float().contiguous()))
        rows_a, rows_c = 4 * dp, dp
        for l, L in enumerate(Ls):
            W = torch.zeros(rows_a, dp, dtype=torch.float32, device=dev)
            b = torch.zeros(rows_a, dtype=torch.float32, device=dev)
            for k, nme in enumerate("ABDE"):
                lin = getattr(L, nme)
                W[k * dp:k * dp + lin.weight.shape[0], :lin.weight.shape[1]].copy_(lin.weight.detach())
                b[k * dp:k * dp + lin.bias.shape[0]].copy_(lin.bias.detach())
            Wc = torch.zeros(rows_c, dp, dtype=torch.float32, device=dev)
            Wc[:L.C.weight.shape[0], :L.C.weight.shape[1]].copy_(L.C.weight.detach())
            hs, ht = ops.bn_fold(L.bn_node_h, dp)
            es, et = ops.bn_fold(L.bn_node_e, rows_c)
            Lp = P.layers[l]
            Lp.wabde = hold(ops.pack_split(W, b, None, None))
            Lp.wc = hold(ops.pack_split(Wc, ops.pad_vec(L.C.bias, rows_c), es, et))
            Lp.h_scale, Lp.h_shift = hold(hs), hold(ht)
            Lp.residual = 1 if L.residual else 0
        self.params, self.d = P, d
        self.max_edges = int(lib().sn_gatedgcn_max_edges(d))
        self.ok = True

    def run(self, plan, h0, e0):
        """h0 [N, d], e0 [E, d] (e0 is consumed: updated in place) -> scores [B, 1]"""
        y = torch.empty(plan.B, dtype=torch.float32, device=h0.device)
        with ops._span("sn_gatedgcn_fused_f32"):
            check(lib().sn_gatedgcn_fused_f32(C.byref(self.params), ptr(h0), ptr(e0), ptr(plan.graph_ptr), plan.B, ptr(plan.rowptr),
                                              ptr(plan.col), ptr(plan.eperm), ptr(plan.status), ptr(y), stream()), "sn_gatedgcn_fused_f32")
        return y.view(-1, 1)


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

    def _forward_behind_side(self, g, h, p, e, fused):
        """The sign-invariant net ran in overlap mode (its event is on the graph): the input encoders (embeddings, PE projection) are
        queued on ITS side stream, right behind it, and only the one-launch GatedGCN stack waits for that stream on the caller's —
        two balanced stages (plan + phi + rho + encoders | 16 layers + read-out) whose consecutive batches overlap."""
        from . import _lib as _lib_mod
        ev, side, tensors = g._sn_side
        N = h.shape[0]
        with torch.no_grad(), torch.cuda.stream(side), _lib_mod.stream_scope():
            plan = cached_plan(g, N)
            hidx, eidx = h.long().reshape(N), e.long().reshape(-1)
            st5 = plan.status[5:6]
            x = ops.embedding_sum(hidx, [self.embedding_h.weight], status=st5)
            pf = p.contiguous().float()
            if self.pe_aggregate == "concat":
                pp = ops.masked_linear(pf, self._pk(self.embedding_p))
                x = ops.masked_linear(torch.cat([x, pp], dim=1), self._pk(self.pe_proj))
            else:
                x = ops.masked_linear(pf, self._pk(self.embedding_p), residual=x)
            ee = ops.embedding_sum(eidx, [self.embedding_e.weight], status=st5)
            ev2 = torch.cuda.Event()
            ev2.record(side)
        for t in (h, e):
            t.record_stream(side)
        g._sn_side = (ev2, side, tuple(tensors) + (x, ee))
        _await_side(g, plan)
        self._last_plan = plan
        with torch.no_grad(), _lib_mod.stream_scope():
            return fused.run(plan, x, ee)

    def forward(self, g, h, p, e, snorm_n=None):
        if getattr(g, "_sn_side", None) is not None and p is not None and not self.training:
            fz = self._fused_gated(g)
            if fz is not None:
                ops.require_cuda(h)
                y = self._forward_behind_side(g, h, p, e, fz)
                self.g = g
                return y, g
        _await_side(g)          # a sign_inv_net in overlap mode hands p over with an event kept on the graph
        ops.require_cuda(h)
        if p is None:
            raise NotImplementedError("HIP GatedGCNNet needs the positional encoding p")
        N = h.shape[0]
        hidx, eidx = h.long().reshape(N), e.long().reshape(-1)
        p = p.contiguous().float()
        train = self.training
        if train and torch.is_grad_enabled() and any(q.requires_grad for q in self.parameters()):
            batch, ei, B = self._plan(g, N)
            y = self._forward_grad(ops.build_plan(batch, ei, B, 0), batch, ei, B, hidx, p, eidx)
        else:
            with torch.no_grad():
                plan = cached_plan(g, N)
                y = self._forward_value(plan, hidx, p, eidx, train, self._fused_gated(g))
        self.g = g
        return y, g

    fused_stages = True      # eval: layers + readout in ONE launch (sn_gatedgcn_fused_f32); False forces the layer path

    def _fused_gated(self, g):
        """The packed stage-kernel parameters if this batch can take the one-launch path, else None."""
        if self.training or not self.fused_stages:
            return None
        c = self.__dict__.setdefault("_cache", {})
        if "fused_gated" not in c:
            c["fused_gated"] = _FusedGated(self)
        fz = c["fused_gated"]
        if not fz.ok:
            return None
        # (a graph with more than 64 nodes, or more in-edges than the kernel's LDS image holds — sn_gatedgcn_max_edges(d), 176 at hidden
        #  68 —, takes the layer path, as the reference evaluates any graph: both counts come with a DGL batch (batch_num_nodes(),
        #  batch_num_edges()) and are read once per graph object.  A duck-typed graph without edge counts reaches the kernel, which
        #  flags such a graph on the device — NaN score, check_last())
        me = _max_in_edges(g)
        return fz if 0 < _max_nodes(g) <= 64 and (me is None or me <= fz.max_edges) else None

    def check_last(self):
        """Raise what the last one-launch eval forward flagged on the device (its scores are NaN in that case): a node / edge type
        outside the embedding tables (IndexError, as nn.Embedding) or a graph the stage kernel could not hold.  One host sync; also
        run by train().  The layer-at-a-time path (fused_stages = False, train mode) raises immediately instead."""
        plan, self._last_plan = getattr(self, "_last_plan", None), None
        if plan is not None:
            st = plan.status.tolist()
            if st[3] or st[5]:
                plan.status[3:6].zero_()         # (the plan is cached on the graph object: a later forward of the same graph starts clean)
            if st[0]:
                raise ValueError("GatedGCNNet: the last batch is malformed (batch_num_nodes / edges do not describe a batched graph; "
                                 f"sn_batch_plan status {st[0]})")
            if st[5]:
                raise IndexError(ops.EMBEDDING_INDEX_ERROR)
            if st[3] & 3:
                raise RuntimeError("GatedGCNNet: a graph of the last batch has more than 64 nodes or more in-edges than the one-launch "
                                   "kernel holds; its score is NaN — set fused_stages = False for such batches")

    def train(self, mode=True):
        self.check_last()
        return super().train(mode)

    def _forward_value(self, plan, hidx, p, eidx, train, fused=None):
        if fused is not None:
            # no host sync on this path: out-of-range ids are flagged in the plan's status block, the stage kernel then returns NaN
            # scores, check_last() raises
            st5 = plan.status[5:6]
            x = ops.embedding_sum(hidx, [self.embedding_h.weight], status=st5)
            if self.pe_aggregate == "concat":
                pp = ops.masked_linear(p, self._pk(self.embedding_p))
                x = ops.masked_linear(torch.cat([x, pp], dim=1), self._pk(self.pe_proj))
            else:
                x = ops.masked_linear(p, self._pk(self.embedding_p), residual=x)
            e = ops.embedding_sum(eidx, [self.embedding_e.weight], status=st5)
            self._last_plan = plan
            return fused.run(plan, x, e)
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


# ---------------------------------------------------------------------------------------------------------------
# PNA: layers/pna_layer.py:16-160, layers/pna_utils.py, nets/ZINC_graph_regression/pna_net.py:19-170
class FCLayer(nn.Module):
    """layers/pna_utils.py:170-243 (parameters only): `linear` (+ the activation name; no dropout, no b_norm in the shipped nets)."""

    def __init__(self, in_size, out_size, activation="relu"):
        super().__init__()
        self.in_size, self.out_size, self.activation = in_size, out_size, activation
        self.linear = nn.Linear(in_size, out_size, bias=True)
        nn.init.xavier_uniform_(self.linear.weight, 1 / in_size)       # FCLayer.reset_parameters (:223-228)
        self.linear.bias.data.zero_()


class _PnaMLP(nn.Module):
    """layers/pna_utils.py:246-279 with layers = 1 (pretrans_layers = posttrans_layers = 1 in every shipped PNA config)."""

    def __init__(self, in_size, hidden_size, out_size, layers, mid_activation="relu", last_activation="none"):
        super().__init__()
        if layers != 1:
            raise NotImplementedError("HIP PNA: pretrans_layers = posttrans_layers = 1 (the shipped configs)")
        self.fully_connected = nn.ModuleList([FCLayer(in_size, out_size, activation=last_activation)])


class PNATower(nn.Module):
    def __init__(self, in_dim, out_dim, edge_dim):
        super().__init__()
        self.batchnorm_h = nn.BatchNorm1d(out_dim)
        self.pretrans_h = _PnaMLP(2 * in_dim + edge_dim, in_dim, in_dim, 1)
        self.posttrans_h = _PnaMLP(13 * in_dim, out_dim, out_dim, 1)          # (4 aggregators x 3 scalers + 1) * in_dim


class PNALayer(nn.Module):
    def __init__(self, in_dim, out_dim, towers, edge_dim, residual):
        super().__init__()
        if in_dim % towers or out_dim % towers:
            raise ValueError("the number of towers has to divide in_dim and out_dim")
        self.in_dim, self.out_dim, self.n_towers = in_dim, out_dim, towers
        self.residual = residual and in_dim == out_dim
        self.towers = nn.ModuleList([PNATower(in_dim // towers, out_dim // towers, edge_dim) for _ in range(towers)])
        self.mixing_network_h = FCLayer(out_dim, out_dim, activation="LeakyReLU")


class PNANet(_PackCache, nn.Module):
    """nets/ZINC_graph_regression/pna_net.py:19-170 for the configuration PNA_ZINC_LapPE_signinv_GIN[_mask].json selects: pe_init
    'lap_pe', lap_lspe False, aggregators 'mean max min std', scalers 'identity amplification attenuation', towers with divided
    input, edge features, graph_norm + batch_norm, residual, no GRU, sum / mean readout.  Same constructor (`net_params`), forward
    contract `model(g, h, p, e, snorm_n) -> (scores, g)`, state_dict keys and attached `sign_inv_net`.  Per layer and tower:
    gather cat[h_src, h_dst, e] -> pretrans Linear -> sn_pna_aggregate_f32 -> posttrans Linear -> sn_pointwise_f32 (snorm_n and
    BatchNorm); then the mixing Linear + LeakyReLU + residual.  In eval mode the towers of a layer run side by side and the pretrans
    gather is folded into the aggregation, on activations padded to 16-byte rows (sn_pna_aggregate_gather_f32, `_forward_eval_padded`).  Eval, train-mode value (batch-statistic BatchNorm, running statistics
    updated) and — with gradients enabled — the differentiable path (`_forward_grad`)."""

    fused_layers = True      # eval, padded layout: one edge-term Linear for all layers, LeakyReLU + residual as the mixing Linear's epilogue


    def __init__(self, net_params):
        super().__init__()
        p = net_params
        hidden, out_dim = p["hidden_dim"], p["out_dim"]
        self.n_layers, self.readout = p["L"], p["readout"]
        self.pe_init, self.lap_method, self.lap_lspe = p["pe_init"], p["lap_method"], p["lap_lspe"]
        self.use_lapeig_loss, self.lambda_loss, self.alpha_loss = p["use_lapeig_loss"], p["lambda_loss"], p["alpha_loss"]
        self.pos_enc_dim, self.device = p["pos_enc_dim"], p["device"]
        self.graph_norm, self.batch_norm, self.residual = p["graph_norm"], p["batch_norm"], p["residual"]
        self.aggregators, self.scalers, self.avg_d, self.towers = p["aggregators"], p["scalers"], p["avg_d"], p["towers"]
        self.edge_feat = p["edge_feat"]
        if self.pe_init != "lap_pe" or self.lap_lspe or self.use_lapeig_loss:
            raise NotImplementedError("HIP PNANet covers pe_init='lap_pe' / lap_lspe=False (the sign-invariant PE configs)")
        if self.aggregators.split() != ["mean", "max", "min", "std"] or self.scalers.split() != ["identity", "amplification", "attenuation"]:
            raise NotImplementedError("HIP PNANet: aggregators 'mean max min std', scalers 'identity amplification attenuation'")
        if not (self.graph_norm and self.batch_norm and self.edge_feat) or p["gru"] or self.readout == "max":
            raise NotImplementedError("HIP PNANet: graph_norm, batch_norm, edge_feat True; gru False; readout sum / mean")
        if not (p["divide_input_first"] and p["divide_input_last"]):
            raise NotImplementedError("HIP PNANet: divide_input_first / _last True (the shipped configs)")
        if p.get("in_feat_dropout", 0.0) or p.get("dropout", 0.0):
            raise NotImplementedError("HIP PNANet: dropout 0.0 (as in the shipped configs)")
        edge_dim = p["edge_dim"]
        self.embedding_p = nn.Linear(self.pos_enc_dim, hidden)
        self.in_feat_dropout = nn.Dropout(0.0)
        self.embedding_h = nn.Embedding(p["num_atom_type"], hidden)
        self.embedding_e = nn.Embedding(p["num_bond_type"], edge_dim)
        self.layers = nn.ModuleList([PNALayer(hidden, hidden, self.towers, edge_dim, self.residual) for _ in range(self.n_layers - 1)] +
                                    [PNALayer(hidden, out_dim, self.towers, edge_dim, self.residual)])
        if p["pretrans_layers"] != 1 or p["posttrans_layers"] != 1:
            raise NotImplementedError("HIP PNANet: pretrans_layers = posttrans_layers = 1")
        self.MLP_layer = MLPReadout(out_dim, 1)
        self.g = None
        if self.lap_method == "sign_inv":
            self.sign_inv_net = get_sign_inv_net(net_params)

    _plan = GINNet._plan

    def forward(self, g, h, p, e, snorm_n):
        _await_side(g)          # a sign_inv_net in overlap mode hands p over with an event kept on the graph
        ops.require_cuda(h)
        if p is None or snorm_n is None:
            raise NotImplementedError("HIP PNANet needs the positional encoding p and snorm_n (graph_norm)")
        N = h.shape[0]
        train = self.training
        avg_log = float(self.avg_d["log"])
        sn = snorm_n.reshape(N).contiguous().float()
        grad = train and torch.is_grad_enabled() and any(q.requires_grad for q in self.parameters())
        if grad:
            batch, ei, B = self._plan(g, N)
            plan = ops.build_plan(batch, ei, B, 0)
            src, dst = ei[0], ei[1]
        else:
            plan = cached_plan(g, N)       # shared with the sign-invariant net: ONE sn_batch_plan per batch
            src, dst = (t.long() for t in g.edges())
        if grad:
            hg = self._forward_grad(plan, batch, ei, B, h.long().reshape(N), p.contiguous().float(), e.long().reshape(-1), sn, avg_log)
            self.g = g
            return hg, g
        with torch.no_grad():
            if not train and self._pna_padded()["ok"]:
                hg = self._forward_eval_padded(plan, h, p, e, sn, avg_log)
                self.g = g
                return hg, g
            x = ops.embedding_sum(h.long().reshape(N), [self.embedding_h.weight])
            x = ops.masked_linear(p.contiguous().float(), self._pk(self.embedding_p), residual=x)                 # h + embedding_p(p)  (:124-126)
            ef = ops.embedding_sum(e.long().reshape(-1), [self.embedding_e.weight])
            for L in self.layers:
                it = L.in_dim // L.n_towers
                outs = []
                for t, T in enumerate(L.towers):
                    ht = x[:, t * it:(t + 1) * it].contiguous()                                                       # copies / concats: plumbing
                    z = torch.cat([ht.index_select(0, src), ht.index_select(0, dst), ef], dim=1)                      # pretrans_edges (:38-44)
                    m = ops.masked_linear(z, self._pk(T.pretrans_h.fully_connected[0].linear))
                    a = ops.pna_aggregate(m, ht, plan, avg_log)                                                       # (:50-56, :69)
                    y = ops.masked_linear(a, self._pk(T.posttrans_h.fully_connected[0].linear))
                    site = self._bn(T.batchnorm_h, train)
                    if not train:
                        y = ops.pointwise(y, rowscale=sn, scale=site.scale, shift=site.shift)                          # * snorm_n, BatchNorm (:75-79)
                    else:
                        y = ops.pointwise(y, rowscale=sn)
                        sc, sh = site.affine(y, True)
                        y = ops.pointwise(y, scale=sc, shift=sh)
                    outs.append(y)
                hc = torch.cat(outs, dim=1)
                mix = ops.masked_linear(hc, self._pk(L.mixing_network_h.linear))
                x = ops.pointwise(mix, act="leaky", slope=0.01, residual=x if L.residual else None)                  # FCLayer LeakyReLU + residual
            hg = ops.segment_pool(x, plan, "mean" if self.readout != "sum" else "add")
            fcs = self.MLP_layer.FC_layers
            for i, fc in enumerate(fcs):
                hg = ops.masked_linear(hg, self._pk(fc), relu=i < len(fcs) - 1)
        self.g = g
        self._h_last = x
        return hg, g

    @staticmethod
    def _pad_map(Cc, nt, dev):
        """Tower t's channels [t*it, (t+1)*it) -> [t*it_p, t*it_p + it), it_p = it rounded up to a multiple of 4: every row of the padded
        activations is 16-byte aligned and every Linear takes the vector / LDS-staged kernels (hidden 70 = 5 towers x 14 would fall back to
        the scalar kernel: 73 % of the forward).  Pad channels are zero everywhere (zero weight rows / columns, BatchNorm scale 0)."""
        it = Cc // nt
        it_p = (it + 3) // 4 * 4
        c = torch.arange(Cc, device=dev)
        return (c // it) * it_p + c % it, nt * it_p, it, it_p

    def _pna_padded(self):
        """Eval cache: every parameter of the net re-laid-out for the padded, all-towers-side-by-side evaluation.  Per layer:
        sd [2 Cin_p, Cin_p] — rows t*it_p.. of the first / second half hold tower t's pretrans weights for h_src / h_dst (block diagonal:
        a tower reads its own input channels; pna_layer.py:38-44 is a Linear over cat[h_src, h_dst, e] = W_s h_src + W_d h_dst + W_e e + b);
        e [Cin_p, edge_dim] + the pretrans biases; post_w [towers, ot_p, 13 it_p] — the towers' posttrans Linears as one grouped (block-diagonal) Linear over
        the tower-major output of the aggregation kernel; the towers' folded BatchNorms side by side (its epilogue, with snorm_n); the mixing Linear in the padded channel order."""
        c = self.__dict__.setdefault("_cache", {})
        if "pna_padded" in c:
            return c["pna_padded"]
        dev = self.embedding_h.weight.device
        mk = lambda W_, b_: ops.PackedLinear(ops.pack_weight(W_.contiguous()), W_.shape[0], W_.shape[1], None if b_ is None else b_.contiguous())
        z = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=dev)
        nt = self.layers[0].n_towers
        hid = self.embedding_h.weight.shape[1]
        pos0, C0, _, _ = self._pad_map(hid, nt, dev)
        P = {"pos_in": pos0, "C_in": C0}
        E = z(self.embedding_h.weight.shape[0], C0)
        E[:, pos0] = self.embedding_h.weight.detach().float()
        Wp_, bp_ = z(C0, self.embedding_p.weight.shape[1]), z(C0)
        Wp_[pos0], bp_[pos0] = self.embedding_p.weight.detach().float(), self.embedding_p.bias.detach().float()
        P["emb_h"], P["emb_p"] = E, mk(Wp_, bp_)
        layers = []
        for L in self.layers:
            Cin, Cout = L.in_dim, L.out_dim
            pin, Cin_p, it, it_p = self._pad_map(Cin, nt, dev)
            pout, Cout_p, ot, ot_p = self._pad_map(Cout, nt, dev)
            pre = [T.pretrans_h.fully_connected[0].linear for T in L.towers]
            post = [T.posttrans_h.fully_connected[0].linear for T in L.towers]
            ed = pre[0].weight.shape[1] - 2 * it
            Wsd, We, be = z(2 * Cin_p, Cin_p), z(Cin_p, ed), z(Cin_p)
            Wpo, bpo = z(nt, ot_p, 13 * it_p), z(Cout_p)            # grouped: tower t reads its own 13 it_p tower-major columns
            sc, sh = z(Cout_p), z(Cout_p)
            for t in range(nt):
                W = pre[t].weight.detach().float()
                r = slice(t * it_p, t * it_p + it)
                Wsd[r, r] = W[:, :it]
                Wsd[Cin_p + t * it_p:Cin_p + t * it_p + it, r] = W[:, it:2 * it]
                We[r] = W[:, 2 * it:]
                be[r] = pre[t].bias.detach().float()
                Q = post[t].weight.detach().float()                   # [ot, 13 it]: cat[h_t, (4 s + a) blocks of it]
                ro = slice(t * ot_p, t * ot_p + ot)
                for j in range(13):
                    Wpo[t, :ot, j * it_p:j * it_p + it] = Q[:, j * it:(j + 1) * it]
                bpo[ro] = post[t].bias.detach().float()
                site = self._bn(L.towers[t].batchnorm_h, False)
                sc[ro], sh[ro] = site.scale, site.shift
            ml = L.mixing_network_h.linear
            Wm, bm = z(Cout_p, Cout_p), z(Cout_p)
            Wm[pout[:, None], pout[None, :]] = ml.weight.detach().float()
            bm[pout] = ml.bias.detach().float()
            layers.append({"sd": mk(Wsd, None), "e": mk(We, be), "e_raw": (We, be), "post_w": Wpo.contiguous(), "post_b": bpo, "nt": nt, "it_p": it_p,
                           "grouped": ot_p <= 16 and 13 * it_p <= 256, "scale": sc, "shift": sh, "mix": mk(Wm, bm),
                           "residual": L.residual, "pos_out": pout, "C_out": Cout_p})
        P["layers"] = layers
        P["ok"] = all(F["grouped"] for F in layers)       # (wider towers: the per-tower layer path below serves the net)
        # the edge features do not change from layer to layer (pna_layer.py updates h only): every layer's edge term W_e e + b as ONE
        # [E, L*C] Linear in front of the layer loop, read in place by the aggregation (column block l, row stride L*C)
        P["e_all"] = None
        if len({tuple(F["e_raw"][0].shape) for F in layers}) == 1:
            P["e_all"] = mk(torch.cat([F["e_raw"][0] for F in layers], 0), torch.cat([F["e_raw"][1] for F in layers], 0))
        fc0 = self.MLP_layer.FC_layers[0]
        W0 = z(fc0.weight.shape[0], layers[-1]["C_out"])
        W0[:, layers[-1]["pos_out"]] = fc0.weight.detach().float()
        P["fc0"] = mk(W0, fc0.bias.detach().float())
        c["pna_padded"] = P
        return P

    def _forward_eval_padded(self, plan, h, p, e, sn, avg_log):
        """Eval forward on the padded layout (`_pna_padded`): all towers of a layer side by side, the pretrans gather folded into the
        aggregation (sn_pna_aggregate_gather_f32): 7 launches per layer instead of 9 per tower, every Linear on the vector kernels."""
        P = self._pna_padded()
        N = h.shape[0]
        x = ops.embedding_sum(h.long().reshape(N), [P["emb_h"]])
        x = ops.masked_linear(p.contiguous().float(), P["emb_p"], residual=x)                                    # h + embedding_p(p)  (:124-126)
        ef = ops.embedding_sum(e.long().reshape(-1), [self.embedding_e.weight])
        fuse = self.fused_layers
        qe_all = ops.masked_linear(ef, P["e_all"]) if (fuse and P["e_all"] is not None) else None                # [E, L*C]
        for li, F in enumerate(P["layers"]):
            psd = ops.masked_linear(x, F["sd"])                                                                  # [N, 2C] = [W_s h | W_d h]
            if qe_all is not None:
                a = ops.pna_aggregate_gather(psd, qe_all, x, plan, avg_log, tower_width=F["it_p"], qe_layer=li)
            else:
                qe = ops.masked_linear(ef, F["e"])                                                               # [E, C]  = W_e e + b
                a = ops.pna_aggregate_gather(psd, qe, x, plan, avg_log, tower_width=F["it_p"])                   # (:50-56, :69), tower-major
            hc = ops.grouped_linear(a, F["post_w"], F["post_b"], F["nt"], rowscale=sn, scale=F["scale"], shift=F["shift"])   # posttrans,
            #                                                                                            * snorm_n, BatchNorm (:69-79)
            if fuse:      # FCLayer's LeakyReLU and the residual as the mixing Linear's epilogue
                x = ops.masked_linear(hc, F["mix"], leaky=True, residual=x if F["residual"] else None)
            else:
                mix = ops.masked_linear(hc, F["mix"])
                x = ops.pointwise(mix, act="leaky", slope=0.01, residual=x if F["residual"] else None)           # FCLayer LeakyReLU + residual
        self._h_last = x.index_select(1, P["layers"][-1]["pos_out"])
        hg = ops.segment_pool(x, plan, "mean" if self.readout != "sum" else "add")
        fcs = self.MLP_layer.FC_layers
        hg = ops.masked_linear(hg, P["fc0"], relu=len(fcs) > 1)
        for i, fc in enumerate(fcs[1:], 1):
            hg = ops.masked_linear(hg, self._pk(fc), relu=i < len(fcs) - 1)
        return hg

    def _forward_grad(self, plan, batch, ei, B, hidx, p, eidx, sn, avg_log):
        """Differentiable train-mode forward (SURVEY.md §8 f1 for this net): the same ops as autograd nodes with hand-written adjoints
        (csrc/dgl_layers.hip: CSR walks, no atomics); towers' column slices and concatenations are torch views / copies."""
        from . import autograd as AG
        rplan = ops.build_plan(batch, ei.flip(0).contiguous(), B, 0)           # edges grouped by SOURCE: adjoint of h[src]
        src, dst = ei[0], ei[1]
        x = AG.masked_add(AG.linear(p, self.embedding_p.weight, self.embedding_p.bias), AG.embedding_sum(hidx, [self.embedding_h.weight]))
        ef = AG.embedding_sum(eidx, [self.embedding_e.weight])
        for L in self.layers:
            it = L.in_dim // L.n_towers
            outs = []
            for t, T in enumerate(L.towers):
                ht = x[:, t * it:(t + 1) * it].contiguous()
                z = torch.cat([AG.gather_rows(ht, src, rplan), AG.gather_rows(ht, dst, plan), ef], dim=1)
                pre, post = T.pretrans_h.fully_connected[0].linear, T.posttrans_h.fully_connected[0].linear
                m = AG.linear(z, pre.weight, pre.bias)
                a = AG.pna_aggregate(m, ht, plan, avg_log)
                y = AG.act_residual(AG.linear(a, post.weight, post.bias), rowscale=sn)             # graph_norm: h * snorm_n
                outs.append(AG.bn_act(y, T.batchnorm_h, relu=False))
            mixl = L.mixing_network_h.linear
            mix = AG.linear(torch.cat(outs, dim=1), mixl.weight, mixl.bias)
            x = AG.act_residual(mix, residual=x if L.residual else None, act="leaky", slope=0.01)
        hg = AG.segment_pool(x, plan, "mean" if self.readout != "sum" else "add")
        fcs = self.MLP_layer.FC_layers
        for i, fc in enumerate(fcs):
            hg = AG.linear(hg, fc.weight, fc.bias, relu=i < len(fcs) - 1)
        self._h_last = x.detach()
        return hg

    def loss(self, scores, targets):
        return (scores - targets).abs().mean()


# ---------------------------------------------------------------------------------------------------------------
# Sparse graph Transformer: layers/transformer.py:112-317, nets/ZINC_graph_regression/transformer_net.py:21-150
class MultiHeadAttentionLayer(nn.Module):
    def __init__(self, gamma, in_dim, out_dim, num_heads):
        super().__init__()
        self.out_dim, self.num_heads, self.gamma = out_dim, num_heads, gamma        # gamma: the owner's Parameter, registered here too
        self.Q = nn.Linear(in_dim, out_dim * num_heads, bias=False)
        self.K = nn.Linear(in_dim, out_dim * num_heads, bias=False)
        self.E = nn.Linear(in_dim, out_dim * num_heads, bias=False)
        self.V = nn.Linear(in_dim, out_dim * num_heads, bias=False)


class BatchedTransformerLayer(nn.Module):
    """layers/transformer.py:234-317 with the arguments transformer_net.py:69-70 passes (the rest at their defaults: residual,
    batch_norm, no layer_norm, use_bias False, dropout 0)."""

    def __init__(self, in_dim, out_dim, num_heads, full_graph, use_edge=True):
        super().__init__()
        if full_graph or not use_edge:
            raise NotImplementedError("HIP graph Transformer: full_graph False with edge features (the shipped sign_inv configs)")
        if out_dim % num_heads or out_dim // num_heads > 32:
            raise ValueError("num_heads must divide out_dim and the head width must be <= 32")
        self.in_channels, self.out_channels, self.num_heads = in_dim, out_dim, num_heads
        self.gamma = nn.Parameter(torch.FloatTensor([0.1]))          # unused when full_graph is False, but part of the state_dict
        self.attention_h = MultiHeadAttentionLayer(self.gamma, in_dim, out_dim // num_heads, num_heads)
        self.O_h = nn.Linear(out_dim, out_dim)
        self.batch_norm1_h = nn.BatchNorm1d(out_dim)
        self.FFN_h_layer1 = nn.Linear(out_dim, out_dim * 2)
        self.FFN_h_layer2 = nn.Linear(out_dim * 2, out_dim)
        self.batch_norm2_h = nn.BatchNorm1d(out_dim)


class _FusedTransformer:
    """Packed eval-mode parameters of a TransformerNet for sn_transformer_net_fused_f32: embeddings, every layer (projections, edge
    attention, O_h, both BatchNorms, the FFN), readout and MLPReadout in ONE launch — a workgroup per graph on the stage kernel of
    csrc/fused_gnn.hip (Transformer mode).  The layers' E projections of the edge embedding stay one Linear in front of it.  `ok`
    False: anything but the shipped shape (hidden = out = 64, 8 heads, BatchNorm, residual, 3-Linear readout; pe_aggregate 'add' or 'concat')."""

    def __init__(self, net):
        from .fused import _GnnParams, GNN_MAX_LAYERS
        from .dgl_deepsigns import _pad_mat
        self.ok = False
        Ls, fcs = list(net.layers), list(net.MLP_layer.FC_layers)
        d, kp = net.embedding_h.weight.shape[1], net.embedding_p.weight.shape[1]
        if not (d == 64 and net.batch_norm and net.residual and kp <= 64):        # (the net's layer_norm flag is not handed to the layers)
            return
        if len(Ls) > GNN_MAX_LAYERS or len(fcs) != 3 or fcs[2].weight.shape[0] != 1:
            return
        if any(L.in_channels != 64 or L.out_channels != 64 or L.num_heads != 8 for L in Ls):
            return
        dev = net.embedding_h.weight.device
        keep = self._keep = []

        def hold(t):
            keep.append(t)
            return t.data_ptr()

        ones = torch.ones(64, dtype=torch.float32, device=dev)
        w = lambda t: t.detach().float().contiguous()
        P = _GnnParams()
        P.d, P.n_layers, P.n_out = 64, len(Ls), 1
        P.node_discrete, P.node_nf, P.edge_discrete, P.edge_nf = 1, 1, 0, 0
        P.node_vocab, P.edge_vocab = net.embedding_h.weight.shape[0], 0
        P.ntab[0] = hold(w(net.embedding_h.weight))
        P.rho_out_w = None
        if net.pe_aggregate == "concat":
            # h = pe_proj(cat[embedding_h, embedding_p(p)]) (transformer_net.py:96-99) = W[:, :d] emb_h + (W[:, d:] W_p) p + (W[:, d:] b_p + b):
            # the two affine maps behind p folded once, in float64 (as the GINE net folds rho.out into its input Linear)
            Wc, bc = net.pe_proj.weight.detach().double(), net.pe_proj.bias.detach().double()
            Wp_, bp_ = net.embedding_p.weight.detach().double(), net.embedding_p.bias.detach().double()
            P.lin_a = hold(ops.pack_split(Wc[:, :64].float().contiguous()))
            P.lin_b = hold(ops.pack_split(_pad_mat((Wc[:, 64:] @ Wp_).float(), 64), (Wc[:, 64:] @ bp_ + bc).float().contiguous()))
        else:
            P.lin_a = hold(ops.pack_split(torch.eye(64, dtype=torch.float32, device=dev)))
            P.lin_b = hold(ops.pack_split(_pad_mat(net.embedding_p.weight, 64), ops.pad_vec(net.embedding_p.bias, 64)))
        for l, L in enumerate(Ls):
            A = L.attention_h
            s1, s2 = net._bn(L.batch_norm1_h, False), net._bn(L.batch_norm2_h, False)
            W1, b1, W2 = w(L.FFN_h_layer1.weight), w(L.FFN_h_layer1.bias), w(L.FFN_h_layer2.weight)
            mats = [ops.pack_split(w(A.Q.weight)), ops.pack_split(w(A.K.weight)), ops.pack_split(w(A.V.weight)),
                    ops.pack_split(w(L.O_h.weight), w(L.O_h.bias), s1.scale, s1.shift),
                    ops.pack_split(W1[:64].contiguous(), b1[:64].contiguous()), ops.pack_split(W1[64:].contiguous(), b1[64:].contiguous()),
                    ops.pack_split(W2[:, :64].contiguous()),
                    ops.pack_split(W2[:, 64:].contiguous(), w(L.FFN_h_layer2.bias), s2.scale, s2.shift)]
            Lp = P.layers[l]
            for j, m in enumerate(mats):
                Lp.etab[j] = hold(m)
            Lp.w1s, Lp.w2s = Lp.etab[0], Lp.etab[1]
        P.head_w1 = hold(ops.pack_split(_pad_mat(fcs[0].weight, 64), ones, ops.pad_vec(fcs[0].bias, 64)))
        self.head_mid = hold(ops.pack_split(_pad_mat(fcs[1].weight, 64), ones, ops.pad_vec(fcs[1].bias, 64)))
        P.head_w2 = hold(ops.pack_split(_pad_mat(fcs[2].weight, 64), ops.pad_vec(fcs[2].bias, 64)))
        self.params, self.kp, self.pool_mean = P, kp, 0 if net.readout == "sum" else 1
        self.ok = True

    def run(self, plan, hidx, p, e_proj):
        """atom types [N] int64, positional encoding [N, k], the layers' E projections [E, L*64] -> scores [B, 1]"""
        y = torch.empty(plan.B, 1, dtype=torch.float32, device=p.device)
        with ops._span("sn_transformer_net_fused_f32"):
            check(lib().sn_transformer_net_fused_f32(C.byref(self.params), self.head_mid, self.pool_mean, ptr(hidx), ptr(p), p.shape[1], self.kp,
                                                     ptr(e_proj), e_proj.shape[1], ptr(plan.graph_ptr), plan.B, ptr(plan.rowptr), ptr(plan.col),
                                                     ptr(plan.eperm), ptr(plan.status), ptr(y), ptr(plan.status), 8, stream()),
                  "sn_transformer_net_fused_f32")
        return y


class TransformerNet(_PackCache, nn.Module):
    """nets/ZINC_graph_regression/transformer_net.py:21-150 for pe_init 'lap_pe', lap_lspe False, edge_feat True, full_graph False
    (Transformer_ZINC_LapPE_signinv_GIN[_masked].json): embedding_h / embedding_p with `add` or `concat` + pe_proj, edge
    embedding, L x [Q/K/V/E projections -> sn_edge_attention_f32 -> O_h + residual -> BatchNorm -> FFN + residual -> BatchNorm],
    sum / mean readout, MLPReadout.  Same constructor, forward contract, state_dict keys and attached `sign_inv_net`.
    Eval, train-mode value and — with gradients enabled — the differentiable path (`_forward_grad`)."""

    def __init__(self, net_params):
        super().__init__()
        p = net_params
        hidden, out_dim = p["hidden_dim"], p["out_dim"]
        self.n_layers, self.readout, self.batch_norm, self.layer_norm = p["L"], p["readout"], p["batch_norm"], p["layer_norm"]
        self.residual, self.edge_feat, self.device = p["residual"], p["edge_feat"], p["device"]
        self.pe_init, self.lap_method, self.lap_lspe = p["pe_init"], p["lap_method"], p["lap_lspe"]
        self.use_lapeig_loss, self.lambda_loss, self.alpha_loss = p["use_lapeig_loss"], p["lambda_loss"], p["alpha_loss"]
        self.pos_enc_dim, self.pe_aggregate = p["pos_enc_dim"], p["pe_aggregate"]
        if self.pe_init != "lap_pe" or self.lap_lspe or self.use_lapeig_loss:
            raise NotImplementedError("HIP TransformerNet covers pe_init='lap_pe' / lap_lspe=False (the sign-invariant PE configs)")
        if self.readout == "max" or not self.edge_feat:
            raise NotImplementedError("HIP TransformerNet: readout sum / mean, edge_feat True")
        if p.get("in_feat_dropout", 0.0) or p.get("dropout", 0.0):
            raise NotImplementedError("HIP TransformerNet: dropout 0.0 (as in the shipped configs)")
        self.embedding_p = nn.Linear(self.pos_enc_dim, hidden)
        self.embedding_h = nn.Embedding(p["num_atom_type"], hidden)
        self.embedding_e = nn.Embedding(p["num_bond_type"], hidden)
        self.in_feat_dropout = nn.Dropout(0.0)
        self.layers = nn.ModuleList([BatchedTransformerLayer(hidden, hidden, p["n_heads"], p["full_graph"], use_edge=True)
                                     for _ in range(self.n_layers - 1)] +
                                    [BatchedTransformerLayer(hidden, out_dim, p["n_heads"], p["full_graph"], use_edge=True)])
        self.MLP_layer = MLPReadout(out_dim, 1)
        self.g = None
        if self.lap_method == "sign_inv":
            self.sign_inv_net = get_sign_inv_net(net_params)
        if self.pe_aggregate == "concat":
            self.pe_proj = nn.Linear(2 * hidden, hidden)

    _plan = GINNet._plan

    fused_layers = True      # eval: fused projections / epilogues (False: one launch per op, as the train-mode value path)
    fused_stages = True      # eval, the shipped shape: everything behind the E projection in ONE launch (sn_transformer_net_fused_f32)

    def _fused_tf(self, g):
        """The packed stage-kernel parameters if this batch can take the one-launch path, else None (see GINNet._fused_gin)."""
        if self.training or not self.fused_stages or not self._fusable():
            return None
        c = self.__dict__.setdefault("_cache", {})
        if "fused_tf" not in c:
            c["fused_tf"] = _FusedTransformer(self)
        fz = c["fused_tf"]
        if not fz.ok:
            return None
        me = _max_in_edges(g)
        return fz if 0 < _max_nodes(g) <= 64 and (me is None or me <= 192) else None

    check_last = GINNet.check_last

    def train(self, mode=True):
        self.check_last()
        return super().train(mode)

    def _fusable(self):
        """Every layer maps hidden -> hidden (the shipped configs: out_dim == hidden_dim), so that the layers' E projections stack."""
        d = self.layers[0].out_channels
        return self.fused_layers and all(L.in_channels == d and L.out_channels == d for L in self.layers)

    def _fused_eval(self):
        """Eval cache: cat[W_Q; W_K; W_V] per layer and cat over the layers of W_E, packed (no biases: use_bias False)."""
        c = self.__dict__.setdefault("_cache", {})
        if "fused" not in c:
            def pack(ws):
                W = torch.cat([w.detach().float() for w in ws], 0).contiguous()
                return ops.PackedLinear(ops.pack_weight(W), W.shape[0], W.shape[1], None)
            c["fused"] = {"qkv": [pack([L.attention_h.Q.weight, L.attention_h.K.weight, L.attention_h.V.weight]) for L in self.layers],
                          "E": pack([L.attention_h.E.weight for L in self.layers])}
        return c["fused"]

    def _bn_res(self, y, res, bn, train):
        """BatchNorm1d(res + y): eval folded into one pointwise pass; train: batch statistics of the sum."""
        site = self._bn(bn, train)
        if not train:
            s = ops.pointwise(y, residual=res)
            return ops.pointwise(s, scale=site.scale, shift=site.shift)
        s = ops.pointwise(y, residual=res)
        sc, sh = site.affine(s, True)
        return ops.pointwise(s, scale=sc, shift=sh)

    def forward(self, g, h, p, e, snorm_n=None):
        _await_side(g)          # a sign_inv_net in overlap mode hands p over with an event kept on the graph
        ops.require_cuda(h)
        if p is None:
            raise NotImplementedError("HIP TransformerNet needs the positional encoding p")
        N = h.shape[0]
        train = self.training
        if train and torch.is_grad_enabled() and any(q.requires_grad for q in self.parameters()):
            batch, ei, B = self._plan(g, N)
            plan = ops.build_plan(batch, ei, B, 0)
            hg = self._forward_grad(plan, batch, ei, B, h.long().reshape(N), p.contiguous().float(), e.long().reshape(-1))
            self.g = g
            return hg, g
        plan = cached_plan(g, N)           # shared with the sign-invariant net: ONE sn_batch_plan per batch
        with torch.no_grad():
            fzs = self._fused_tf(g)
            if fzs is not None:
                # no host sync on this path: a bad atom type / an oversize graph -> NaN score + check_last(); a bond type outside its table
                # is flagged in the plan's status block by the embedding (status[5]) and raised by check_last() as well
                ef = ops.embedding_sum(e.long().reshape(-1), [self.embedding_e.weight], status=plan.status[5:6])
                Ee_all = ops.masked_linear(ef, self._fused_eval()["E"])
                self._last_plan = plan
                self.g = g
                return fzs.run(plan, h.long().reshape(N).contiguous(), p.contiguous().float(), Ee_all), g
            x = ops.embedding_sum(h.long().reshape(N), [self.embedding_h.weight])
            pp = p.contiguous().float()
            if self.pe_aggregate == "concat":
                pe = ops.masked_linear(pp, self._pk(self.embedding_p))
                x = ops.masked_linear(torch.cat([x, pe], dim=1), self._pk(self.pe_proj))                              # (:96-99)
            else:
                x = ops.masked_linear(pp, self._pk(self.embedding_p), residual=x)                                     # (:101-102)
            ef = ops.embedding_sum(e.long().reshape(-1), [self.embedding_e.weight])
            if not train and self._fusable():
                # eval (round 4): 5 launches per layer instead of 12.  Q | K | V are ONE [3d, d] Linear whose column blocks the attention
                # reads in place; every layer's E projection of the (layer-independent) edge embedding is one [L*d, d] Linear up front;
                # `BatchNorm(x + Linear(h))` is the Linear's epilogue (residual in front of the folded affine: the same operations in the
                # same order as the add + affine passes they replace — bit-identical).
                fz = self._fused_eval()
                Ee_all = ops.masked_linear(ef, fz["E"])
                for li, L in enumerate(self.layers):
                    qkv = ops.masked_linear(x, fz["qkv"][li])
                    a = ops.edge_attention_fused(qkv, Ee_all, li, plan, L.num_heads)                                  # (:150-228)
                    s1, s2 = self._bn(L.batch_norm1_h, False), self._bn(L.batch_norm2_h, False)
                    x1 = ops.masked_linear(a, self._pk(L.O_h), residual=x, residual_pre=True, scale=s1.scale, shift=s1.shift)       # (:283-290)
                    f = ops.masked_linear(x1, self._pk(L.FFN_h_layer1), relu=True)
                    x = ops.masked_linear(f, self._pk(L.FFN_h_layer2), residual=x1, residual_pre=True, scale=s2.scale, shift=s2.shift)  # (:300-308)
            else:
              for L in self.layers:
                A = L.attention_h
                Q, K, V = (ops.masked_linear(x, self._pk(getattr(A, n))) for n in "QKV")
                Ee = ops.masked_linear(ef, self._pk(A.E))
                a = ops.edge_attention(Q, K, V, Ee, plan, L.num_heads)                                                # (:150-228)
                o = ops.masked_linear(a, self._pk(L.O_h))
                x1 = self._bn_res(o, x, L.batch_norm1_h, train)                                                       # residual, BatchNorm (:283-290)
                f = ops.masked_linear(ops.masked_linear(x1, self._pk(L.FFN_h_layer1), relu=True), self._pk(L.FFN_h_layer2))
                x = self._bn_res(f, x1, L.batch_norm2_h, train)                                                       # (:300-308)
            hg = ops.segment_pool(x, plan, "mean" if self.readout != "sum" else "add")
            fcs = self.MLP_layer.FC_layers
            for i, fc in enumerate(fcs):
                hg = ops.masked_linear(hg, self._pk(fc), relu=i < len(fcs) - 1)
        self.g = g
        self._h_last = x
        return hg, g

    def _forward_grad(self, plan, batch, ei, B, hidx, p, eidx):
        """Differentiable train-mode forward: the sparse attention's adjoint is sn_edge_attention_bwd_f32 (destination pass + source
        pass over the reverse CSR, no atomics); BatchNorm with batch statistics, residuals, FFN as in the value path."""
        from . import autograd as AG
        rplan = ops.build_plan(batch, ei.flip(0).contiguous(), B, 0)
        x = AG.embedding_sum(hidx, [self.embedding_h.weight])
        pe = AG.linear(p, self.embedding_p.weight, self.embedding_p.bias)
        if self.pe_aggregate == "concat":
            x = AG.linear(torch.cat([x, pe], dim=1), self.pe_proj.weight, self.pe_proj.bias)
        else:
            x = AG.masked_add(pe, x)
        ef = AG.embedding_sum(eidx, [self.embedding_e.weight])
        for L in self.layers:
            A = L.attention_h
            Q, K, V = (AG.linear(x, getattr(A, n).weight, None) for n in "QKV")
            Ee = AG.linear(ef, A.E.weight, None)
            a = AG.edge_attention(Q, K, V, Ee, plan, rplan, L.num_heads)
            o = AG.linear(a, L.O_h.weight, L.O_h.bias)
            x1 = AG.bn_act(AG.masked_add(o, x), L.batch_norm1_h, relu=False)
            f = AG.linear(AG.linear(x1, L.FFN_h_layer1.weight, L.FFN_h_layer1.bias, relu=True), L.FFN_h_layer2.weight, L.FFN_h_layer2.bias)
            x = AG.bn_act(AG.masked_add(f, x1), L.batch_norm2_h, relu=False)
        hg = AG.segment_pool(x, plan, "mean" if self.readout != "sum" else "add")
        fcs = self.MLP_layer.FC_layers
        for i, fc in enumerate(fcs):
            hg = AG.linear(hg, fc.weight, fc.bias, relu=i < len(fcs) - 1)
        self._h_last = x.detach()
        return hg

    def loss(self, scores, targets):
        return (scores - targets).abs().mean()


# ---------------------------------------------------------------------------------------------------------------
# GAT: nets/ZINC_graph_regression/gat_net.py:19-148 on dgl.nn.pytorch.GATConv
class GATConv(nn.Module):
    """Parameter container with dgl.nn.pytorch.GATConv's names and initialisation (fc without bias, attn_l / attn_r [1, heads, out],
    bias [heads*out]; xavier_normal with the ReLU gain, zero bias); the arithmetic is sn_masked_linear_f32 + sn_gat_aggregate_f32."""

    def __init__(self, in_feats, out_feats, num_heads, negative_slope=0.2):
        super().__init__()
        if out_feats > 64:
            raise ValueError("HIP GATConv: head width <= 64")
        self.in_feats, self.out_feats, self.num_heads, self.negative_slope = in_feats, out_feats, num_heads, negative_slope
        self.fc = nn.Linear(in_feats, out_feats * num_heads, bias=False)
        self.attn_l = nn.Parameter(torch.empty(1, num_heads, out_feats))
        self.attn_r = nn.Parameter(torch.empty(1, num_heads, out_feats))
        self.bias = nn.Parameter(torch.zeros(num_heads * out_feats))
        gain = nn.init.calculate_gain("relu")
        nn.init.xavier_normal_(self.fc.weight, gain=gain)
        nn.init.xavier_normal_(self.attn_l, gain=gain)
        nn.init.xavier_normal_(self.attn_r, gain=gain)


class GATNet(_PackCache, nn.Module):
    """nets/ZINC_graph_regression/gat_net.py:19-148 for pe_init = 'lap_pe', lap_lspe = False (GAT_ZINC_LapPE_signinv_GIN.json):
    h = embedding_h(h) + embedding_p(p), L GATConv(ReLU) layers — the first L-1 with their heads flattened, the last averaged over the
    heads (:108-110) — mean / sum readout, MLPReadout.  'GAT (no edge feature)': the edge embedding is a parameter of the
    state_dict only.  Same constructor, forward contract `model(g, h, p, e, snorm_n) -> (scores, g)`, state_dict keys and attached
    `sign_inv_net`.  No BatchNorm / dropout in this net: eval and train mode compute the same value; with gradients enabled in train
    mode the differentiable path (`_forward_grad`)."""

    def __init__(self, net_params):
        super().__init__()
        p = net_params
        hidden, out_dim = p["hidden_dim"], p["out_dim"]
        self.n_layers, self.readout, self.n_heads = p["L"], p["readout"], p["n_heads"]
        self.pe_init, self.lap_method, self.lap_lspe = p["pe_init"], p["lap_method"], p["lap_lspe"]
        self.use_lapeig_loss, self.lambda_loss, self.alpha_loss = p["use_lapeig_loss"], p["lambda_loss"], p["alpha_loss"]
        self.pos_enc_dim, self.device, self.edge_feat = p["pos_enc_dim"], p["device"], p["edge_feat"]
        self.batch_norm, self.residual = p["batch_norm"], p["residual"]              # read and ignored by the reference too
        if self.pe_init != "lap_pe" or self.lap_lspe or self.use_lapeig_loss:
            raise NotImplementedError("HIP GATNet covers pe_init='lap_pe' / lap_lspe=False (the sign-invariant PE configs)")
        if self.readout == "max":
            raise NotImplementedError("HIP GATNet: readout sum / mean")
        if p.get("in_feat_dropout", 0.0) or p.get("dropout", 0.0):
            raise NotImplementedError("HIP GATNet: dropout 0.0 (as in the shipped configs)")
        if self.n_layers < 2:
            raise ValueError("GATNet: L >= 2 (gat_net.py:62-66 always builds a first and a last layer)")
        H = self.n_heads
        self.embedding_p = nn.Linear(self.pos_enc_dim, hidden)
        self.embedding_h = nn.Embedding(p["num_atom_type"], hidden)
        self.embedding_e = nn.Embedding(p["num_bond_type"], hidden) if self.edge_feat else nn.Linear(1, hidden)
        self.in_feat_dropout = nn.Dropout(0.0)
        self.layers = nn.ModuleList([GATConv(hidden, hidden, H)] + [GATConv(H * hidden, hidden, H) for _ in range(1, self.n_layers - 1)] +
                                    [GATConv(H * hidden, out_dim, H)])
        self.MLP_layer = MLPReadout(out_dim, 1)
        self.g = None
        if self.lap_method == "sign_inv":
            self.sign_inv_net = get_sign_inv_net(net_params)

    def forward(self, g, h, p, e, snorm_n=None):
        _await_side(g)          # a sign_inv_net in overlap mode hands p over with an event kept on the graph
        ops.require_cuda(h)
        if p is None:
            raise NotImplementedError("HIP GATNet needs the positional encoding p")
        N = h.shape[0]
        if self.training and torch.is_grad_enabled() and any(q.requires_grad for q in self.parameters()):
            batch, ei, B = self._plan(g, N)
            hg = self._forward_grad(ops.build_plan(batch, ei, B, 0), batch, ei, B, h.long().reshape(N), p.contiguous().float())
            self.g = g
            return hg, g
        plan = cached_plan(g, N)
        with torch.no_grad():
            zero_deg = (plan.rowptr[1:] == plan.rowptr[:-1]).any()
            x = ops.embedding_sum(h.long().reshape(N), [self.embedding_h.weight])     # raises IndexError as nn.Embedding (one host read)
            x = ops.masked_linear(p.contiguous().float(), self._pk(self.embedding_p), residual=x)                 # h + embedding_p(p)  (:97-99)
            H = self.n_heads
            for i, L in enumerate(self.layers):
                f = ops.masked_linear(x, self._fc(L))
                x = ops.gat_aggregate(f, L.attn_l.detach(), L.attn_r.detach(), L.bias.detach(), plan, H, L.negative_slope, relu=True)
            # the last layer's heads are averaged (:110): sum over the head axis, then 1/H
            x = ops.slot_sum(x.view(N * H, -1), N, H)
            x = ops.pointwise(x, scale=self._const(1.0 / H, x.shape[1]), shift=self._const(0.0, x.shape[1]))
            hg = ops.segment_pool(x, plan, "mean" if self.readout != "sum" else "add")
            fcs = self.MLP_layer.FC_layers
            for i, fc in enumerate(fcs):
                hg = ops.masked_linear(hg, self._pk(fc), relu=i < len(fcs) - 1)
        zero_msg = ("There are 0-in-degree nodes in the graph: GATConv's edge softmax is undefined for them (DGL raises "
                    "DGLError here unless allow_zero_in_degree is set, which gat_net.py:62-66 leaves at False)")
        if ops.deferring():              # a recorded step (serving.GraphedDGLForward) must not wait for the host: its check() reads these
            ops.defer("plan", plan)
            ops.defer("flag", zero_deg, zero_msg)
        else:
            plan.check()                 # malformed batch
            if bool(zero_deg):
                raise ValueError(zero_msg)
        self.g = g
        self._h_last = x
        return hg, g

    _plan = GINNet._plan

    def _forward_grad(self, plan, batch, ei, B, hidx, p):
        """Differentiable train-mode forward (SURVEY.md §8 f1 for this net): the same ops as autograd nodes; the GATConv adjoint is
        sn_gat_aggregate_bwd_f32 (CSR walks over the in-edges, then over the out-edges; no atomics)."""
        from . import autograd as AG
        plan.check()
        if bool((plan.rowptr[1:] == plan.rowptr[:-1]).any()):
            raise ValueError("There are 0-in-degree nodes in the graph: GATConv's edge softmax is undefined for them")
        rplan = ops.build_plan(batch, ei.flip(0).contiguous(), B, 0)           # edges grouped by SOURCE
        N, H = hidx.shape[0], self.n_heads
        x = AG.masked_add(AG.linear(p, self.embedding_p.weight, self.embedding_p.bias), AG.embedding_sum(hidx, [self.embedding_h.weight]))
        for L in self.layers:
            x = AG.gat_aggregate(AG.linear(x, L.fc.weight), L.attn_l, L.attn_r, L.bias, plan, rplan, H, L.negative_slope, True)
        x = AG.slot_sum(x.view(N * H, -1), N, H)                               # mean over the heads (:110)
        x = AG.act_residual(x, rowscale=torch.full((N,), 1.0 / H, dtype=torch.float32, device=x.device))
        hg = AG.segment_pool(x, plan, "mean" if self.readout != "sum" else "add")
        fcs = self.MLP_layer.FC_layers
        for i, fc in enumerate(fcs):
            hg = AG.linear(hg, fc.weight, fc.bias, relu=i < len(fcs) - 1)
        self._h_last = x.detach()
        return hg

    def _fc(self, L):
        c = self.__dict__.setdefault("_cache", {})
        if self.training or ("fc", id(L)) not in c:
            w = L.fc.weight.detach()
            pl = ops.PackedLinear(ops.pack_weight(w), w.shape[0], w.shape[1], None)
            if self.training:
                return pl
            c[("fc", id(L))] = pl
        return c[("fc", id(L))]

    def _const(self, v, n):
        c = self.__dict__.setdefault("_cache", {})
        key = ("const", float(v), int(n))
        if key not in c:
            c[key] = torch.full((n,), float(v), dtype=torch.float32, device=self.embedding_p.weight.device)
        return c[key]

    def loss(self, scores, targets):
        return (scores - targets).abs().mean()
