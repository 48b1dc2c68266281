"""DGL-tree sign-invariant networks (drop-in `nn.Module` surface, HIP forward).

Mirrors GraphPrediction/layers/deepsigns.py:33-86 (GINDeepSigns, MaskedGINDeepSigns), layers/gnns.py:81-114 (GIN),
layers/mlp.py:5-56 (MLP) and the factory nets/ZINC_graph_regression/sign_inv_net.py:3-17 — same constructor
arguments, same `forward(g, x[N,K,1]) -> [N,K,1]`, same state_dict keys (`enc.layers.{l}.apply_func.lins.{i}.*`,
`enc.layers.{l}.eps`, `enc.bns.{l}.*`, `rho.lins.{i}.*`, `rho.bns.{i}.*`).

`g` is duck-typed: anything with `.edges() -> (src, dst)` int64 tensors and `.batch_num_nodes()` (a DGL batched
graph, or `signnet_basisnet_amd.dgl_deepsigns.Graph`).  Eval mode (BatchNorm with running statistics) runs
entirely in the HIP kernels of libsignnet_hip.so; there is no CPU path.  Train mode gives the forward VALUE with
batch-statistic BatchNorm (and updates the running statistics); there is no autograd.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops


class Graph:
    """Minimal batched graph: edge list + per-graph node counts (what the reference reads from a DGLGraph)."""

    def __init__(self, src, dst, batch_num_nodes, batch_num_edges=None):
        self.src, self.dst = src, dst
        self._bnn = torch.as_tensor(batch_num_nodes)
        self._bne = None if batch_num_edges is None else torch.as_tensor(batch_num_edges)    # per-graph edge counts, as DGL keeps them

    def edges(self):
        return self.src, self.dst

    def batch_num_nodes(self):
        return self._bnn

    def batch_num_edges(self):
        return self._bne

    def to(self, device):
        return Graph(self.src.to(device), self.dst.to(device), self._bnn.to(device), None if self._bne is None else self._bne.to(device))


class MLP(nn.Module):
    """Linear -> activation -> BatchNorm per hidden layer, final Linear (mlp.py:5-56); all Linears have bias."""

    def __init__(self, in_channels, hidden_channels, out_channels, num_layers, use_bn=False, use_ln=False, dropout=0.5,
                 activation="relu", residual=False):
        super().__init__()
        if use_ln or residual or activation != "relu":
            raise ValueError("HIP path: only relu / no LayerNorm / no residual (what every shipped config uses)")
        self.lins = nn.ModuleList()
        if use_bn:
            self.bns = nn.ModuleList()
        if num_layers == 1:
            self.lins.append(nn.Linear(in_channels, out_channels))
        else:
            self.lins.append(nn.Linear(in_channels, hidden_channels))
            if use_bn:
                self.bns.append(nn.BatchNorm1d(hidden_channels))
            for _ in range(num_layers - 2):
                self.lins.append(nn.Linear(hidden_channels, hidden_channels))
                if use_bn:
                    self.bns.append(nn.BatchNorm1d(hidden_channels))
            self.lins.append(nn.Linear(hidden_channels, out_channels))
        self.use_bn, self.dropout = use_bn, dropout


class _GINConv(nn.Module):
    """dgl.nn.pytorch.GINConv(apply_func, 'sum'): eps is a non-learned buffer initialised to 0."""

    def __init__(self, apply_func):
        super().__init__()
        self.apply_func = apply_func
        self.register_buffer("eps", torch.zeros(1))


class GIN(nn.Module):
    def __init__(self, in_channels, hidden_channels, out_channels, n_layers, use_bn=True, dropout=0.5, activation="relu"):
        super().__init__()
        self.layers = nn.ModuleList()
        if use_bn:
            self.bns = nn.ModuleList()
        self.use_bn = use_bn
        self.layers.append(_GINConv(MLP(in_channels, hidden_channels, hidden_channels, 2, use_bn=use_bn, dropout=dropout,
                                        activation=activation)))
        for _ in range(n_layers - 2):
            self.layers.append(_GINConv(MLP(hidden_channels, hidden_channels, hidden_channels, 2, use_bn=use_bn,
                                            dropout=dropout, activation=activation)))
            if use_bn:
                self.bns.append(nn.BatchNorm1d(hidden_channels))
        self.layers.append(_GINConv(MLP(hidden_channels, hidden_channels, out_channels, 2, use_bn=use_bn, dropout=dropout,
                                        activation=activation)))
        if use_bn:
            self.bns.append(nn.BatchNorm1d(hidden_channels))


def _pack(lin):
    w = lin.weight.detach()
    return ops.PackedLinear(ops.pack_weight(w), w.shape[0], w.shape[1], None if lin.bias is None else lin.bias.detach().contiguous())


class _BNSite:
    """A BatchNorm1d of the layer path: eval -> running statistics folded to (scale, shift) once; train -> the module,
    whose batch statistics are taken per forward (and whose running statistics are updated)."""
    __slots__ = ("mod", "scale", "shift")

    def __init__(self, bn, train):
        self.mod = bn
        self.scale, self.shift = (None, None) if train else ops.bn_fold(bn)

    def affine(self, y, train):
        """(scale, shift) to apply to the rows `y` [R, C] (train: from y's own column statistics)."""
        if not train:
            return self.scale, self.shift
        m = self.mod
        mean, var, count = ops.masked_colstats(y)
        ops.bn_running_update(m, mean, var, count)
        return ops.bn_fold_stats(None if m.weight is None else m.weight.detach(), None if m.bias is None else m.bias.detach(),
                                 mean, var, m.eps)


def _prep_mlp(mlp: MLP, train=False):
    """[(packed linear, BN site or None)] per layer."""
    out = []
    for i, lin in enumerate(mlp.lins):
        bn = _BNSite(mlp.bns[i], train) if (mlp.use_bn and i < len(mlp.lins) - 1) else None
        out.append((_pack(lin), bn))
    return out


def _run_mlp(prep, x, nvalid=None, K=0, tail_bn=None, train=False):
    """mlp.py:37-56 (dropout 0): hidden layers = bias -> relu -> BN; the final Linear optionally followed by the
    BatchNorm that GIN.forward applies before the NEXT GINConv (gnns.py:105-112; in eval folded into the GEMM epilogue).
    train: BatchNorm over ALL rows of the layer (the reference normalises [N, C, K] tensors, padded slots included)."""
    for i, (pl, bn) in enumerate(prep):
        last = i == len(prep) - 1
        site = tail_bn if last else bn
        if not train:
            sc, sh = (None, None) if site is None else (site.scale, site.shift)
            x = ops.masked_linear(x, pl, nvalid, K, relu_pre=not last, scale=sc, shift=sh)
        else:
            x = ops.masked_linear(x, pl, nvalid, K, relu=not last)
            if site is not None:
                sc, sh = site.affine(x, True)
                x = ops.masked_affine(x, scale=sc, shift=sh)
    return x


def _no_dropout(dropout, who):
    """The reference applies F.dropout(p=dropout) in MLP.forward / GIN.forward while training (mlp.py:52, gnns.py:104); the shipped
    sign_inv configs all set dropout 0.0.  A non-zero value is refused instead of being silently ignored."""
    if float(dropout) != 0.0:
        raise NotImplementedError(f"{who}: dropout={dropout} is not implemented by the HIP modules (the shipped configs use 0.0); "
                                  "pass dropout=0.0")


def _pad_mat(W, dp):
    out = torch.zeros(dp, dp, dtype=torch.float32, device=W.device)
    out[:W.shape[0], :W.shape[1]].copy_(W.detach())
    return out


def _fold_bn_before(lin, site):
    """Linear(BatchNorm(h)) with an eval BatchNorm (scale, shift) -> one Linear: W' = W diag(scale), b' = b + W shift.  On the device
    ops (a column scaling and one [1, d] GEMM), run once per parameter version."""
    W, b = lin.weight.detach().contiguous(), lin.bias.detach().contiguous()
    if site is None:
        return W, b
    Wf = ops.masked_affine(W, scale=site.scale, shift=torch.zeros_like(site.shift))
    bf = ops.masked_linear(site.shift.view(1, -1).contiguous(), ops.PackedLinear(ops.pack_weight(W), W.shape[0], W.shape[1], b)).view(-1)
    return Wf, bf


class _FusedDeepSigns:
    """Packed eval-mode parameters of a (Masked)GINDeepSigns for the two stage kernels: sn_deepsigns_phi_f32 (enc(g,x) + enc(g,-x),
    all GIN layers, one launch) and sn_mlp_chain_f32 (rho, with the masked slot sum in front for the masked variant).  With
    sn_batch_plan that is three launches per forward (deepsigns.py:45-51 / :72-86).  `ok` is False for shapes the stage kernels do
    not take (hidden > 112, k > 64, k * phi_out > 128, MLPs that are not the reference's 2-layer GIN MLP): the layer path serves those."""

    def __init__(self, mod):
        from .fused import _PhiParams
        import ctypes as C
        enc, rho, K = mod.enc, mod.rho, mod.k
        self.ok = False
        L = len(enc.layers)
        mlps = [c.apply_func for c in enc.layers]
        if any(len(m.lins) != 2 for m in mlps) or mlps[0].lins[0].weight.shape[1] != 1:
            return
        hidden = mlps[0].lins[0].weight.shape[0]
        out = mlps[-1].lins[1].weight.shape[0]
        dp = 16 * ((max(hidden, out) + 15) // 16)
        dp = max(dp, 48)
        d_in = out if mod.masked else K * out
        rdims = [d_in] + [l.weight.shape[0] for l in rho.lins]
        rp = max(48, 16 * ((max(rdims) + 15) // 16))
        if not (dp <= 112 and rp <= 128 and 1 <= L <= 16 and K <= 64 and len(rho.lins) <= 16):
            return
        keep = self._keep = []

        def hold(t):
            keep.append(t)
            return t.data_ptr()

        dev = mlps[0].lins[0].weight.device
        ones = ops.pad_vec(torch.ones(hidden, device=dev), dp)
        P = _PhiParams()
        P.d, P.n_layers, P.hid0, P.reserved = dp, L, dp, out
        for l, m in enumerate(mlps):
            site_mid = _BNSite(m.bns[0], False) if m.use_bn else None                       # after the ReLU, before lins[1]
            site_out = _BNSite(enc.bns[l], False) if (enc.use_bn and l < L - 1) else None   # in front of layer l + 1 (gnns.py:105-110)
            W1, b1 = _fold_bn_before(m.lins[1], site_mid)
            no = W1.shape[0]
            e1 = ops.pad_vec(site_out.scale if site_out is not None else torch.ones(no, device=dev), dp)
            e2 = ops.pad_vec(site_out.shift if site_out is not None else torch.zeros(no, device=dev), dp)
            w2 = ops.pack_split(_pad_mat(W1, dp), ops.pad_vec(b1, dp), e1, e2)
            b0 = ops.pad_vec(m.lins[0].bias, dp)
            if l == 0:
                P.l0_w1 = hold(ops.pad_vec(m.lins[0].weight[:, 0], dp))
                P.l0_bn0_scale, P.l0_bn0_shift = hold(ones), hold(b0)
                P.l0_w2 = hold(w2)
                P.l0_eps = hold(enc.layers[0].eps.detach().float().contiguous())
            else:
                Lp = P.layers[l - 1]
                Lp.w1s = hold(ops.pack_split(_pad_mat(m.lins[0].weight, dp), ones, b0, None))
                Lp.w2s = hold(w2)
                Lp.eps = hold(enc.layers[l].eps.detach().float().contiguous())
        self.phi = P
        # rho: relu(W0 x + b0), then every later Linear with the BatchNorm in front of it folded in
        ws = []
        for i, lin in enumerate(rho.lins):
            site = _BNSite(rho.bns[i - 1], False) if (rho.use_bn and i > 0) else None
            W, b = _fold_bn_before(lin, site)
            ws.append(hold(ops.pack_split(_pad_mat(W, rp), ops.pad_vec(b, rp), None, None)))
        self.rho_w = (C.c_void_p * len(ws))(*ws)
        self.n_rho, self.rp, self.d_in, self.out, self.K, self.masked = len(ws), rp, d_in, out, K, mod.masked
        self.ok = True

    def run(self, plan, x, N):
        import ctypes as C
        from ._lib import check, lib, ptr, stream
        K, out = self.K, self.out
        z = torch.empty(N * K, out, dtype=torch.float32, device=x.device)
        with ops._span("sn_deepsigns_phi_f32"):
            check(lib().sn_deepsigns_phi_f32(C.byref(self.phi), ptr(x), K, ptr(plan.graph_ptr), ptr(plan.rowptr), ptr(plan.col),
                                             C.byref(plan.bins.cstruct), K, ptr(z), stream()), "sn_deepsigns_phi_f32")
        y = torch.empty(N, K, dtype=torch.float32, device=x.device)
        with ops._span("sn_mlp_chain_f32"):
            if self.masked:
                check(lib().sn_mlp_chain_f32(ptr(z), out, N, out, ptr(plan.nvalid), K, self.rho_w, self.n_rho, self.rp, ptr(y), K, K,
                                             stream()), "sn_mlp_chain_f32")
            else:
                check(lib().sn_mlp_chain_f32(ptr(z), K * out, N, K * out, None, 0, self.rho_w, self.n_rho, self.rp, ptr(y), K, K,
                                             stream()), "sn_mlp_chain_f32")
        return y, z


def _node_counts(g):
    """(largest graph, total nodes) of the batch, read ONCE per graph object — free when batch_num_nodes() is a host tensor, one
    two-scalar read otherwise — and cached on it."""
    mc = getattr(g, "_sn_node_counts", None)
    if mc is None:
        bnn = g.batch_num_nodes()
        mc = tuple(int(v) for v in torch.stack([bnn.max(), bnn.sum()]).tolist()) if bnn.numel() else (0, 0)
        try:
            g._sn_node_counts = mc
        except Exception:
            pass
    return mc


def _max_nodes(g):
    """Largest graph of the batch (the stage kernels keep a whole graph in one 64-row bin column)."""
    return _node_counts(g)[0]


def _max_in_edges(g):
    """Most edges of one graph of the batch (the one-launch GatedGCN kernel stages a graph's in-edges in LDS), or None when the graph
    object does not say: DGL's batch_num_edges() — per-graph counts DGL keeps with the batch, like batch_num_nodes() — read ONCE per
    graph object and cached on it.  Never derived from the edge list: that is a device reduction and a host wait per new graph object,
    which the one-launch path exists to avoid (the device-side guard covers graphs whose counts are not known)."""
    me = getattr(g, "_sn_max_edges", None)
    if me is None:
        bne = getattr(g, "batch_num_edges", None)
        t = bne() if callable(bne) else None
        if t is None:
            return None
        me = int(t.max()) if t.numel() else 0
        try:
            g._sn_max_edges = me
        except Exception:
            pass
    return me


def _await_side(g, plan=None):
    """The sign-invariant net may have built this graph's plan — and its own output — on its side stream (`overlap = True`): whoever picks
    the plan up on another stream first waits for that stream's event there and tells the allocator (once per graph object)."""
    pend = getattr(g, "_sn_side", None)
    if pend is None:
        return
    ev, side, tensors = pend
    cur = torch.cuda.current_stream()
    if cur.cuda_stream == side.cuda_stream:
        return
    cur.wait_event(ev)
    plans = [plan] if plan is not None else list(getattr(g, "_sn_plans", {}).values())
    for t in tuple(tensors) + tuple(a for pl in plans for a in (pl.graph_ptr, pl.evoff)):
        t.record_stream(cur)
    g._sn_side = None


def cached_plan(g, N, k=None):
    """The CSR / graph_ptr plan of a batched graph (sn_batch_plan), built once per graph OBJECT and kept on it — as DGL keeps a
    graph's sparse formats on the graph.  k: also lay out the stage kernels' work bins over all k eigenvector slots (kmax = -k);
    a request without k is served by any plan already on the graph (the CSR arrays do not depend on k), so the sign-invariant net
    and the base network that consumes its output share ONE launch per batch."""
    cache = getattr(g, "_sn_plans", None)
    if cache is None:
        cache = {}
        try:
            g._sn_plans = cache
        except Exception:
            pass
    key = ("bins", int(k)) if k else ("csr",)
    if key in cache or (k is None and cache):
        plan = cache[key] if key in cache else next(iter(cache.values()))
        _await_side(g, plan)
        return plan
    src, dst = g.edges()
    bnn = g.batch_num_nodes().to(src.device)
    B = int(bnn.numel())
    total = _node_counts(g)[1]
    if total != N:            # repeat_interleave(output_size=N) trusts N: a mismatch would be an uninitialised tail or a write past it
        raise ValueError(f"batch_num_nodes() sums to {total} but the feature matrix has {N} rows")
    batch = torch.repeat_interleave(torch.arange(B, device=src.device), bnn, output_size=N)
    ei = torch.stack([src.long(), dst.long()])
    plan = ops.build_plan(batch.long(), ei, B, -int(k), bins=True) if k else ops.build_plan(batch.long(), ei, B, 0)
    cache[key] = plan
    return plan


class _DeepSignsBase(nn.Module):
    masked = False

    def __init__(self):
        super().__init__()
        # fires also when a PARENT module's load_state_dict recurses into this one (its own override below does not)
        self.register_load_state_dict_post_hook(lambda m, keys=None: m._invalidate())

    fused_stages = True      # eval: the two stage kernels (3 launches); False forces the layer-at-a-time path

    def _invalidate(self):
        self._prep = None
        self._fused = None

    def train(self, mode=True):
        self._invalidate()
        return super().train(mode)

    def _apply(self, fn, *a, **k):
        self._invalidate()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._invalidate()
        return super().load_state_dict(*a, **k)

    def _prepare(self, train=False):
        enc = self.enc
        P = dict(gin=[], rho=_prep_mlp(self.rho, train))
        L = len(enc.layers)
        for l, conv in enumerate(enc.layers):
            nxt = _BNSite(enc.bns[l], train) if (enc.use_bn and l < L - 1) else None   # BN applied before layer l+1
            P["gin"].append(dict(eps=conv.eps, mlp=_prep_mlp(conv.apply_func, train), next_bn=nxt))
        return P

    def _plan(self, g, N, fused=False):
        src, dst = g.edges()
        bnn = g.batch_num_nodes().to(src.device)
        B = int(bnn.numel())
        # (the node total comes from the graph object's cached host counts: repeat_interleave without output_size would read the sum
        #  back from the device — a host wait per batch, and not recordable in a HIP graph)
        if _node_counts(g)[1] != N:
            raise ValueError("batch_num_nodes does not sum to the number of feature rows")
        batch = torch.repeat_interleave(torch.arange(B, device=src.device), bnn, output_size=N)      # index plumbing only
        if fused:     # kmax = -k: work bins over all k slots of every graph (zero-padded columns are evaluated like any other)
            return ops.build_plan(batch.long(), torch.stack([src.long(), dst.long()]), B, -self.k, bins=True)
        return ops.build_plan(batch.long(), torch.stack([src.long(), dst.long()]), B, self.k)

    def _phi(self, P, plan, x, N, K, train=False):
        """enc(g, x) + enc(g, -x): GIN.forward, gnns.py:102-114, twice."""
        outs = []
        for sign in (0, 1):
            h = x
            for l, Lp in enumerate(P["gin"]):
                a = ops.gin_aggregate(h.reshape(N, -1), plan, Lp["eps"], negate=(sign == 1 and l == 0))
                h = _run_mlp(Lp["mlp"], a.view(N * K, -1), tail_bn=Lp["next_bn"], train=train)
            outs.append(h)
        return outs

    def _forward_grad(self, g, x):
        """Differentiable train-mode forward: the same launches as the value path, recorded as
        torch.autograd.Function nodes whose backward are the hand-written adjoints of csrc/backward.hip, so that the
        gradient a DGL base network sends back into `p = sign_inv_net(g, pos_enc)` (train_ZINC_graph_regression.py:20-25)
        reaches these parameters.  Dropout is not applied (the shipped configs use 0.0)."""
        from . import autograd as AG
        N, K = x.shape[0], self.k
        plan = self._plan(g, N)
        src, dst = g.edges()
        rplan = self._plan(Graph(dst, src, g.batch_num_nodes()), N)              # out-edge CSR for the aggregation adjoint
        enc = self.enc

        def run_mlp(mlp, h, tail_bn=None):
            n = len(mlp.lins)
            for i, lin in enumerate(mlp.lins):
                last = i == n - 1
                h = AG.linear(h, lin.weight, lin.bias, relu=not last)             # mlp.py:40-46: bias -> relu -> BN
                bn = tail_bn if last else (mlp.bns[i] if mlp.use_bn else None)
                if bn is not None:
                    h = AG.bn_act(h, bn, relu=False)                               # statistics over ALL rows, as the reference
            return h

        outs = []
        L = len(enc.layers)
        x = x.contiguous().float()
        for sign in (0, 1):
            h = x
            for l, conv in enumerate(enc.layers):
                a = AG.gin_aggregate(h.reshape(N, -1), conv.eps, plan, rplan, negate=(sign == 1 and l == 0))
                h = run_mlp(conv.apply_func, a.view(N * K, -1), enc.bns[l] if (enc.use_bn and l < L - 1) else None)
            outs.append(h)
        if self.masked:
            z = AG.masked_add(outs[0], outs[1], plan.nvalid, K)
            y = run_mlp(self.rho, AG.slot_sum(z, N, K, plan.nvalid))
        else:
            z = AG.masked_add(outs[0], outs[1])
            y = run_mlp(self.rho, z.view(N, -1))
        return y.view(N, K, 1)

    def _phi_eval_merged(self, P, plan, x, N, K):
        """Eval: enc(g, x) and enc(g, -x) in ONE pass — the two signs ride in the feature axis ([N, 2, K, C]), so a layer is one
        aggregation launch and two Linears instead of two and four (folded BatchNorm is row-wise, the signs never mix until the
        final sum).  Returns phi(x) + phi(-x) as [N*K, c]."""
        if "neg1" not in P:
            P["neg1"] = (torch.full((1,), -1.0, device=x.device), torch.zeros(1, device=x.device))
        xm = ops.masked_affine(x.view(N * K, 1), scale=P["neg1"][0], shift=P["neg1"][1]).view(N, K, 1)      # -x
        h = torch.stack([x, xm], dim=1).contiguous()                                                        # [N, 2, K, 1]
        for Lp in P["gin"]:
            a = ops.gin_aggregate(h.reshape(N, -1), plan, Lp["eps"])
            h = _run_mlp(Lp["mlp"], a.view(N * 2 * K, -1), tail_bn=Lp["next_bn"])
        return ops.slot_sum(h.view(N * 2, -1), N, 2).view(N * K, -1)                                         # sum over the sign axis

    def _forward_side(self, g, xin, N, K):
        """`overlap = True` (opt-in, eval, stage kernels): the batch plan and the two stage launches are queued on a side stream of this
        module and the result is handed over with an event kept on the graph object; the base network that consumes it (any net of
        dgl_nets: its first `cached_plan(g, ...)` waits for the event on ITS stream) then runs behind it, and the NEXT batch's
        sign-invariant net — queued while that network is still running — shares the GPU with it.  The returned tensor is complete only
        for consumers that go through the graph's plan (the reference's loop does: train_ZINC_graph_regression.py:20-25 hands it
        straight to the model); anything else must `torch.cuda.current_stream().wait_event(g._sn_side[0])` first.  Precondition as for
        pyg.SignNetGNN.overlap_front: the graph's tensors and x are complete on the device when forward is called."""
        from . import _lib as _lib_mod
        if getattr(self, "_side_stream", None) is None:
            self._side_stream = torch.cuda.Stream(device=xin.device)
        side = self._side_stream
        if getattr(self, "_side_wait", False):
            # x was converted (dtype / layout) by a kernel queued on the caller's stream a moment ago: the side stream must see it
            side.wait_stream(torch.cuda.current_stream(xin.device))
        with torch.cuda.stream(side), _lib_mod.stream_scope():
            y, z = self._fused.run(cached_plan(g, N, K), xin, N)
            ev = torch.cuda.Event()
            ev.record(side)
        src, dst = g.edges()
        for t in (xin, src, dst):
            t.record_stream(side)
        y = y.view(N, K, 1)
        g._sn_side = (ev, side, (y,))
        return y

    def forward(self, g, x):
        train = self.training     # train: batch statistics + running-statistics update (dropout 0)
        ops.require_cuda(x)
        if x.dim() != 3 or x.shape[1] != self.k or x.shape[2] != 1:
            raise ValueError(f"expected x of shape [N, {self.k}, 1]")
        if train and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return self._forward_grad(g, x)
        if train:
            P = self._prepare(True)          # parameters may change between training-mode calls: nothing is cached
        else:
            if getattr(self, "_prep", None) is None:
                self._prep = self._prepare()
            P = self._prep
        N, K = x.shape[0], self.k
        if not train and self.fused_stages:
            if getattr(self, "_fused", None) is None:
                self._fused = _FusedDeepSigns(self)
            if self._fused.ok and N > 0 and _max_nodes(g) <= ops.PHI_BIN_ROWS and int(g.batch_num_nodes().numel()) <= 6144:
                xin = x.contiguous().float().view(N, K)
                if getattr(self, "overlap", False):
                    self._side_wait = xin.data_ptr() != x.data_ptr()     # a copy / cast was queued on the caller's stream
                    return self._forward_side(g, xin, N, K)
                y, _ = self._fused.run(cached_plan(g, N, K), xin, N)
                return y.view(N, K, 1)
        plan = self._plan(g, N)
        if not train:
            z = self._phi_eval_merged(P, plan, x.contiguous().float(), N, K)
            if self.masked:
                z = ops.masked_affine(z, plan.nvalid, K)                     # x[~mask] = 0       (deepsigns.py:76-80)
        else:
            zp, zm = self._phi(P, plan, x.contiguous().float(), N, K, train)
            z = ops.masked_affine(zp, plan.nvalid, K, residual=zm) if self.masked else ops.masked_affine(zp, residual=zm)
        if self.masked:
            # sum over K ; rho(c -> hidden -> K)                              (deepsigns.py:81-84)
            y = _run_mlp(P["rho"], ops.slot_sum(z, N, K), train=train)
        else:
            y = _run_mlp(P["rho"], z.view(N, -1), train=train)                 # deepsigns.py:47-49
        return y.view(N, K, 1)


class GINDeepSigns(_DeepSignsBase):
    def __init__(self, in_channels, hidden_channels, out_channels, num_layers, k, use_bn=False, use_ln=False, dropout=0.5,
                 activation="relu"):
        super().__init__()
        _no_dropout(dropout, "GINDeepSigns")
        self.enc = GIN(in_channels, hidden_channels, out_channels, num_layers, use_bn=use_bn, dropout=dropout, activation=activation)
        self.rho = MLP(out_channels * k, hidden_channels, k, num_layers, use_bn=use_bn, dropout=dropout, activation=activation)
        self.k = k
        self._prep = None


class MaskedGINDeepSigns(_DeepSignsBase):
    masked = True

    def __init__(self, in_channels, hidden_channels, out_channels, num_layers, k, device=None, use_bn=False, use_ln=False,
                 dropout=0.5, activation="relu"):
        super().__init__()
        _no_dropout(dropout, "MaskedGINDeepSigns")
        self.device = device
        self.enc = GIN(in_channels, hidden_channels, out_channels, num_layers, use_bn=use_bn, dropout=dropout, activation=activation)
        self.rho = MLP(out_channels, hidden_channels, k, num_layers, use_bn=use_bn, dropout=dropout, activation=activation)
        self.k = k
        self._prep = None


def get_sign_inv_net(net_params):
    """nets/ZINC_graph_regression/sign_inv_net.py:3-17 (the 'gin' and 'masked_gin' branches — the only ones a shipped
    config selects)."""
    assert net_params["sign_inv_net"] is not None, "did not specify sign inv net"
    kind = net_params["sign_inv_net"]
    args = (1, net_params["hidden_dim"], net_params["phi_out_dim"], net_params["sign_inv_layers"], net_params["pos_enc_dim"])
    kw = dict(use_bn=True, dropout=net_params["dropout"], activation=net_params["sign_inv_activation"])
    if kind == "gin":
        return GINDeepSigns(*args, **kw)
    if kind == "masked_gin":
        return MaskedGINDeepSigns(*args, net_params["device"], **kw)
    raise ValueError("Invalid sign inv net")
