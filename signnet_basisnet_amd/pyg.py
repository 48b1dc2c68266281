"""PyG-tree SignNet + GINE modules (drop-in `nn.Module` surface, HIP forward).

Mirrors the operator boundary of the reference's two PyG trees:
  Alchemy/sign_net/sign_net.py:120-132         SignNetGNN(node_feat, edge_feat, n_hid, n_out,
                                                nl_signnet, nl_gnn, nl_rho=4, ignore_eigval, gnn_type)
  GINESignNetPyG/core/sign_net.py:122-134      SignNetGNN(node_feat, edge_feat, n_hid, n_out,
                                                nl_signnet, nl_gnn)
Same constructor arguments, same `forward(data) -> [B, n_out]`, same `state_dict` keys
 so reference-trained weights load unchanged.  The module classes below only
hold parameters; all arithmetic runs in the HIP kernels of libsignnet_hip.so through
`signnet_basisnet_amd.ops` / the fused engine.  There is no CPU path: tensors must be on the GPU.
"""
from __future__ import annotations

import time

import torch
import torch.nn as nn

from . import _lib as _lib_mod
from . import fused, ops

import os as _os
MAX_FUSED_GRAPHS = 6144      # sn_batch_plan's work-bin limit per call
_NO_KERNEL_FLAGS = bool(_os.environ.get("SN_NO_KERNEL_FLAGS"))   # debugging: report flags with a stream copy instead
N_HEAD = 4          # TransformerEncoderLayer(nhid, n_head=4): sign_net.py:50 / core/sign_net.py:57
LN_EPS = 1e-6       # masked_layers.py:25


# ----------------------------------------------------------------------------- parameter holders
class MaskedBN(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.bn = nn.BatchNorm1d(c)

    def reset_parameters(self):
        self.bn.reset_parameters()


class MaskedLN(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.ln = nn.LayerNorm(c, eps=LN_EPS)

    def reset_parameters(self):
        self.ln.reset_parameters()


class _Identity(nn.Module):
    def reset_parameters(self):
        pass


def _mlp_layers(nin, nout, nlayer, final_act, with_norm, bias, nhid):
    """Layer widths / bias rule shared by MaskedMLP (masked_layers.py:35-47) and MLP (elements.py:40-51)."""
    h = nin if nhid is None else nhid
    dims = [(nin if i == 0 else h, h if i < nlayer - 1 else nout) for i in range(nlayer)]
    has_bias = [bool((i == nlayer - 1 and not final_act and bias) or (not with_norm)) for i in range(nlayer)]
    return dims, has_bias


class MaskedMLP(nn.Module):
    def __init__(self, nin, nout, nlayer=2, with_final_activation=True, with_norm=True, bias=True, nhid=None):
        super().__init__()
        dims, hb = _mlp_layers(nin, nout, nlayer, with_final_activation, with_norm, bias, nhid)
        self.layers = nn.ModuleList([nn.Linear(a, b, bias=h) for (a, b), h in zip(dims, hb)])
        self.norms = nn.ModuleList([MaskedBN(b) if with_norm else _Identity() for a, b in dims])
        self.nlayer, self.with_final_activation = nlayer, with_final_activation

    def reset_parameters(self):
        for l, n in zip(self.layers, self.norms):
            l.reset_parameters()
            n.reset_parameters()


class MLP(nn.Module):
    def __init__(self, nin, nout, nlayer=2, with_final_activation=True, with_norm=True, bias=True, nhid=None):
        super().__init__()
        dims, hb = _mlp_layers(nin, nout, nlayer, with_final_activation, with_norm, bias, nhid)
        self.layers = nn.ModuleList([nn.Linear(a, b, bias=h) for (a, b), h in zip(dims, hb)])
        self.norms = nn.ModuleList([nn.BatchNorm1d(b) if with_norm else _Identity() for a, b in dims])
        self.nlayer, self.with_final_activation = nlayer, with_final_activation

    def reset_parameters(self):
        for l, n in zip(self.layers, self.norms):
            l.reset_parameters()
            n.reset_parameters()


class DiscreteEncoder(nn.Module):
    def __init__(self, hidden, max_num_features=10, max_num_values=500):
        super().__init__()
        self.embeddings = nn.ModuleList([nn.Embedding(max_num_values, hidden) for _ in range(max_num_features)])

    def reset_parameters(self):
        for e in self.embeddings:
            e.reset_parameters()


class _GINEps(nn.Module):
    """Holder of PyG GINConv/GINEConv's `eps` (train_eps=True, initial 0) and, for GINE, the MLP
    that PyG registers a second time under `.layer.nn` (pyg_gnn_wrapper.py:22-23)."""
    def __init__(self, mlp=None):
        super().__init__()
        if mlp is not None:
            self.nn = mlp
        self.eps = nn.Parameter(torch.zeros(1))

    def reset_parameters(self):
        self.eps.data.fill_(0.0)


class MaskedGINConv(nn.Module):
    def __init__(self, nin, nout, bias=True, nhid=None):
        super().__init__()
        self.nn = MaskedMLP(nin, nout, 2, False, bias=bias, nhid=nhid)
        self.layer = _GINEps()

    def reset_parameters(self):
        self.nn.reset_parameters()
        self.layer.reset_parameters()


class GINEConv(nn.Module):
    def __init__(self, nin, nout, bias=True):
        super().__init__()
        self.nn = MLP(nin, nout, 2, False, bias=bias)
        self.layer = _GINEps(self.nn)

    def reset_parameters(self):
        self.nn.reset_parameters()
        self.layer.reset_parameters()


class GNN3d(nn.Module):
    def __init__(self, n_in, n_out, n_layer, variant):
        super().__init__()
        if variant == "alchemy":   # sign_net.py:20
            self.convs = nn.ModuleList([MaskedGINConv(n_in if i == 0 else n_out, n_out, bias=True, nhid=n_out)
                                        for i in range(n_layer)])
        else:                      # core/sign_net.py:20
            self.convs = nn.ModuleList([MaskedGINConv(n_in if i == 0 else n_out, n_out, bias=False)
                                        for i in range(n_layer)])
        self.norms = nn.ModuleList([MaskedBN(n_out) for _ in range(n_layer)])
        if variant != "alchemy":   # registered but never used by the reference (core/sign_net.py:22,40)
            self.edge_encoders = nn.ModuleList([DiscreteEncoder(n_in if i == 0 else n_out) for i in range(n_layer)])

    def reset_parameters(self):
        for m in list(self.convs) + list(self.norms) + list(getattr(self, "edge_encoders", [])):
            m.reset_parameters()


class _MHA(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.w_qs = nn.Linear(d, d, bias=False)
        self.w_ks = nn.Linear(d, d, bias=False)
        self.w_vs = nn.Linear(d, d, bias=False)
        self.fc = nn.Linear(d, d, bias=False)
        self.norm = MaskedLN(d)


class _FFN(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.w_1 = nn.Linear(d, d)
        self.w_2 = nn.Linear(d, d)
        self.norm = MaskedLN(d)


class TransformerEncoderLayer(nn.Module):
    def __init__(self, d):
        super().__init__()
        if d % N_HEAD:
            raise ValueError(f"hidden width {d} must be divisible by the {N_HEAD} attention heads")
        self.slf_attn = _MHA(d)
        self.pos_ffn = _FFN(d)


class SetTransformer(nn.Module):
    def __init__(self, nhid, nlayer, variant):
        super().__init__()
        if variant != "alchemy":   # unused parameter of the reference (core/sign_net.py:54)
            self.pos_encoder = MaskedMLP(1, nhid, nlayer=2)
        self.transformer_layers = nn.ModuleList(TransformerEncoderLayer(nhid) for _ in range(nlayer))
        self.out = nn.Sequential(nn.Linear(nhid, nhid, bias=False), nn.BatchNorm1d(nhid))

    def reset_parameters(self):
        if hasattr(self, "pos_encoder"):
            self.pos_encoder.reset_parameters()


class SignNet(nn.Module):
    def __init__(self, n_hid, nl_phi, nl_rho, variant, ignore_eigval=False):
        super().__init__()
        self.variant = variant
        self.phi = GNN3d(1, n_hid, nl_phi, variant)
        self.rho = SetTransformer(n_hid, nl_rho, variant)
        self.ignore_eigval = ignore_eigval
        if variant == "alchemy":
            if not ignore_eigval:
                self.eigen_encoder = MaskedMLP(1, n_hid, nlayer=2)
        else:
            self.eigen_encoder1 = MaskedMLP(1, n_hid, nlayer=1)
            self.eigen_encoder2 = MaskedMLP(1, n_hid, nlayer=2)

    def reset_parameters(self):
        self.phi.reset_parameters()
        self.rho.reset_parameters()
        for n in ("eigen_encoder", "eigen_encoder1", "eigen_encoder2"):
            if hasattr(self, n):
                getattr(self, n).reset_parameters()


class GNN(nn.Module):
    def __init__(self, nfeat_node, nfeat_edge, nhid, nout, nlayer, variant, pooling="add"):
        super().__init__()
        nv = 6 if variant == "alchemy" else 500      # elements.py:22
        self.input_encoder = DiscreteEncoder(nhid, max_num_values=nv) if nfeat_node is None else MLP(nfeat_node, nhid, 1)
        self.edge_encoders = nn.ModuleList([DiscreteEncoder(nhid, max_num_values=nv) if nfeat_edge is None
                                            else MLP(nfeat_edge, nhid, 1) for _ in range(nlayer)])
        self.convs = nn.ModuleList([GINEConv(nhid, nhid, bias=False) for _ in range(nlayer)])
        self.norms = nn.ModuleList([nn.BatchNorm1d(nhid) for _ in range(nlayer)])
        self.output_encoder = MLP(nhid, nout, nlayer=2, with_final_activation=False,
                                  with_norm=False if pooling == "mean" else True)
        if variant != "alchemy":   # core/model.py:18
            self.size_embedder = nn.Embedding(200, nhid)
        self.linear = nn.Linear(2 * nhid, nhid)
        self.pooling = pooling

    def reset_parameters(self):
        self.input_encoder.reset_parameters()
        self.output_encoder.reset_parameters()
        if hasattr(self, "size_embedder"):
            self.size_embedder.reset_parameters()
        self.linear.reset_parameters()
        for e, c, n in zip(self.edge_encoders, self.convs, self.norms):
            e.reset_parameters()
            c.reset_parameters()
            n.reset_parameters()


# ----------------------------------------------------------------------------- prepared (packed) parameters
class _BN:
    """A BatchNorm1d site of the layer path: the module (train mode: batch statistics) and, in eval mode, its running
    statistics folded to y = x*scale + shift on the device."""
    __slots__ = ("mod", "scale", "shift")

    def __init__(self, bn: nn.BatchNorm1d, fold: bool):
        self.mod = bn
        self.scale, self.shift = ops.bn_fold(bn) if fold else (None, None)

    def __getitem__(self, i):
        return (self.scale, self.shift)[i]


def _lin_bn(x, pl, bn: _BN, train: bool, nvalid=None, K=0, relu=True, residual=None):
    """Linear -> [mask] -> BatchNorm -> [ReLU] [-> + residual].  eval: one launch (BN folded into the GEMM epilogue).
    train: GEMM, masked column statistics over the valid rows, fold, affine (+ running-statistics update) — the
    reference's `x[mask] = bn(x[mask])` in training mode (masked_layers.py:13-20) / plain BatchNorm1d (model.py:50)."""
    if not train:
        return ops.masked_linear(x, pl, nvalid, K, scale=bn.scale, shift=bn.shift, relu=relu, residual=residual)
    y = ops.masked_linear(x, pl, nvalid, K)
    m = bn.mod
    mean, var, count = ops.masked_colstats(y, nvalid, K)
    sc, sh = ops.bn_fold_stats(None if m.weight is None else m.weight.detach(), None if m.bias is None else m.bias.detach(),
                               mean, var, m.eps)
    ops.bn_running_update(m, mean, var, count)
    return ops.masked_affine(y, nvalid, K, scale=sc, shift=sh, relu=relu, residual=residual)


def _pack(lin: nn.Linear) -> ops.PackedLinear:
    w = lin.weight.detach()
    return ops.PackedLinear(ops.pack_weight(w), w.shape[0], w.shape[1],
                            None if lin.bias is None else lin.bias.detach().contiguous())


def _drop_prepared(module, incompatible_keys=None):
    """load_state_dict post hook: the packed (eval-mode) weights are stale."""
    module._prep = None


def host_max_nodes(data):
    """The largest graph of a batch from HOST-side bookkeeping the batch object already carries, or None: `data.sizes` (a list of
    node counts: synth.make_batch, dist.shard_batch), PyG's `Batch._slice_dict` (CPU tensors of per-graph offsets that collate builds
    and `.to(device)` leaves on the host) or a CPU `ptr`.  With it the all-eigenvector mode (max_k=None: K = N_max, core/transform.py:
    29-38) sizes its tensors without reading a graph size back from the device (the reference does: `int(num_nodes.max())`)."""
    N, B = int(data.batch.numel()), int(data.num_graphs)
    s = getattr(data, "sizes", None)
    if s is not None and not torch.is_tensor(s):
        # (bookkeeping that does not even add up to the batch is ignored: the device is asked instead)
        return (int(max(s)) if len(s) else 0) if (len(s) == B and sum(s) == N) else None
    cands = []
    sd = getattr(data, "_slice_dict", None)
    if isinstance(sd, dict):
        cands += [sd.get(key) for key in ("x", "eigen_values", "batch")]
    cands.append(getattr(data, "ptr", None))
    for t in cands:
        if torch.is_tensor(t) and not t.is_cuda and t.dim() == 1 and t.numel() == B + 1 and B >= 1 and int(t[-1]) - int(t[0]) == N:
            return int((t[1:] - t[:-1]).max())
    return None


class SignNetGNN(nn.Module):
    """HIP SignNet + GINE.  `variant` selects which reference tree's semantics are reproduced.

    Extra keyword (not in the reference): `max_k` — use only the first `max_k` eigenvectors
    (BASELINE.json's "k=16" reading); None = all eigenvectors as the reference does.
    """

    def __init__(self, node_feat, edge_feat, n_hid, n_out, nl_signnet, nl_gnn, nl_rho=4, ignore_eigval=False,
                 gnn_type="GINEConv", *, variant="alchemy", max_k=None):
        super().__init__()
        if gnn_type != "GINEConv":
            raise ValueError("only gnn_type='GINEConv' is on the SignNet hot path (SURVEY.md §2 row 3)")
        if variant not in ("alchemy", "gine"):
            raise ValueError("variant must be 'alchemy' or 'gine'")
        self.variant, self.max_k = variant, max_k
        self.cfg = dict(node_feat=node_feat, edge_feat=edge_feat, n_hid=n_hid, n_out=n_out,
                        nl_signnet=nl_signnet, nl_gnn=nl_gnn)
        # nl_rho is fixed by the reference constructors (sign_net.py:123 ignores the argument)
        self.nl_rho = 4 if variant == "alchemy" else 1
        self.use_fused = True       # whole-stage kernels (eval mode); False = layer-at-a-time kernels only
        self.train_stages = True    # training: the one-pass link kernels of train_stage.py; False = one launch per op (autograd.py)
        # Fused stages require graphs of <= 64 nodes (and <= 192 edges for the GINE stage); a batch that violates this, a
        # malformed batch, or a discrete feature value outside its embedding table raises device-side flags.
        #   strict=True : (default) the flags are read in every forward; an oversize batch is re-run on the layer path, anything
        #                 else raises on the spot — every input the reference evaluates is evaluated, errors are the reference's
        #                 errors.  Since round 5 the flags are DECIDED BY THE PLAN KERNEL, the first ~20 us of a forward (graph
        #                 sizes, in-edge counts, malformed batch, feature ids against their embedding tables), which writes them
        #                 to pinned memory itself (ops.EarlyReport / sn_batch_plan_ex): the host queues the three stage kernels
        #                 behind the plan, then polls words that are already there — the GPU never waits for the host (until
        #                 round 4 the GINE kernel reported them at the END of the forward: one host round trip per step, 9 %).
        #                 Batches beyond the one-launch plan (> 4096 nodes / 12288 edges / 1024 graphs) keep that older wait.
        #   strict=False: the serving / throughput mode (bench.py): no host wait.  The outputs of every graph that could not
        #                 be evaluated are NaN (never uninitialised memory: the GINE kernel fills them), and the error itself
        #                 is raised at the next forward, at check_last() or at train()/eval(), whichever comes first (a
        #                 module deleted with an unreported error warns).
        self.strict = True
        # overlap_front: see _forward_overlapped (serving loops over resident batches; needs strict = False)
        self.overlap_front, self.overlap_inputs_ready, self._side_streams = False, True, None
        # train-mode dropout of the attention probabilities (ScaledDotProductAttention's default attn_dropout=0.1,
        # transformer_module.py:46-55 — the only dropout the reference leaves active); 0.0 switches it off
        self.attn_dropout = 0.1
        self._pending, self._free_hosts, self._flags_host = [], [], None
        self.sign_net = SignNet(n_hid, nl_signnet, self.nl_rho, variant, ignore_eigval)
        self.gnn = GNN(node_feat, edge_feat, n_hid, n_out, nl_gnn, variant)
        self._prep = None
        # nn.Module.load_state_dict on a PARENT (a wrapper, DDP, a bigger model) recurses through _load_from_state_dict and never
        # calls this module's load_state_dict override — the post hook below does fire for every submodule of that recursion
        self.register_load_state_dict_post_hook(_drop_prepared)

    def reset_parameters(self):
        self.sign_net.reset_parameters()
        self.gnn.reset_parameters()
        self._prep = None

    # cache invalidation: packed weights depend on parameters, buffers, device and mode
    def train(self, mode=True):
        self._prep = None
        out = super().train(mode)       # switch the whole tree first: a pending error must not leave it half-switched
        self._drain_status()
        return out

    def _apply(self, fn, *a, **k):
        self._prep = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._prep = None
        return super().load_state_dict(*a, **k)

    def invalidate(self):
        """Call after modifying parameters in place while in eval mode (an optimiser step, `p.data.copy_()`, ...): the packed
        copies are rebuilt on the next forward.  train()/eval(), .to()/.cuda(), load_state_dict (also through a parent module)
        and reset_parameters() do this themselves."""
        self._prep = None

    def __del__(self):
        # Python never propagates an exception out of __del__: an unreported error of the last batch is WARNED about here
        try:
            if self._pending:
                self._drain_status()
        except (RuntimeError, IndexError, ValueError) as e:
            try:
                import warnings
                warnings.warn(f"SignNetGNN deleted with an unreported device error: {e}", RuntimeWarning)
            except Exception:
                pass
        except Exception:               # interpreter shutdown etc.
            pass

    def _drain_status(self):
        if getattr(self, "_pending", None):
            self.check_last()

    # ------------------------------------------------------------------ prepare
    def _prepare(self, train=False):
        P = {}
        fold = not train
        use_fused = self.use_fused and not train
        sn, g = self.sign_net, self.gnn
        d = self.cfg["n_hid"]
        P["phi_fused"] = fused.PhiPlan(sn.phi) if (use_fused and d <= 128 and d % 4 == 0 and len(sn.phi.convs) <= 16) else None
        ee = sn.eigen_encoder if (self.variant == "alchemy" and not sn.ignore_eigval) else None
        P["rho_fused"] = (fused.RhoPlan(sn.rho, ee, N_HEAD, LN_EPS)
                          if (use_fused and d <= 128 and d % 4 == 0 and len(sn.rho.transformer_layers) <= fused.RHO_MAX_LAYERS) else None)
        P["gnn_fused"] = None
        if use_fused and d <= 128 and len(g.convs) <= fused.GNN_MAX_LAYERS and self.cfg["n_out"] <= 16 \
                and (self.cfg["node_feat"] or 0) <= 16 and (self.cfg["edge_feat"] or 0) <= 16:
            P["gnn_fused"] = fused.GnnPlan(sn.rho.out, g, self.cfg["node_feat"], self.cfg["edge_feat"])
        P["phi"] = []
        for conv, norm in zip(sn.phi.convs, sn.phi.norms):
            P["phi"].append(dict(eps=conv.layer.eps.detach(), l0=_pack(conv.nn.layers[0]), bn0=_BN(conv.nn.norms[0].bn, fold),
                                 l1=_pack(conv.nn.layers[1]), bn=_BN(norm.bn, fold)))
        if self.variant == "alchemy" and not sn.ignore_eigval:
            ee = sn.eigen_encoder
            P["eig"] = dict(l0=_pack(ee.layers[0]), bn0=_BN(ee.norms[0].bn, fold), l1=_pack(ee.layers[1]),
                            bn1=_BN(ee.norms[1].bn, fold))
        if train and self.variant != "alchemy":
            # GINESignNetPyG evaluates eigen_encoder2 and throws the result away (core/sign_net.py:111-112); in training
            # mode that call still moves the running statistics of its two BatchNorms — reproduced, value discarded
            ee2 = sn.eigen_encoder2
            P["eig2"] = dict(l0=_pack(ee2.layers[0]), bn0=_BN(ee2.norms[0].bn, False), l1=_pack(ee2.layers[1]),
                             bn1=_BN(ee2.norms[1].bn, False))
        P["rho"] = []
        for tl in sn.rho.transformer_layers:
            a, f = tl.slf_attn, tl.pos_ffn
            P["rho"].append(dict(q=_pack(a.w_qs), k=_pack(a.w_ks), v=_pack(a.w_vs), fc=_pack(a.fc),
                                 ln1=(a.norm.ln.weight.detach(), a.norm.ln.bias.detach()),
                                 w1=_pack(f.w_1), w2=_pack(f.w_2),
                                 ln2=(f.norm.ln.weight.detach(), f.norm.ln.bias.detach())))
        P["rho_out"] = dict(l=_pack(sn.rho.out[0]), bn=_BN(sn.rho.out[1], fold))
        if isinstance(g.input_encoder, DiscreteEncoder):
            P["in_tabs"] = [e.weight.detach() for e in g.input_encoder.embeddings]
        else:
            P["in_mlp"] = dict(l=_pack(g.input_encoder.layers[0]), bn=_BN(g.input_encoder.norms[0], fold))
        P["lin"] = _pack(g.linear)
        P["gine"] = []
        for enc, conv, norm in zip(g.edge_encoders, g.convs, g.norms):
            d = dict(eps=conv.layer.eps.detach(), l0=_pack(conv.nn.layers[0]), bn0=_BN(conv.nn.norms[0], fold),
                     l1=_pack(conv.nn.layers[1]), bn=_BN(norm, fold))
            if isinstance(enc, DiscreteEncoder):
                d["tabs"] = [e.weight.detach() for e in enc.embeddings]
            else:
                d["emlp"] = dict(l=_pack(enc.layers[0]), bn=_BN(enc.norms[0], fold))
            P["gine"].append(d)
        oe = g.output_encoder
        P["head"] = dict(l0=_pack(oe.layers[0]), bn0=_BN(oe.norms[0], fold), l1=_pack(oe.layers[1]))
        return P

    # ------------------------------------------------------------------ device-side status of the fused stages
    @staticmethod
    def _flags_bad(host):
        # layout: [status(8) | meta(8)]: status[0] plan errors, status[3] gnn flags, status[5] embedding index (layer path),
        # meta[1] phi, meta[5] rho bin errors
        return bool(host[0] or host[3] or host[5] or host[9] or host[13] or (host[SignNetGNN._KWORD] and host[1] > host[SignNetGNN._KWORD]))

    @staticmethod
    def _flags_error(host):
        """The exception an offending batch deserves (same types as the reference raises: IndexError from nn.Embedding)."""
        if (host[3] & 4) or host[5]:
            return IndexError(ops.EMBEDDING_INDEX_ERROR + " — the outputs of the affected graphs are NaN")
        if host[SignNetGNN._KWORD] and host[1] > host[SignNetGNN._KWORD]:
            return ValueError(f"the batch's host-side graph sizes (largest: {host[SignNetGNN._KWORD]}) disagree with its batch vector (largest graph: "
                              f"{host[1]} nodes): eigenvector slots beyond the host-side count were dropped")
        if host[0]:
            return ValueError("an earlier batch was malformed (unsorted batch vector, graph id / edge endpoint out of range or an edge "
                              "across graphs): its outputs are NaN")
        if host[3] & 8:
            return RuntimeError("an earlier batch had a graph without nodes: the fused GINE stage returns NaN for it; set model.strict = True "
                                "(re-runs such batches layer by layer) or model.use_fused = False")
        return RuntimeError("an earlier batch had a graph too large for the fused SignNet kernels (> 64 nodes or > 192 edges): "
                            "the outputs of that batch are NaN; set model.strict = True (re-runs such batches layer by layer) or "
                            "model.use_fused = False")

    def check_last(self, wait=True):
        """Raise if an earlier forward's batch could not be served by the fused kernels (wait=False: only look at
        status reports that have already arrived)."""
        while self._pending:
            ev, host = self._pending[0]
            if not wait and len(self._pending) <= 4 and not self._arrived(ev, host):
                break
            self._wait_status(ev, host)
            self._pending.pop(0)
            self._free_hosts.append(host)
            if self._flags_bad(host[1]):
                self._pending.clear()
                raise self._flags_error(host[1])

    _READY = 16          # word of the pinned buffer the GINE kernel sets to 1 after the 16 flags (sn_gnn_fused_f32)
    _KWORD = 24          # word the HOST writes: the slot count an all-eigenvector forward sized its tensors with (0: max_k given)

    def check_captured(self):
        """After replaying a HIP graph that captured this module's forward: read the captured plan's status words (one host wait) and
        raise what the eager forward would have raised for the batch of the LAST replay."""
        plan = getattr(self, "_captured_plan", None)
        if plan is None:
            return
        flags = plan.flags.cpu().tolist() + [0] * 16
        flags[self._KWORD] = 0 if self.max_k else int(self._captured_K)
        if self._flags_bad(flags):
            raise self._flags_error(flags)

    @classmethod
    def _arrived(cls, ev, host):
        return bool(host[1][cls._READY]) if ev is None else ev.query()

    @classmethod
    def _wait_status(cls, ev, host):
        if ev is not None:
            ev.synchronize()
            return
        view = host[1]
        t_end = time.perf_counter() + 2e-3
        while not view[cls._READY]:                       # normally already there; spin briefly, then block on the device
            if time.perf_counter() > t_end:
                torch.cuda.synchronize()
                if not view[cls._READY]:
                    raise RuntimeError("the fused GINE kernel did not report its status flags")
                break

    def _host_flags(self):
        """A pooled pinned buffer (tensor, numpy view) for one forward's device flags + the kernel's ready word."""
        if self._free_hosts:
            return self._free_hosts.pop()
        t = torch.zeros(32, dtype=torch.int32, pin_memory=True)
        return t, t.numpy()

    def _post_status(self, plan):
        """(event | None, buffer) of this forward's flags.  With the fused GINE stage its last workgroup writes them to the
        pinned buffer itself and then sets the ready word — no copy and no event (a marker packet costs ~6 us of idle
        stream per forward); otherwise an asynchronous copy + event is queued."""
        host = self._flags_host
        self._flags_host = None
        if host is not None:
            return None, host
        host = self._host_flags()
        n = plan.flags.numel()
        host[0][:n].copy_(plan.flags, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return ev, host

    def forward(self, data, return_stages=False):
        if self.training:
            # train mode: BatchNorm with batch statistics over the valid rows and the running-statistics update, layer by
            # layer, and the attention dropout the reference leaves active (self.attn_dropout; the fixtures use 0).
            if torch.is_grad_enabled() and not return_stages and any(p.requires_grad for p in self.parameters()):
                with ops.batched_bn_counters():
                    return self._forward_grad(data)    # differentiable: autograd.Function per layer op (csrc/backward.hip)
            self._prep = None                      # parameters may have changed since the last call
            try:
                with ops.batched_bn_counters():
                    return self._forward(data, return_stages, train=True)
            finally:
                self._prep = None
        if self.use_fused and not return_stages and int(data.num_graphs) > MAX_FUSED_GRAPHS:
            # the work bins of the fused stages are laid out by one workgroup (<= 6144 graphs per plan): larger batches run
            # as consecutive graph ranges — the eval forward never mixes graphs, so the concatenation is the same result
            from . import dist as D
            n = -(-int(data.num_graphs) // MAX_FUSED_GRAPHS)
            return torch.cat([self.forward(D.shard_batch(data, i, n)) for i in range(n)], 0)
        capturing = torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()       # (no GPU: the first op raises 'GPU only')
        if not capturing:
            self.check_last(wait=False)
        self._early = None            # (a report armed by a forward that raised must never be read as this batch's)
        try:
            with _lib_mod.stream_scope():
                y = self._forward(data, return_stages)
        except BaseException:
            stale, self._early = self._early, None
            if stale is not None:     # the plan kernel may still write to the pinned words: hand them back only once it has
                try:
                    stale.wait()
                    stale.release()
                except Exception:
                    pass              # (left to the garbage collector: never recycled while a launch may write to it)
            raise
        early, self._early = self._early, None
        if capturing:
            # A forward being captured into a HIP graph never waits on the host — whatever `strict` says: the flags stay on the device
            # (status words of the captured plan, rewritten by every replay); `check_captured()` reads them after a replay.
            self._captured_plan, self._captured_K = self._last_plan, self._last_K
            self._last_plan = None
            return y
        if early is not None:
            # strict mode, flags from the plan kernel (already in pinned memory: the stage kernels were queued in the meantime)
            fl = early.wait()
            early.release()
            E_ = ops.EarlyReport
            if fl[E_.ERR]:
                raise ValueError("malformed graph batch (unsorted batch vector, graph id / edge endpoint out of range or an edge across graphs)")
            if fl[E_.IDS]:
                raise IndexError(ops.EMBEDDING_INDEX_ERROR)
            if not self.max_k and fl[E_.NMAX] > self._last_K:
                raise ValueError(f"the batch's host-side graph sizes (largest: {self._last_K}) disagree with its batch vector (largest graph: {fl[E_.NMAX]} nodes)")
            if fl[E_.EDGES] or fl[E_.PHI] or fl[E_.RHO]:      # a graph beyond the fused stages' limits: those graphs layer by layer
                y = self._serve_beyond_limits(data)
        elif self.use_fused and not return_stages and self._used_fused:
            ev, host = self._post_status(self._last_plan)
            # all-eigenvector mode: the slot count came from host bookkeeping (data.sizes / ptr): the device's largest graph (status[1])
            # is compared with it when the flags arrive — here, or in check_last() for a queued report
            host[1][self._KWORD] = 0 if self.max_k else int(self._last_K)
            if self.strict:
                self._wait_status(ev, host)
                flags = host[1].tolist()
                self._free_hosts.append(host)
                if self._flags_bad(flags):
                    if flags[0] or flags[5] or (flags[3] & 4) or (flags[self._KWORD] and flags[1] > flags[self._KWORD]):
                        raise self._flags_error(flags)
                    y = self._serve_beyond_limits(data)
            else:
                self._pending.append((ev, host))
        self._last_plan = None
        return y

    def _serve_beyond_limits(self, data):
        """Strict mode, a batch the stage kernels refused (a graph of more than 64 nodes or 192 in-edges, or without nodes — the
        reference takes any graph: Alchemy/sign_net/sign_net.py:96-118, model.py:36-64): ONLY the offending graphs run layer by layer;
        the contiguous runs of graphs around them go through the stage kernels as batches of their own, and the rows are returned in
        graph order.  (Until round 6 the whole batch went layer by layer — and re-packed every weight on the way: 3.7 ms for 127
        molecules + one 70-node graph; now fused(127) + layer path(1).)  The eval forward never mixes graphs, so a graph's rows do
        not depend on which batch it is evaluated in."""
        from . import dist as D
        B = int(data.num_graphs)
        sizes = list(data.sizes) if hasattr(data, "sizes") and len(data.sizes) == B else torch.bincount(data.batch, minlength=B).tolist()
        if data.edge_index.numel():
            ecount = torch.bincount(data.batch[data.edge_index[1]], minlength=B).tolist()
        else:
            ecount = [0] * B
        bad = [n > fused.GNN_MAX_NODES or e > fused.GNN_MAX_EDGES for n, e in zip(sizes, ecount)]
        if not any(bad) or 2 * sum(bad) > B or min(sizes) <= 0:
            # flagged for another reason (a graph without nodes, ...), or mostly oversize graphs: the whole batch layer by layer
            return self._forward(data, False, force_layer=True)
        runs, lo = [], 0
        for g in range(1, B + 1):
            if g == B or bad[g] != bad[lo]:
                runs.append((lo, g, bad[lo]))
                lo = g
        outs = []
        saved_strict, self.strict = self.strict, False       # (the sub-batches' own flags are queued, not waited for)
        try:
            for lo, hi, is_bad in runs:
                sub = D.slice_graphs(data, lo, hi, sizes)
                outs.append(self._forward(sub, False, force_layer=is_bad))
                if not is_bad and self._used_fused:
                    ev, host = self._post_status(self._last_plan)
                    host[1][self._KWORD] = 0 if self.max_k else int(self._last_K)
                    self._pending.append((ev, host))
        finally:
            self.strict = saved_strict
        return torch.cat(outs, 0)

    # ------------------------------------------------------------------ differentiable train-mode forward
    def _forward_grad(self, data):
        """Train-mode forward recorded for torch.autograd: the same layer-at-a-time HIP launches as `_forward(train=True)`,
        each wrapped in a torch.autograd.Function whose backward is its hand-written adjoint (autograd.py).  `.backward()`
        on a loss of the result fills `.grad` of every parameter the reference's forward uses (loss.backward() at
        Alchemy/main_alchemy.py:108, GINESignNetPyG/core/train.py:62).  The attention dropout (self.attn_dropout,
        transformer_module.py:46,55) is an explicit mask drawn once per layer and shared by forward and backward."""
        from . import autograd as AG
        ops.require_cuda(data.edge_index, data.batch, data.eigen_vectors)
        sn, g = self.sign_net, self.gnn
        B = int(data.num_graphs)
        plan = ops.build_plan(data.batch, data.edge_index, B, self.max_k or 0)
        rplan = ops.build_plan(data.batch, data.edge_index.flip(0).contiguous(), B, self.max_k or 0)   # out-edge CSR
        K = int(self.max_k) if self.max_k else plan.check()[1]
        N, nv = plan.N, plan.nvalid
        alchemy_eig = self.variant == "alchemy" and not sn.ignore_eigval
        want_vals = alchemy_eig or self.variant != "alchemy"
        x0, s0 = ops.pack_eig(plan, data.eigen_vectors, data.eigen_values if want_vals else None, K, want_vals)

        def lin_bn(x, lin, norm, nvalid=None, K_=0, relu=True, residual=None):
            if isinstance(norm, _Identity):
                y = AG.linear(x, lin.weight, lin.bias, nvalid, K_, relu=relu)
                return y if residual is None else AG.masked_add(y, residual, nvalid, K_)
            bn = norm.bn if isinstance(norm, MaskedBN) else norm
            if stage and isinstance(bn, nn.BatchNorm1d) and T.supported(lin.weight.shape[1], lin.weight.shape[0]):
                return T.lin_bn(x, lin, bn, nvalid, K_, relu=relu, residual=residual)
            return AG.linear_bn_act(x, lin.weight, lin.bias, bn, nvalid, K_, relu=relu, residual=residual)

        from . import train_stage as T
        d = self.cfg["n_hid"]
        stage = self.train_stages and T.supported(d, d)

        def scalar_form(l0, l1):     # Linear(1, 1, bias=False) -> Linear(1, d): the closed-form 1 -> 1 -> d kernels apply
            return stage and l0.weight.numel() == 1 and l0.bias is None and l1.weight.shape[1] == 1

        if self.variant != "alchemy":      # computed and discarded by the reference (core/sign_net.py:111-112): side effects only
            with torch.no_grad():
                ee2 = sn.eigen_encoder2
                if scalar_form(ee2.layers[0], ee2.layers[1]):
                    T.scalar_mlp_stats(s0, ee2.layers[0], ee2.norms[0].bn, ee2.layers[1], ee2.norms[1].bn, nv, K)
                else:
                    p2 = lin_bn(s0.view(N * K, 1), ee2.layers[0], ee2.norms[0], nv, K)
                    lin_bn(p2, ee2.layers[1], ee2.norms[1], nv, K)
        # ---- phi(x) + phi(-x)
        convs, norms = list(sn.phi.convs), list(sn.phi.norms)
        if stage and len(convs) > 1:
            # layers >= 1 (the [d,d] links): both sign passes stacked group-major [2, N*K, d] — shared weights, separate batch
            # statistics — through the stage kernels: one pass over the rows per link and direction (train_stage.py)
            if scalar_form(convs[0].nn.layers[0], convs[0].nn.layers[1]):
                # GINESignNetPyG's first layer, 1 -> 1 -> d on the scalar aggregate, both signs (aggregate(-x) = -aggregate(x)): closed form
                a0 = AG.gin_aggregate(x0.view(N, -1), convs[0].layer.eps, plan, rplan)
                x = T.scalar_mlp(a0.view(-1), convs[0].nn.layers[0], convs[0].nn.norms[0].bn, convs[0].nn.layers[1], norms[0].bn, nv, K,
                                 G=2, negate_second=True)
            else:
                xs = []
                for sign in (0, 1):
                    a = AG.gin_aggregate(x0.view(N, -1), convs[0].layer.eps, plan, rplan, negate=(sign == 1))
                    h = lin_bn(a.view(N * K, -1), convs[0].nn.layers[0], convs[0].nn.norms[0], nv, K)
                    xs.append(lin_bn(h, convs[0].nn.layers[1], norms[0], nv, K))
                x = torch.cat(xs, 0)
            plan2, rplan2 = ops.doubled_plan(plan), ops.doubled_plan(rplan)
            for conv, norm in zip(convs[1:], norms[1:]):     # aggregate -> link -> link -> + x, adjoints fused the same way
                x = T.gin_layer(x, conv.layer.eps, conv.nn.layers[0], conv.nn.norms[0].bn, conv.nn.layers[1], norm.bn, plan2, rplan2, nv, K, 2)
            x = T.sign_sum(x, nv, K)
        else:
            phis = []
            for sign in (0, 1):
                x, prev = x0, None
                for l, (conv, norm) in enumerate(zip(convs, norms)):
                    a = AG.gin_aggregate(x.view(N, -1), conv.layer.eps, plan, rplan, negate=(sign == 1 and l == 0))
                    h = lin_bn(a.view(N * K, -1), conv.nn.layers[0], conv.nn.norms[0], nv, K)
                    x = lin_bn(h, conv.nn.layers[1], norm, nv, K, residual=prev)
                    prev = x
                phis.append(x)
            x = AG.masked_add(phis[0], phis[1], nv, K)
        # ---- rho
        if alchemy_eig:
            ee = sn.eigen_encoder
            p = lin_bn(s0.view(N * K, 1), ee.layers[0], ee.norms[0], nv, K)
            p = lin_bn(p, ee.layers[1], ee.norms[1], nv, K)
            x = AG.masked_add(x, p, nv, K)
        def lin(x_, m, relu=False):
            return T.linear_module(x_, m, nv, K, relu) if stage else AG.linear(x_, m.weight, m.bias, nv, K, relu=relu)
        static_masks = getattr(self, "_attn_masks", None)       # train_graph.GraphedStep: masks drawn outside the captured step
        for li, tl in enumerate(sn.rho.transformer_layers):
            a, f = tl.slf_attn, tl.pos_ffn
            q, k, v = T.qkv(x, a.w_qs, a.w_ks, a.w_vs, nv, K) if stage else (lin(x, a.w_qs), lin(x, a.w_ks), lin(x, a.w_vs))
            pm = static_masks[li] if static_masks is not None else None
            if pm is not None and tuple(pm.shape) != (N, N_HEAD, K, K):   # masks of another batch shape: never reuse them
                pm = None
            if pm is None:
                pm = ops.attention_dropout_mask(N, K, N_HEAD, self.attn_dropout, x.device)
            o = AG.set_attention(q, k, v, N, K, N_HEAD, nv, pm)
            o = lin(o, a.fc)
            y = AG.masked_layernorm(o, x, a.norm.ln.weight, a.norm.ln.bias, LN_EPS, nv, K)
            z = lin(y, f.w_1, relu=True)
            z = lin(z, f.w_2)
            x = AG.masked_layernorm(z, y, f.norm.ln.weight, f.norm.ln.bias, LN_EPS, nv, K)
        s = AG.slot_sum(x, N, K, nv)
        pe = lin_bn(s, sn.rho.out[0], sn.rho.out[1], relu=False)
        # ---- GINE network
        xin = data.x.squeeze() if data.x.dim() > 1 and data.x.shape[-1] == 1 else data.x
        if isinstance(g.input_encoder, DiscreteEncoder):
            h = AG.embedding_sum(xin, [e.weight for e in g.input_encoder.embeddings], plan.status[5:6])
        else:
            h = lin_bn(xin.contiguous(), g.input_encoder.layers[0], g.input_encoder.norms[0])
        h = AG.linear(torch.cat([h, pe], dim=-1), g.linear.weight, g.linear.bias)
        staged = [stage and isinstance(conv.nn.norms[0], nn.BatchNorm1d) and isinstance(norm, nn.BatchNorm1d)
                  for conv, norm in zip(g.convs, g.norms)]
        # every layer embeds the same edge_attr with its own tables: one [L, E, C] block, one adjoint launch pair for all layers
        e_all = None
        if all(staged) and all(isinstance(enc, DiscreteEncoder) for enc in g.edge_encoders) and 1 < len(g.edge_encoders) <= 16:
            e_all = AG.embedding_sum_layers(data.edge_attr, [[t.weight for t in enc.embeddings] for enc in g.edge_encoders], plan.status[5:6])
        for li, (enc, conv, norm) in enumerate(zip(g.edge_encoders, g.convs, g.norms)):
            if e_all is not None:
                h = T.gine_layer(h, e_all, conv.layer.eps, conv.nn.layers[0], conv.nn.norms[0], conv.nn.layers[1], norm, plan, rplan, layer=li)
                continue
            if isinstance(enc, DiscreteEncoder):
                e = AG.embedding_sum(data.edge_attr, [t.weight for t in enc.embeddings], plan.status[5:6])
            else:
                e = lin_bn(data.edge_attr.contiguous(), enc.layers[0], enc.norms[0])
            if staged[li]:
                h = T.gine_layer(h, e, conv.layer.eps, conv.nn.layers[0], conv.nn.norms[0], conv.nn.layers[1], norm, plan, rplan)
            else:
                u = AG.gine_aggregate(h, e, conv.layer.eps, plan, rplan)
                u = lin_bn(u, conv.nn.layers[0], conv.nn.norms[0])
                h = lin_bn(u, conv.nn.layers[1], norm, residual=h)
        pooled = AG.segment_pool(h, plan, g.pooling)
        oe = g.output_encoder
        y = lin_bn(pooled, oe.layers[0], oe.norms[0])
        y = AG.linear(y, oe.layers[1].weight, oe.layers[1].bias)
        discrete = isinstance(g.input_encoder, DiscreteEncoder) or any(isinstance(e, DiscreteEncoder) for e in g.edge_encoders)
        self._train_status = plan.status if discrete else None
        if not getattr(self, "_defer_status", False):
            self.check_train()                   # one sync per training step; node OR edge tables (nn.Embedding raises for either)
        return y

    def check_train(self):
        """The embedding-index flag of the last differentiable forward (read here, not inside the step, when `_defer_status` is set:
        a captured step — train_graph.GraphedStep — must not synchronise)."""
        st, self._train_status = getattr(self, "_train_status", None), None
        if st is not None and int(st[5]):
            raise IndexError(ops.EMBEDDING_INDEX_ERROR)

    # ------------------------------------------------------------------ forward
    @staticmethod
    def _arm_early(gplan, data):
        """An EarlyReport for this batch: the plan kernel also checks what the fused GINE stage would flag at the END of the forward —
        a graph with more than 192 in-edges, discrete feature ids outside their embedding tables (nn.Embedding's IndexError)."""
        xi = ei = None
        nv = ev = 0
        if gplan.node_discrete and data.x.dtype == torch.int64 and data.x.is_cuda:
            xi, nv = data.x.contiguous(), int(gplan.params.node_vocab)
        if gplan.edge_discrete and int(gplan.params.n_layers) > 0 and data.edge_attr.dtype == torch.int64 and data.edge_attr.is_cuda \
                and data.edge_attr.numel() > 0:
            ei, ev = data.edge_attr.contiguous(), int(gplan.params.edge_vocab)
        return ops.EarlyReport().arm(xi, nv, ei, ev, fused.GNN_MAX_EDGES)

    def _forward_overlapped(self, data, P, B):
        """The fused eval forward as a three-stage pipeline over the module's two side streams and the caller's stream:
            side A: batch plan + phi      -> event ->      side B: rho      -> event ->      caller's stream: GINE stage.
        The output is ordered on the caller's stream like any other op's; the NEXT calls' earlier stages — queued while this call's
        later ones are still running — share the GPU with them (the GINE stage is one workgroup per graph: half the chip at 128
        graphs; rho's attention phases and phi's gathers leave the matrix pipe idle).  Measured on the headline batch, ms per forward:
        0.27 on one stream, 0.25 with only the GINE stage behind an event, 0.23 with rho + GINE behind it, 0.197 with the three stages.
        Bit-identical outputs.
        Opt-in (`overlap_front = True`, async status mode): side A does not wait for the caller's stream, so the batch's
        tensors must be complete on the device when forward is called (the resident batches of a serving loop; NOT a batch whose
        host-to-device copy was just queued on the current stream — `overlap_inputs_ready = False` covers that: correct, side A then
        waits for the caller's stream and consecutive forwards no longer overlap)."""
        d = self.cfg["n_hid"]
        dev = data.batch.device
        cur = torch.cuda.current_stream(dev)
        if self._side_streams is None:
            self._side_streams = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev))
        side_a, side_b = self._side_streams
        if not self.overlap_inputs_ready:
            side_a.wait_stream(cur)
        want_vals = "eig" in P or "eig2" in P
        with torch.cuda.stream(side_a), _lib_mod.stream_scope():
            plan = ops.build_plan(data.batch, data.edge_index, B, self.max_k or 0, bins=True)
            # all-eigenvector mode (max_k None): K = the largest graph, read back as the reference's to_dense_EVD does — the host waits
            # for side A only (the previous forward's rho / GINE keep running)
            K_host = None if self.max_k else host_max_nodes(data)
            K = int(self.max_k) if self.max_k else (int(K_host) if K_host is not None else plan.check()[1])
            x = P["phi_fused"].run(plan, data.eigen_vectors, K, zero_invalid=False).view(plan.N * K, d)
            ev_a = torch.cuda.Event()
            ev_a.record(side_a)
        side_b.wait_event(ev_a)
        with torch.cuda.stream(side_b), _lib_mod.stream_scope():
            s = P["rho_fused"].run(plan, x, data.eigen_values if want_vals else None, K)
            ev_b = torch.cuda.Event()
            ev_b.record(side_b)
        cur.wait_event(ev_b)
        # allocator bookkeeping: a tensor is recorded on every stream that reads it besides the one it was allocated on, so that
        # dropping it (the caller its batch, this function its temporaries) never hands the memory out while a stage still reads it
        for t in (data.batch, data.edge_index, data.eigen_vectors):
            t.record_stream(side_a)
        if want_vals:
            data.eigen_values.record_stream(side_b)
        for t in (plan.graph_ptr, plan.evoff):         # (the plan's arrays are views of one arena)
            t.record_stream(side_b)
            t.record_stream(cur)
        x.record_stream(side_b)
        s.record_stream(cur)
        self._last_plan, self._used_fused, self._last_K = plan, True, K
        with _lib_mod.stream_scope():
            self._flags_host = self._host_flags() if (plan.bins is not None and not _NO_KERNEL_FLAGS) else None
            if self._flags_host is not None:
                self._flags_host[1][:] = 0
            return P["gnn_fused"].run(plan, data.x, data.edge_attr, s, None if self._flags_host is None else self._flags_host[0])

    def _forward(self, data, return_stages=False, train=False, force_layer=False):
        ops.require_cuda(data.edge_index, data.batch, data.eigen_vectors)
        if self._prep is None:
            self._prep = self._prepare(train)
        P = self._prep
        B = int(data.num_graphs)
        # (force_layer: the layer-at-a-time launches from the SAME prepared parameters — _prepare packs them beside the stage kernels')
        use_phi_fused = P["phi_fused"] is not None and not force_layer
        use_rho_fused = P["rho_fused"] is not None and not force_layer
        use_gnn_fused = P["gnn_fused"] is not None and not force_layer
        if (self.overlap_front and use_phi_fused and use_rho_fused and use_gnn_fused and not return_stages and not train
                and not self.strict):
            return self._forward_overlapped(data, P, B)
        all_fused = use_phi_fused and use_rho_fused and use_gnn_fused and not return_stages and not train
        early = None
        N_, E_n = int(data.batch.numel()), (int(data.edge_index.shape[1]) if data.edge_index.numel() else 0)
        # (a forward being captured into a HIP graph must not wait on the host: no early report, the flags stay on the device)
        capturing = torch.cuda.is_current_stream_capturing()
        if all_fused and self.strict and not capturing and ops.early_supported(N_, E_n, B):
            early = self._arm_early(P["gnn_fused"], data)
        K_host = None if self.max_k else host_max_nodes(data)
        if not self.max_k and K_host is None and early is None and not return_stages and not capturing and ops.early_supported(N_, E_n, B):
            early = ops.EarlyReport().arm()          # only for the largest graph: a poll of pinned memory instead of a device read-back
        plan = ops.build_plan(data.batch, data.edge_index, B, self.max_k or 0, bins=use_phi_fused or use_rho_fused, early=early)
        self._last_plan, self._used_fused = plan, (use_phi_fused or use_rho_fused or use_gnn_fused)
        if self.max_k:
            K = int(self.max_k)
            plan.check() if return_stages else None
        elif K_host is not None:
            K = int(K_host)              # N_max from the batch object's host-side sizes: no device read-back
            plan.check() if return_stages else None
        elif early is not None:
            K = early.wait_nmax()        # N_max as the plan kernel reports it to pinned memory (the reference's to_dense_EVD reads it back too)
        else:
            K = plan.check()[1]          # N_max: one host sync, as the reference's to_dense_EVD does
        if early is not None and not (all_fused and self.strict):
            early.wait()
            early.release()
            early = None
        self._early, self._last_K = early, K
        N, d = plan.N, self.cfg["n_hid"]
        nv = plan.nvalid
        want_vals = "eig" in P or "eig2" in P
        x0 = s0 = None
        if return_stages or not (use_phi_fused and use_rho_fused):     # only the layer path needs the dense [N,K] blocks
            x0, s0 = ops.pack_eig(plan, data.eigen_vectors, data.eigen_values if want_vals else None, K, want_vals)
        stages = {}

        if "eig2" in P:      # train mode, GINESignNetPyG: side effects only (see _prepare)
            E2 = P["eig2"]
            p2 = _lin_bn(s0.view(N * K, 1), E2["l0"], E2["bn0"], True, nv, K, relu=True)
            _lin_bn(p2, E2["l1"], E2["bn1"], True, nv, K, relu=True)
        # ---- phi(x) + phi(-x)      (GNN3d.forward, sign_net.py:28-44)
        phis = None
        if use_phi_fused and not return_stages:
            x = P["phi_fused"].run(plan, data.eigen_vectors, K, zero_invalid=not use_rho_fused).view(N * K, d)
        else:
            phis = []
            for sign in (0, 1):
                x, prev = x0, None
                for l, L in enumerate(P["phi"]):
                    a = ops.gin_aggregate(x.view(N, -1), plan, L["eps"], negate=(sign == 1 and l == 0))
                    h = _lin_bn(a.view(N * K, -1), L["l0"], L["bn0"], train, nv, K, relu=True)
                    x = _lin_bn(h, L["l1"], L["bn"], train, nv, K, relu=True, residual=prev)
                    prev = x
                phis.append(x)
            x = ops.masked_affine(phis[0], nv, K, residual=phis[1])          # phi(x) + phi(-x)
        if return_stages:
            stages.update(phi_plus=phis[0].view(N, K, d), phi_minus=phis[1].view(N, K, d), phi=x.view(N, K, d))
            if P["phi_fused"] is not None:      # cross-check target for the fused kernel
                stages["phi_fused"] = P["phi_fused"].run(plan, data.eigen_vectors, K)
                stages["bins_meta"] = plan.bins.meta
        # ---- rho                    (SetTransformer.forward, sign_net.py:60-72)
        x_phi = x
        if use_rho_fused and not return_stages:
            s = P["rho_fused"].run(plan, x_phi, data.eigen_values if want_vals else None, K)
        else:
            if "eig" in P:
                E_ = P["eig"]
                p = _lin_bn(s0.view(N * K, 1), E_["l0"], E_["bn0"], train, nv, K, relu=True)
                p = _lin_bn(p, E_["l1"], E_["bn1"], train, nv, K, relu=True)
                x = ops.masked_affine(x, nv, K, residual=p)
            given = getattr(self, "_attn_masks", None) if train else None      # explicit masks: the reference's own draws (tests), GraphedStep
            for li, L in enumerate(P["rho"]):
                q = ops.masked_linear(x, L["q"], nv, K)
                k = ops.masked_linear(x, L["k"], nv, K)
                v = ops.masked_linear(x, L["v"], nv, K)
                pm = None
                if train:
                    pm = given[li] if given is not None else ops.attention_dropout_mask(N, K, N_HEAD, self.attn_dropout, x.device)
                o = ops.set_attention(q, k, v, N, K, N_HEAD, nv, pm)
                o = ops.masked_linear(o, L["fc"], nv, K)
                y = ops.masked_layernorm(o, x, L["ln1"][0], L["ln1"][1], LN_EPS, nv, K)
                z = ops.masked_linear(y, L["w1"], nv, K, relu=True)
                z = ops.masked_linear(z, L["w2"], nv, K)
                x = ops.masked_layernorm(z, y, L["ln2"][0], L["ln2"][1], LN_EPS, nv, K)
            s = ops.slot_sum(x, N, K)
        if use_gnn_fused and not return_stages:
            self._flags_host = self._host_flags() if (plan.bins is not None and not _NO_KERNEL_FLAGS and self._early is None) else None
            if self._flags_host is not None:
                self._flags_host[1][:] = 0        # host-side; the kernel's last workgroup overwrites it, ready word last
            return P["gnn_fused"].run(plan, data.x, data.edge_attr, s, None if self._flags_host is None else self._flags_host[0])
        pe = _lin_bn(s, P["rho_out"]["l"], P["rho_out"]["bn"], train, relu=False)
        if return_stages:
            stages["pos"] = pe
            stages["rho_sum"] = s
            if use_rho_fused:
                stages["rho_sum_fused"] = P["rho_fused"].run(plan, x_phi, data.eigen_values if want_vals else None, K)
        # ---- GINE network           (GNN.forward, model.py:36-64)
        xin = data.x.squeeze() if data.x.dim() > 1 and data.x.shape[-1] == 1 else data.x
        if "in_tabs" in P:
            h = ops.embedding_sum(xin, P["in_tabs"], plan.status[5:6])
        else:
            h = _lin_bn(xin.contiguous(), P["in_mlp"]["l"], P["in_mlp"]["bn"], train, relu=True)
        h = ops.masked_linear(torch.cat([h, pe], dim=-1), P["lin"])
        for l, L in enumerate(P["gine"]):
            if "tabs" in L:
                e = ops.embedding_sum(data.edge_attr, L["tabs"], plan.status[5:6])
            else:
                e = _lin_bn(data.edge_attr.contiguous(), L["emlp"]["l"], L["emlp"]["bn"], train, relu=True)
            u = ops.gine_aggregate(h, e, plan, L["eps"])
            u = _lin_bn(u, L["l0"], L["bn0"], train, relu=True)
            h = _lin_bn(u, L["l1"], L["bn"], train, relu=True, residual=h)
            if return_stages:
                stages[f"gine{l}"] = h
        pooled = ops.segment_pool(h, plan, self.gnn.pooling)
        y = _lin_bn(pooled, P["head"]["l0"], P["head"]["bn0"], train, relu=True)
        y = ops.masked_linear(y, P["head"]["l1"])
        if ("in_tabs" in P or any("tabs" in d for d in P["gine"])) and int(plan.status[5]):      # layer path only (train mode / return_stages / fallback): one host sync
            raise IndexError(ops.EMBEDDING_INDEX_ERROR)
        if return_stages:
            stages["y"] = y
            if use_gnn_fused:
                stages["y_gnn_fused"] = P["gnn_fused"].run(plan, data.x, data.edge_attr, s)
            return y, stages
        return y
