"""CPU: host-side logic of the product package (no kernels run here)."""
import pytest
import torch

import golden_util as G


@pytest.mark.parametrize("name", G.PYG_CASES)
def test_state_dict_layout_matches_reference(name):
    """Same keys and shapes as the reference modules' state_dict, so reference weights load."""
    from signnet_basisnet_amd.pyg import SignNetGNN
    fx = G.load(name)
    c = [None if v < 0 else int(v) for v in fx.meta["ctor"]]
    m = SignNetGNN(*c, variant=str(fx.meta["variant"]))
    ref = {str(k): tuple(int(s) for s in str(shp).split(",") if s) for k, shp in zip(fx.meta["sd_keys"], fx.meta["sd_shapes"])}
    mine = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert mine == ref
    m.load_state_dict(G.full_state_dict(fx))        # strict


def test_reference_constructor_signatures():
    from signnet_basisnet_amd.pyg import SignNetGNN
    a = SignNetGNN(6, 4, 108, 12, 8, 16, nl_rho=8)                  # main_alchemy.py:35 (nl_rho is ignored -> 4)
    assert len(a.sign_net.rho.transformer_layers) == 4 and len(a.sign_net.phi.convs) == 8 and len(a.gnn.convs) == 16
    g = SignNetGNN(None, None, 128, 1, 4, 6, variant="gine")         # train/zinc.py:33-37
    assert len(g.sign_net.rho.transformer_layers) == 1
    with pytest.raises(ValueError):
        SignNetGNN(6, 4, 108, 12, 8, 16, gnn_type="GATConv")
    with pytest.raises(ValueError):
        SignNetGNN(None, None, 30, 1, 2, 2, variant="gine")         # 30 not divisible by the 4 heads


def test_no_cpu_fallback():
    from signnet_basisnet_amd import synth
    from signnet_basisnet_amd.pyg import SignNetGNN
    m = SignNetGNN(None, None, 16, 1, 2, 2, variant="gine").eval()
    with pytest.raises(RuntimeError, match="GPU only"):
        m(synth.make_batch(3, seed=0))


def test_synth_batches_are_deterministic_and_well_formed():
    from signnet_basisnet_amd import synth
    a, b = synth.make_batch(8, seed=5), synth.make_batch(8, seed=5)
    assert torch.equal(a.edge_index, b.edge_index) and torch.equal(a.eigen_vectors, b.eigen_vectors)
    n = torch.tensor(a.sizes)
    assert a.batch.numel() == int(n.sum()) and a.eigen_vectors.numel() == int((n * n).sum())
    assert (a.batch[1:] >= a.batch[:-1]).all()
    src, dst = a.edge_index
    assert (a.batch[src] == a.batch[dst]).all() and (src != dst).all()
    key = src * a.batch.numel() + dst
    assert key.unique().numel() == key.numel()                       # no duplicate edges
    rev = dst * a.batch.numel() + src
    assert set(key.tolist()) == set(rev.tolist())                    # symmetric
    # eigenvectors really are eigenvectors of the sym-normalised Laplacian of graph 0
    n0 = a.sizes[0]
    V = a.eigen_vectors[:n0 * n0].view(n0, n0)
    torch.testing.assert_close(V.T @ V, torch.eye(n0), rtol=0, atol=1e-4)
    pe = synth.dgl_pos_enc(a, 8)
    assert pe.shape == (a.batch.numel(), 8)


def test_shard_ranges_cover_everything():
    from signnet_basisnet_amd.dist import shard_range
    for B in (1, 7, 128, 1024):
        for W in (1, 2, 3, 8):
            rs = [shard_range(B, r, W) for r in range(W)]
            assert rs[0][0] == 0 and rs[-1][1] == B
            assert all(rs[i][1] == rs[i + 1][0] for i in range(W - 1))
            assert max(h - l for l, h in rs) - min(h - l for l, h in rs) <= 1


def test_work_balanced_shard_ranges():
    """ranks get contiguous graph ranges of (nearly) equal phi work sum n_b * min(n_b, k), not equal graph count."""
    import numpy as np
    from signnet_basisnet_amd import dist, synth
    rng = np.random.default_rng(0)
    sizes = rng.integers(9, 38, size=1024).tolist()
    for k in (None, 16):
        work = [n * (min(n, k) if k else n) for n in sizes]
        for W in (2, 4, 8):
            rs = [dist.shard_range(len(sizes), r, W, work) for r in range(W)]
            assert rs[0][0] == 0 and rs[-1][1] == len(sizes) and all(rs[i][1] == rs[i + 1][0] for i in range(W - 1))
            loads = [sum(work[l:h]) for l, h in rs]
            assert max(loads) - min(loads) <= 2 * max(work), (k, W, loads)
            by_count = [sum(work[l:h]) for l, h in (dist.shard_range(len(sizes), r, W) for r in range(W))]
            assert max(loads) <= max(by_count) + max(work)
    # one heavy graph must not starve a rank: no empty range while there is a graph for everyone
    assert [dist.shard_range(4, r, 4, [100, 1, 1, 1]) for r in range(4)] == [(0, 1), (1, 2), (2, 3), (3, 4)]
    assert all(h > l for l, h in (dist.shard_range(6, r, 3, [1, 1, 1, 1, 1, 100]) for r in range(3)))
    # degenerate inputs
    assert dist.shard_range(3, 0, 8, [5, 1, 1])[0] == 0
    assert [dist.shard_range(3, r, 8, [5, 1, 1]) for r in range(8)][-1][1] == 3
    data = synth.make_batch(12, seed=3)
    parts = [dist.shard_batch(data, r, 3, balance="rows", max_k=8) for r in range(3)]
    assert sum(p.num_graphs for p in parts) == 12 and sum(p.num_nodes for p in parts) == data.num_nodes


def test_dropin_import_paths(monkeypatch):
    """The dotted names the reference's entry scripts import resolve to the HIP modules."""
    import importlib
    import os
    import sys
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "signnet_basisnet_amd", "dropin")
    for sub, mod, names in (("alchemy", "sign_net.sign_net", ["SignNetGNN"]), ("alchemy", "sign_net.transform", ["EVDTransform"]),
                            ("gine_pyg", "core.sign_net", ["SignNetGNN"]), ("gine_pyg", "core.transform", ["EVDTransform"]),
                            ("graphprediction", "layers.deepsigns", ["GINDeepSigns", "MaskedGINDeepSigns"]),
                            ("graphprediction", "nets.ZINC_graph_regression.sign_inv_net", ["get_sign_inv_net"]),
                            ("graphprediction", "nets.ZINC_graph_regression.gin_net", ["GINNet"]),
                            ("graphprediction", "nets.ZINC_graph_regression.gatedgcn_net", ["GatedGCNNet"]),
                            ("graphprediction", "nets.ZINC_graph_regression.pna_net", ["PNANet"]),
                            ("graphprediction", "nets.ZINC_graph_regression.transformer_net", ["TransformerNet"]),
                            ("graphprediction", "nets.ZINC_graph_regression.gat_net", ["GATNet"]),
                            ("learningfilters", "signbasisnet", ["SignPlus", "IGNBasisInv"]), ("learningfilters", "ign", ["IGN2to1"])):
        for m in list(sys.modules):
            if m.split(".")[0] in ("sign_net", "core", "layers", "nets", "signbasisnet", "ign"):
                del sys.modules[m]
        monkeypatch.syspath_prepend(os.path.join(root, sub))
        module = importlib.import_module(mod)
        for n in names:
            assert hasattr(module, n)


def test_evd_transform_wire_format():
    import types
    from signnet_basisnet_amd import synth
    from signnet_basisnet_amd.transform import EVDTransform
    b = synth.make_batch(1, seed=2, sizes=[11])
    d = types.SimpleNamespace(edge_index=b.edge_index, num_nodes=11, x=b.x)
    d = EVDTransform("sym")(d)
    assert d.eigen_values.shape == (11,) and d.eigen_vectors.shape == (121,)
    torch.testing.assert_close(d.eigen_values, b.eigen_values, rtol=1e-5, atol=1e-5)
    V = d.eigen_vectors.view(11, 11)
    L = V @ torch.diag(d.eigen_values) @ V.T
    assert abs(float(L.diagonal().mean()) - 1.0) < 1e-4             # sym-normalised Laplacian has unit diagonal


def test_widening_modules_import_without_a_gpu_and_keep_the_reference_surface():
    """The f-row modules (training, eigendecomposition, DGL base nets, serving) import on a CPU-only box; constructors mirror the
    reference's (net_params dictionaries / positional arguments) and the state_dict key sets match the fixtures'."""
    import torch
    import golden_util as G
    from signnet_basisnet_amd import autograd, dgl_nets, optim, serving, transform  # noqa: F401
    for name, cls, kind, agg in (("dgl_ginnet_k6", dgl_nets.GINNet, "gin", "add"),
                                 ("dgl_gatedgcn_concat_k6", dgl_nets.GatedGCNNet, "masked_gin", "concat")):
        fx = G.load(name)
        hidden, L, k = (int(v) for v in fx.meta["params"])
        net = cls(dict(num_atom_type=28, num_bond_type=4, hidden_dim=hidden, out_dim=hidden, in_feat_dropout=0.0, dropout=0.0, L=L,
                       readout="mean", batch_norm=True, residual=True, edge_feat=True, device="cpu", pe_init="lap_pe",
                       lap_method="sign_inv", lap_lspe=False, use_lapeig_loss=False, lambda_loss=1, alpha_loss=1e-4, pos_enc_dim=k,
                       sign_inv_net=kind, sign_inv_layers=3, sign_inv_activation="relu", pe_aggregate=agg, phi_out_dim=4))
        assert sorted(net.state_dict().keys()) == sorted(fx.sd.keys())
        net.load_state_dict(fx.sd)
    with pytest.raises(NotImplementedError):
        dgl_nets.GINNet(dict(num_atom_type=28, num_bond_type=4, hidden_dim=8, out_dim=8, in_feat_dropout=0.0, dropout=0.0, L=2,
                             readout="mean", batch_norm=True, residual=True, edge_feat=True, device="cpu", pe_init="rand_walk",
                             lap_method="sign_inv", lap_lspe=True, use_lapeig_loss=False, lambda_loss=1, alpha_loss=1e-4, pos_enc_dim=4,
                             sign_inv_net="gin", sign_inv_layers=2, sign_inv_activation="relu", pe_aggregate="add", phi_out_dim=4))
    t = transform.EVDTransform("sym")          # the per-sample host transform keeps the reference's call contract
    import types
    d = t(types.SimpleNamespace(edge_index=torch.tensor([[0, 1, 1, 2], [1, 0, 2, 1]]), num_nodes=3, x=torch.zeros(3, 1)))
    assert d.eigen_values.shape == (3,) and d.eigen_vectors.shape == (9,)
    with pytest.raises(RuntimeError, match="GPU only|No HIP|cuda"):
        transform.BatchEVDTransform("sym")(types.SimpleNamespace(edge_index=torch.tensor([[0, 1], [1, 0]]), batch=torch.zeros(2, dtype=torch.long),
                                                                 num_graphs=1, sizes=[2]))


def test_packed_weight_caches_are_dropped_when_a_parent_loads_a_state_dict():
    """nn.Module.load_state_dict on a PARENT recurses through _load_from_state_dict and never calls a child's
    load_state_dict override: every module that caches packed eval-mode weights must notice it anyway."""
    import torch.nn as nn
    from signnet_basisnet_amd import basisnet, dgl_deepsigns, dgl_nets
    from signnet_basisnet_amd.pyg import SignNetGNN
    stale = object()
    m = SignNetGNN(None, None, 16, 1, 2, 2, variant="gine")
    wrap = nn.ModuleDict({"model": m})
    m._prep = stale
    wrap.load_state_dict(wrap.state_dict())
    assert m._prep is None
    inv = basisnet.IGNBasisInv([1, 2], 1, hidden_channels=8)
    for e in inv.encs:
        e._prep = stale
    inv.load_state_dict(inv.state_dict())
    assert all(e._prep is None for e in inv.encs)
    ds = dgl_deepsigns.GINDeepSigns(1, 8, 4, 3, 6, use_bn=True, dropout=0.0)
    holder = nn.Sequential(ds)
    ds._prep = stale
    holder.load_state_dict(holder.state_dict())
    assert ds._prep is None
    params = dict(num_atom_type=28, num_bond_type=4, hidden_dim=16, out_dim=16, in_feat_dropout=0.0, dropout=0.0, L=2, readout="mean",
                  batch_norm=True, residual=True, edge_feat=True, device="cpu", pe_init="lap_pe", lap_method="sign_inv", lap_lspe=False,
                  use_lapeig_loss=False, lambda_loss=1, alpha_loss=1e-4, pos_enc_dim=6, sign_inv_net="gin", sign_inv_layers=3,
                  sign_inv_activation="relu", pe_aggregate="add", phi_out_dim=4)
    net = dgl_nets.GINNet(params)
    net.sign_inv_net._prep = stale
    net._cache = {"x": stale}
    outer = nn.ModuleDict({"net": net})
    outer.load_state_dict(outer.state_dict())
    assert net.sign_inv_net._prep is None and net._cache == {}


def test_unimplemented_dropout_is_refused_not_ignored():
    from signnet_basisnet_amd import dgl_deepsigns
    with pytest.raises(NotImplementedError, match="dropout"):
        dgl_deepsigns.GINDeepSigns(1, 8, 4, 3, 6, use_bn=True)                 # the reference's default dropout=0.5
    with pytest.raises(NotImplementedError, match="dropout"):
        dgl_deepsigns.MaskedGINDeepSigns(1, 8, 4, 3, 6, use_bn=True, dropout=0.1)


def test_learning_filters_factories_have_the_reference_state_dict_keys():
    """gen_model (training.py:152-181) builds modules whose state_dict keys are exactly the reference's, for
    every configuration of the fixture (DS / MLP / Linear / Transformer bases; DS / MLP / Transformer sign-invariant nets; IGN and
    IGNShared basis-invariant nets) — checked by a strict load of the reference's own tensors."""
    import types

    import golden_util as G
    from signnet_basisnet_amd import learning_filters as LF
    from signnet_basisnet_amd.dropin.learningfilters import models as dropin_models
    assert dropin_models.Transformer is LF.Transformer and dropin_models.MLP is LF.MLP
    fx = G.load_filters()
    for name, c in fx.cases.items():
        args = LF.FilterArgs(**c["args"])
        eig = types.SimpleNamespace(N=36, pe_dim=32 if args.lap_method != "none" else 72, uniq_mults=c["cfg"]["mults"], num_eigenspaces=0)
        model = LF.gen_model(args, eig, "cpu")
        model.load_state_dict(c["sd"], strict=True)
        assert sum(p.numel() for p in model.parameters()) == sum(v.numel() for k, v in c["grad"].items()), name
    with pytest.raises(NotImplementedError):
        LF.gen_model(LF.FilterArgs(net="BernNet"), types.SimpleNamespace(N=36, pe_dim=0, uniq_mults=[], num_eigenspaces=0), "cpu")
    with pytest.raises(NotImplementedError):
        LF.Transformer(4, dropout=0.1)
    with pytest.raises(RuntimeError):            # no CPU path: the forward needs the HIP library and device tensors
        LF.MLP(3)(torch.zeros(5, 3))


def test_slice_graphs_is_the_oracle_of_the_sub_batch():
    """dist.slice_graphs (what the default mode uses to serve the graphs around an oversize one through the stage kernels, and shard_batch to
    cut a global batch): the graphs [lo, hi) of a collated batch as a batch of their own — the CPU oracle of the slice equals the rows of the
    oracle of the whole batch (the eval forward never mixes graphs), for a middle range, the first graph and the last."""
    import torch
    from oracle import pyg_signnet as O
    from signnet_basisnet_amd import dist as D
    from signnet_basisnet_amd import synth
    from signnet_basisnet_amd.pyg import SignNetGNN
    torch.manual_seed(3)
    ctor = (None, None, 16, 1, 2, 2)
    m = SignNetGNN(*ctor, variant="gine", max_k=8)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    cfg = O.make_cfg("gine", *ctor)
    data = synth.make_batch(7, seed=11)
    with torch.no_grad():
        whole = O.signnet_gnn(sd, cfg, data, training=False, max_k=8)
        for lo, hi in ((2, 5), (0, 1), (6, 7), (0, 7)):
            sub = D.slice_graphs(data, lo, hi)
            assert sub.num_graphs == hi - lo and sub.sizes == data.sizes[lo:hi] and int(sub.batch.max()) == hi - lo - 1
            assert sub.eigen_vectors.numel() == sum(n * n for n in sub.sizes) and int(sub.edge_index.max()) < sub.num_nodes
            part = O.signnet_gnn(sd, cfg, sub, training=False, max_k=8)
            assert (part - whole[lo:hi]).abs().max().item() <= 1e-5 * whole.abs().max().item()
    # shard_batch is slice_graphs over shard_range
    s1 = D.shard_batch(data, 1, 3)
    lo, hi = D.shard_range(7, 1, 3)
    s2 = D.slice_graphs(data, lo, hi)
    assert torch.equal(s1.edge_index, s2.edge_index) and torch.equal(s1.x, s2.x) and s1.sizes == s2.sizes


def test_deferred_reduction_queue_is_armed_once_per_backward_pass():
    """train_stage's deferred dW reductions are flushed by ONE end-of-backward callback per backward pass (autograd's graph-task id),
    re-armed for the next pass even if an earlier pass never ran its callback; outside a backward pass a job is reduced at once.  Host
    logic only: the flush is replaced by a recorder."""
    import torch
    from signnet_basisnet_amd import train_stage as T
    calls = []
    real = T.flush_deferred

    def recorder():
        calls.append(len(T._Deferred.jobs))
        T._Deferred.armed = False
        T._Deferred.jobs, T._Deferred.outs = [], set()

    T.flush_deferred = recorder
    try:
        outs = [torch.zeros(4) for _ in range(3)]

        class F(torch.autograd.Function):
            @staticmethod
            def forward(ctx, x):
                return x * 1.0

            @staticmethod
            def backward(ctx, g):
                for o in outs:
                    T._defer_reduce(torch.zeros(8), 0, 2, 4, 4, o)
                return g

        x = torch.ones(2, requires_grad=True)
        F.apply(x).sum().backward()
        assert calls == [3], calls                      # one flush, at the end of the pass, with the three jobs
        T._Deferred.armed = True                         # (a pass that raised before its callback ran leaves this behind)
        F.apply(x).sum().backward()
        assert calls == [3, 3], calls                    # the next pass still flushes its own jobs at its end
        T._defer_reduce(torch.zeros(8), 0, 2, 4, 4, outs[0])
        assert calls == [3, 3, 1], calls                 # no backward pass running: reduced at once
        # two adds into one gradient never share a launch: the second one flushes the first
        F2_calls = len(calls)

        class G2(torch.autograd.Function):
            @staticmethod
            def forward(ctx, x):
                return x * 1.0

            @staticmethod
            def backward(ctx, g):
                T._defer_reduce(torch.zeros(8), 0, 2, 4, 4, outs[0])
                T._defer_reduce(torch.zeros(8), 0, 2, 4, 4, outs[0])
                return g

        G2.apply(x).sum().backward()
        assert calls[F2_calls:] == [1, 1, 0], calls       # (the mid-pass flush re-arms: the pass's second callback finds nothing queued)
    finally:
        T.flush_deferred = real
        T._Deferred.jobs, T._Deferred.outs, T._Deferred.armed = [], set(), False
