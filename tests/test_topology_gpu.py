"""-m gpu: graph TOPOLOGIES outside the molecule generator of `synth.make_batch` — everything else in tests/ runs on symmetric,
degree <= 4, sorted edge lists.  The reference accepts any `edge_index` (PyG's scatter-add over whatever the collate produced:
GINESignNetPyG/core/model_utils/pyg_gnn_wrapper.py:19-28, Alchemy/sign_net/masked_layers.py:118-131), and the kernels have paths
only such inputs reach: more than PHI_NBR = 8 in-neighbours of a row (CSR read from global memory), more than four in-edges per
node in the GINE stage, self loops, duplicate edges, directed (asymmetric) edges, isolated nodes, an edge list in random order
(the aggregation sums in edge-id order), graphs at the 64-node / 192-edge limits of the stage kernels and just beyond them (layer
path).  Forward in eval mode (fused stages and layer path) and the train-mode forward + parameter gradients against the CPU oracle
(fp32, attributed to float64 as everywhere else)."""
import types

import numpy as np
import pytest
import torch

import parity_util as PU

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _sym(e):
    e = np.asarray(e, dtype=np.int64).reshape(-1, 2)
    return np.concatenate([e, e[:, ::-1]], 0)


def _topologies(rng):
    """(name, n, directed edge list [E, 2] as (src, dst))"""
    out = []
    n = 40
    out.append(("star40", n, _sym([(0, i) for i in range(1, n)])))                     # hub: 39 in-edges
    n = 12
    out.append(("complete12", n, np.array([(i, j) for i in range(n) for j in range(n) if i != j], dtype=np.int64)))
    n = 17
    out.append(("directed_cycle17", n, np.array([(i, (i + 1) % n) for i in range(n)], dtype=np.int64)))
    n = 9
    e = rng.integers(0, n, size=(30, 2))
    e = np.concatenate([e, e[:7], np.array([(i, i) for i in range(n)])], 0)            # duplicates + self loops
    out.append(("multigraph9", n, e[rng.permutation(len(e))]))
    n = 23
    e = _sym(rng.integers(0, 10, size=(25, 2)))                                        # nodes 10..22 isolated
    out.append(("isolated23", n, e[rng.permutation(len(e))]))
    n = 64
    out.append(("path64", n, _sym([(i, i + 1) for i in range(n - 1)])))                # the stage kernels' row limit
    n = 33
    e = _sym(rng.integers(0, n, size=(96, 2)))                                         # exactly 192 edges: the GINE stage's limit
    out.append(("edges192", n, e[rng.permutation(len(e))]))
    n = 5
    out.append(("wheel5_in_only", n, np.array([(i, 0) for i in range(1, n)] + [(0, 0)], dtype=np.int64)))   # in-edges of one node only
    return out


def _batch(topos, features, seed):
    g = torch.Generator().manual_seed(seed)
    eis, evecs, evals, batch, sizes = [], [], [], [], []
    off = 0
    for b, (_, n, e) in enumerate(topos):
        eis.append(torch.from_numpy(np.ascontiguousarray(e.T)) + off)
        q, _ = torch.linalg.qr(torch.randn(n, n, generator=g))                         # any orthonormal basis: the nets only read it
        evecs.append(q.reshape(-1).float())
        evals.append(torch.rand(n, generator=g))
        batch.append(torch.full((n,), b, dtype=torch.long))
        sizes.append(n)
        off += n
    edge_index = torch.cat(eis, 1).contiguous()
    N, E = off, edge_index.shape[1]
    if features == "zinc":
        x = torch.randint(0, 28, (N, 1), generator=g, dtype=torch.long)
        edge_attr = torch.randint(1, 4, (E,), generator=g, dtype=torch.long)
    else:
        x = torch.rand(N, 6, generator=g)
        edge_attr = torch.rand(E, 4, generator=g)
    d = types.SimpleNamespace(x=x, edge_index=edge_index, edge_attr=edge_attr, batch=torch.cat(batch), eigen_values=torch.cat(evals),
                              eigen_vectors=torch.cat(evecs), num_graphs=len(sizes), num_nodes=N)
    d.sizes = sizes
    return d


def _model(variant, ctor, max_k, seed=0):
    from signnet_basisnet_amd.pyg import SignNetGNN
    torch.manual_seed(seed)
    model = SignNetGNN(*ctor, variant=variant, max_k=max_k)
    model.attn_dropout = 0.0
    PU.bn_randomize(model, 2)
    return model


CASES = [("gine", "zinc", (None, None, 64, 1, 3, 3), 8), ("gine", "zinc", (None, None, 128, 1, 4, 6), 16),
         ("alchemy", "alchemy", (6, 4, 44, 5, 3, 3), None)]


@pytest.mark.parametrize("variant,feats,ctor,max_k", CASES, ids=["gine_d64_k8", "gine_d128_k16", "alchemy_d44_allk"])
def test_forward_on_arbitrary_topologies(variant, feats, ctor, max_k):
    from oracle import pyg_signnet as O
    from signnet_basisnet_amd import synth
    rng = np.random.default_rng(11)
    topos = _topologies(rng)
    model = _model(variant, ctor, max_k)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    cfg = O.make_cfg(variant, *ctor)
    m = model.to(DEV).eval()
    # every topology alone (a one-graph batch reaches the per-graph limits on its own), then all of them in one batch
    for group in [[t] for t in topos] + [topos, topos[::-1]]:
        host = _batch(group, feats, seed=3)
        names = "+".join(t[0] for t in group)
        ref = O.signnet_gnn(sd, cfg, host, training=False, max_k=max_k)
        ref64 = O.signnet_gnn(PU.to_f64(sd), cfg, PU.data_f64(host), training=False, max_k=max_k)
        dd = synth.batch_to(host, DEV)
        with torch.no_grad():
            y = m(dd)
            y_layer, _ = m(dd, return_stages=True)
        assert torch.isfinite(y).all(), names
        # (a one-graph batch has a single output row: its own cancellation sets the scale, so the case's measured conditioning — the
        #  fp32 oracle's distance from float64 — sets the bar, as for the small training fixtures)
        chk = PU.close_conditioned if len(group) == 1 else (lambda a, r32, r64, what: PU.close(a, r32, what, ref64=r64))
        chk(y, ref, ref64, f"{names}: forward (module default)")
        chk(y_layer, ref, ref64, f"{names}: layer path")


def test_forward_beyond_the_stage_kernels_limits_is_served_by_the_layer_path():
    """65 nodes / 193+ in-edges in one graph: the stage kernels flag the batch, strict mode (the default) re-runs it layer by layer."""
    from oracle import pyg_signnet as O
    from signnet_basisnet_amd import synth
    rng = np.random.default_rng(5)
    big = [("path65", 65, _sym([(i, i + 1) for i in range(64)])),
           ("dense30", 30, np.array([(i, j) for i in range(30) for j in range(30) if i != j][:400], dtype=np.int64)),
           ("star40", 40, _sym([(0, i) for i in range(1, 40)]))]
    variant, feats, ctor, max_k = CASES[0]
    model = _model(variant, ctor, max_k)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    m = model.to(DEV).eval()
    for group in ([big[0]], [big[1]], big):
        host = _batch(group, feats, seed=4)
        ref = O.signnet_gnn(sd, O.make_cfg(variant, *ctor), host, training=False, max_k=max_k)
        ref64 = O.signnet_gnn(PU.to_f64(sd), O.make_cfg(variant, *ctor), PU.data_f64(host), training=False, max_k=max_k)
        with torch.no_grad():
            y = m(synth.batch_to(host, DEV))
        PU.close_conditioned(y, ref, ref64, "+".join(t[0] for t in group) + ": forward beyond the stage limits")


@pytest.mark.parametrize("width", [8, 44, 128])
def test_aggregation_adjoints_on_arbitrary_topologies(width):
    """The GIN / GINE aggregations and their adjoints (out-edge CSR of the flipped edge list, edge permutation for the per-edge
    embeddings) on the batch of all topologies against dense float64 autograd.  (Whole-network gradients on such a small batch say
    nothing: one ReLU decision that falls on the other side of zero in either fp32 evaluation moves every upstream gradient by
    1e-4 ... 1e-3 — measured on ordinary molecule batches of this size as often as on these — so the adjoint kernels, which are
    linear in the cotangent, are checked directly; the gradients of the whole nets are checked at the BASELINE sizes in
    test_training_gpu.py.)"""
    from signnet_basisnet_amd import autograd as AG
    from signnet_basisnet_amd import ops
    from test_backward_gpu import run_pair
    rng = np.random.default_rng(11)
    topos = _topologies(rng)
    host = _batch(topos, "zinc", seed=3)
    ei, N, E, B = host.edge_index, host.num_nodes, host.edge_index.shape[1], host.num_graphs
    plan = ops.build_plan(host.batch.to(DEV), ei.to(DEV), B, 0)
    rplan = ops.build_plan(host.batch.to(DEV), ei.flip(0).contiguous().to(DEV), B, 0)
    g = torch.Generator().manual_seed(width)
    A = torch.zeros(N, N, dtype=torch.float64)
    A.index_put_((ei[1], ei[0]), torch.ones(E, dtype=torch.float64), accumulate=True)      # duplicate edges count twice
    x = torch.randn(N, 3 * width, generator=g)
    eps = torch.tensor([0.3])
    for neg in (False, True):
        run_pair(lambda x, eps: AG.gin_aggregate(x, eps, plan, rplan, negate=neg),
                 lambda x, eps: ((1 + eps) * x + A @ x) * (-1 if neg else 1), [x, eps], f"gin negate={neg} width={3 * width}")
    h, ee = torch.randn(N, width, generator=g), torch.randn(E, width, generator=g)
    # keep every ReLU argument away from zero: the comparison is of the adjoint, not of a decision at 1e-8
    s_ = h[ei[0]] + ee
    ee = ee + torch.where(s_.abs() < 1e-3, torch.sign(s_) * 1e-2 + (s_ == 0) * 1e-2, torch.zeros_like(s_))

    def ref(h, ee, eps):
        m = torch.relu(h[ei[0]] + ee)
        return (1 + eps) * h + torch.zeros_like(h).index_add_(0, ei[1], m)
    run_pair(lambda h, ee, eps: AG.gine_aggregate(h, ee, eps, plan, rplan), ref, [h, ee, eps], f"gine width={width}")


@pytest.mark.parametrize("name", ["gin", "gatedgcn", "gat", "pna", "transformer", "gatedgcn_mask", "pna_mask", "transformer_mask"])
def test_dgl_base_nets_on_arbitrary_topologies(name):
    """The GraphPrediction tree (sign-invariant net + each base net at its shipped width, eval mode) on the same graphs: nodes
    WITHOUT in-edges (DGL's reducers leave zeros there: layers/pna_layer.py:38-56 mean / max / min / std over an empty mailbox,
    gat_layer.py's edge softmax, graph_transformer_edge_layer.py's z = 0), 39 in-edges on one node (PNA's degree scalers, the
    wave-per-(node, head) GAT / Transformer aggregations), duplicate edges and self loops, among ordinary molecules."""
    from signnet_basisnet_amd import synth
    from test_full_size_parity_gpu import run_shipped_dgl_config
    rng = np.random.default_rng(11)
    topos = [t for t in _topologies(rng) if t[1] <= 37]          # k = 37 = the largest ZINC graph in the masked configs
    topos.append(("star37", 37, _sym([(0, i) for i in range(1, 37)])))
    mols = [(f"mol{i}", n, synth._random_molecule(rng, n).T) for i, n in enumerate(rng.integers(9, 31, size=10).tolist())]
    mixed = [t for pair in zip(mols, topos) for t in pair] + mols[len(topos):]
    if name == "gat":
        # DGL's GATConv refuses a graph with 0-in-degree nodes (DGLError; nets/ZINC_graph_regression/gat_net.py:62-66 leaves
        # allow_zero_in_degree at False): so does the HIP net — and evaluates the batch without those two graphs
        with pytest.raises(ValueError, match="0-in-degree"):
            run_shipped_dgl_config(name, _batch(mixed, "zinc", seed=8), elementwise=False)
        mixed = [t for t in mixed if t[0] not in ("isolated23", "wheel5_in_only")]
    run_shipped_dgl_config(name, _batch(mixed, "zinc", seed=8), elementwise=False)


def test_gatedgcn_graph_beyond_the_one_launch_kernel_takes_the_layer_path():
    """A graph with more in-edges than the one-launch GatedGCN kernel stages in LDS (sn_gatedgcn_max_edges: 176 at hidden 68) is
    evaluated by the layer path, decided on the host from the per-graph edge counts a DGL batch carries (batch_num_edges(), cached on
    the graph object) — the reference evaluates any graph.  Without them the device-side guard applies (NaN score, check_last())."""
    from signnet_basisnet_amd import dgl_deepsigns as DS
    from signnet_basisnet_amd import dgl_nets
    from test_full_size_parity_gpu import SHIPPED, _COMMON
    params = dict(_COMMON, device=DEV)
    params.update({k: v for k, v in SHIPPED["gatedgcn"].items() if k != "cls"})
    net = dgl_nets.GatedGCNNet(params).to(DEV).eval()
    rng = np.random.default_rng(3)
    small = _batch([("path20", 20, _sym([(i, i + 1) for i in range(19)]))], "zinc", seed=1)
    dense = _batch([("k15", 15, np.array([(i, j) for i in range(15) for j in range(15) if i != j], dtype=np.int64))], "zinc", seed=1)   # 210 in-edges
    for host, fused in ((small, True), (dense, False)):
        g = DS.Graph(host.edge_index[0].to(DEV), host.edge_index[1].to(DEV), host.sizes, [host.edge_index.shape[1]])
        assert (net._fused_gated(g) is not None) == fused
        assert DS._max_in_edges(g) == host.edge_index.shape[1]
    g = DS.Graph(dense.edge_index[0].to(DEV), dense.edge_index[1].to(DEV), dense.sizes)        # no batch_num_edges(): not known on the host
    assert DS._max_in_edges(g) is None and net._fused_gated(g) is not None


@pytest.mark.parametrize("norm", [None, "sym"])
def test_evd_on_degenerate_spectra(norm):
    """The on-device eigendecomposition (row f2) on graphs whose Laplacians have heavily repeated eigenvalues — a star (eigenvalue 1
    with multiplicity n - 2), K12 (n - 1 times the same eigenvalue), cycles (pairs), 13 isolated nodes (a 14-fold zero), a 64-node
    path, one- and two-node graphs — where the molecule batches of test_evd_gpu.py have almost simple spectra: eigenvalues,
    residual, orthogonality and the projectors onto separated clusters against the CPU oracle (LAPACK), same tolerance."""
    from signnet_basisnet_amd import transform as TR
    from test_evd_gpu import _check_batch, _gptr
    rng = np.random.default_rng(11)
    topos = _topologies(rng)
    topos += [("cycle16", 16, _sym([(i, (i + 1) % 16) for i in range(16)])), ("single", 1, np.zeros((0, 2), dtype=np.int64)),
              ("pair", 2, _sym([(0, 1)])), ("empty5", 5, np.zeros((0, 2), dtype=np.int64)),
              ("grid8x8", 64, _sym([(8 * i + j, 8 * i + j + 1) for i in range(8) for j in range(7)] + [(8 * i + j, 8 * i + j + 8) for i in range(7) for j in range(8)]))]
    host = _batch(topos, "zinc", seed=2)
    # the Laplacian of the transform is that of the undirected simple graph: symmetrised, without self loops and duplicates, as the
    # oracle builds it (GINESignNetPyG/core/transform.py:29-52 reads to_dense_adj of an undirected edge_index)
    ei = host.edge_index
    ei = torch.cat([ei, ei.flip(0)], 1)
    ei = ei[:, ei[0] != ei[1]]
    ei = torch.unique(ei, dim=1)
    D, V = TR.evd_laplacian_batch(ei.to(DEV), ptr=_gptr(host.sizes), norm=norm)[:2]
    worst = _check_batch(ei.numpy(), list(host.sizes), norm, D.cpu().numpy(), V.cpu().numpy())
    print(f"\nEVD on degenerate spectra, norm = {norm}: worst {worst}")


@pytest.mark.parametrize("variant,feats,ctor,max_k", CASES, ids=["gine_d64_k8", "gine_d128_k16", "alchemy_d44_allk"])
def test_train_mode_forward_on_arbitrary_topologies(variant, feats, ctor, max_k):
    """The differentiable train-mode forward (batch-statistics BatchNorm, the one-pass stage kernels of csrc/train.hip and the
    layer-at-a-time kernels) on the topologies among ordinary molecules: value against the fp32 / float64 oracle, both train paths."""
    from oracle import pyg_signnet as O
    from signnet_basisnet_amd import synth
    rng = np.random.default_rng(11)
    topos = _topologies(rng)
    mols = [(f"mol{i}", n, synth._random_molecule(rng, n).T) for i, n in enumerate(rng.integers(9, 31, size=24).tolist())]
    host = _batch([t for pair in zip(mols[:8], topos) for t in pair] + mols[8:], feats, seed=3)
    model = _model(variant, ctor, max_k)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    cfg = O.make_cfg(variant, *ctor)
    with torch.no_grad():
        ref = O.signnet_gnn({k: v.clone() for k, v in sd.items()}, cfg, host, training=True, max_k=max_k)
        ref64 = O.signnet_gnn(PU.to_f64(sd), cfg, PU.data_f64(host), training=True, max_k=max_k)
    m = model.to(DEV).train()
    dd = synth.batch_to(host, DEV)
    for stages in (True, False):
        m.load_state_dict(sd)                      # (the train-mode forward updates the running statistics)
        m.train_stages = stages
        y = m(dd)
        assert y.requires_grad
        # (train-mode BatchNorm over the 32 rows of the readout: the case's measured conditioning sets the bar, as for the fixtures)
        e = PU.close_conditioned(y.detach(), ref, ref64, f"train-mode forward, stage kernels {stages}")
        print(f"\n{variant} d={ctor[2]} stage kernels {stages}: |hip - ref| {e:.2e}, |cpu32 - f64| {PU.relerr(ref, ref64):.2e}")


def test_overlap_mode_on_odd_batches_is_bit_identical():
    """SignNetGNN.overlap_front (three-stage stream pipeline inside the module) over a sequence of very different resident batches —
    the topology batch, a batch without edges, a single one-node graph, an ordinary batch — equals the sequential forwards bit for
    bit, in order (a stage of one forward runs beside other stages of its neighbours: batches of 1 ... 1300 rows)."""
    from signnet_basisnet_amd import synth
    rng = np.random.default_rng(11)
    topos = _topologies(rng)
    hosts = [_batch(topos, "zinc", seed=3), synth.make_batch(4, seed=5, sizes=[1, 1, 1, 1]), synth.make_batch(1, seed=6, sizes=[1]),
             synth.make_batch(48, seed=7), _batch(topos[::-1], "zinc", seed=4), synth.make_batch(2, seed=8, sizes=[1, 30])]
    m = _model("gine", (None, None, 128, 1, 4, 6), 16).to(DEV).eval()
    with torch.no_grad():
        ref = [m(synth.batch_to(h, DEV)).clone() for h in hosts]
        assert all(torch.isfinite(r).all() for r in ref)
        m.strict, m.overlap_front = False, True
        outs = []
        for rep in range(4):
            for i, h in enumerate(hosts):
                b = synth.batch_to(h, DEV)
                torch.cuda.synchronize()                 # resident batch: the mode's precondition
                outs.append((i, m(b) * 1.0))
                del b
        m.check_last()
        torch.cuda.synchronize()
    for i, y in outs:
        assert torch.equal(y, ref[i]), i


def test_whole_network_gradients_on_the_topology_batch():
    """d loss / d theta of ONE training step on the eight-topology batch (hub with 39 in-edges, K12, directed cycle, multigraph with
    self loops, isolated nodes, the 64-node / 192-edge limits, in-edges only), fixed seed, every parameter tensor, against
    torch.autograd over the fp32 AND the float64 oracle — the population rule of tests/test_training_gpu.py (a wrong adjoint is O(1);
    two fp32 evaluations of a network with batch-statistics BatchNorms differ by their ReLU / BatchNorm decisions).  Until round 4 the
    gradients on these topologies were only checked op by op (the aggregation adjoints above)."""
    from oracle import pyg_signnet as O
    from signnet_basisnet_amd import synth
    from test_training_gpu import _assert_gradient_population
    variant, feats, ctor, max_k = CASES[0]
    topos = _topologies(np.random.default_rng(11))
    host = _batch(topos, feats, seed=3)
    model = _model(variant, ctor, max_k, seed=4)
    cfg = O.make_cfg(variant, *ctor)
    cot = torch.randn(host.num_graphs, ctor[3], generator=torch.Generator().manual_seed(1), dtype=torch.float64)

    def oracle(dt):
        sd = {}
        for k, v in model.state_dict().items():
            if v.is_floating_point():
                t = v.detach().clone().to(dt)
                sd[k] = t.requires_grad_(True) if "running" not in k else t
            else:
                sd[k] = v.detach().clone()
        dd = PU.data_f64(host) if dt == torch.float64 else host
        y = O.signnet_gnn(sd, cfg, dd, training=True, max_k=max_k)
        (y * cot.to(dt)).sum().backward()
        return y.detach(), sd

    y64, sd64 = oracle(torch.float64)
    y32, sd32 = oracle(torch.float32)
    m = model.to(DEV).train()
    y = m(synth.batch_to(host, DEV))
    assert y.requires_grad
    (y * cot.float().to(DEV)).sum().backward()
    PU.close(y, y32, "train-mode forward on the topology batch", ref64=y64)
    _assert_gradient_population(m.named_parameters(), sd32, sd64, "topology batch", 30)


def test_a_graph_without_nodes_is_evaluated_like_the_reference_or_flagged():
    """A batch whose graph 1 has no nodes (PyG's add-pooling gives such a graph a zero row, the output encoder a finite score): the
    module's default mode evaluates it (the plan kernel's early report routes the batch to the layer path), the serving mode returns
    NaN for that graph only and raises at the next check."""
    from oracle import pyg_signnet as O
    from signnet_basisnet_amd import synth
    from signnet_basisnet_amd.pyg import SignNetGNN
    torch.manual_seed(0)
    ctor = (None, None, 32, 1, 2, 2)
    model = SignNetGNN(*ctor, variant="gine", max_k=8)
    data = synth.make_batch(4, seed=3, sizes=[5, 0, 7, 3])
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    ref = O.signnet_gnn(sd, O.make_cfg("gine", *ctor), data, training=False, max_k=8)
    model = model.cuda().eval()
    dd = synth.batch_to(data, "cuda:0")
    with torch.no_grad():
        y = model(dd)
    err = (y.cpu() - ref).abs().max().item() / ref.abs().max().item()
    assert err <= 1e-5, err
    model.strict = False
    with torch.no_grad():
        y2 = model(dd)
    torch.cuda.synchronize()
    assert torch.isnan(y2[1]).all() and torch.allclose(y2[[0, 2, 3]].cpu(), ref[[0, 2, 3]], rtol=1e-4, atol=1e-5)
    with pytest.raises(RuntimeError, match="without nodes"):
        model.check_last()
