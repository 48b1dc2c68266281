"""-m gpu: serving.GraphedDGLForward — the DGL tree's eval forward (sign_inv_net + base network) recorded once as a HIP graph for a
fixed batch SHAPE and replayed: bit-identical to the eager forward, also for another batch of the same shape taken through the
static input buffers; a batch of another shape is refused."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _inputs(data, k):
    from signnet_basisnet_amd import dgl_deepsigns as DS
    from signnet_basisnet_amd import synth
    pe = synth.dgl_pos_enc(data, k).unsqueeze(-1).to(DEV)
    src, dst = data.edge_index
    sizes = torch.tensor(data.sizes)
    bne = torch.bincount(torch.bucketize(dst, torch.cumsum(sizes, 0), right=True), minlength=len(data.sizes))
    g = DS.Graph(src.to(DEV), dst.to(DEV), sizes, bne)
    sn = torch.cat([torch.full((n, 1), 1.0 / n) for n in data.sizes]).sqrt().to(DEV)
    return g, data.x.squeeze(-1).to(DEV), pe, data.edge_attr.to(DEV), sn


def _permuted(data, perm):
    """The same graphs in another order: same node / edge / graph counts, different batch."""
    import types
    sizes = list(data.sizes)
    starts = [0]
    for n in sizes:
        starts.append(starts[-1] + n)
    vst = [0]
    for n in sizes:
        vst.append(vst[-1] + n * n)
    xs, evs, evl, eis, eas, new_sizes = [], [], [], [], [], []
    off = 0
    src, dst = data.edge_index
    for b in perm:
        lo, hi = starts[b], starts[b + 1]
        m = (src >= lo) & (src < hi)
        eis.append(data.edge_index[:, m] - lo + off)
        eas.append(data.edge_attr[m])
        xs.append(data.x[lo:hi]); evl.append(data.eigen_values[lo:hi]); evs.append(data.eigen_vectors[vst[b]:vst[b + 1]])
        new_sizes.append(sizes[b])
        off += sizes[b]
    d = types.SimpleNamespace(x=torch.cat(xs), edge_index=torch.cat(eis, 1), edge_attr=torch.cat(eas),
                              batch=torch.repeat_interleave(torch.arange(len(perm)), torch.tensor(new_sizes)),
                              eigen_values=torch.cat(evl), eigen_vectors=torch.cat(evs), num_graphs=len(perm), num_nodes=off)
    d.sizes = new_sizes
    return d


@pytest.mark.parametrize("name", ["gin", "gat", "pna", "transformer", "gatedgcn"])
def test_graphed_dgl_forward_replays_the_eager_forward(name):
    from signnet_basisnet_amd import dgl_configs, dgl_nets, synth
    from signnet_basisnet_amd.serving import GraphedDGLForward
    import parity_util as PU
    cls, params = dgl_configs.net_params(name, DEV)
    params.update(L=3)                                       # the shipped widths / heads / towers, three layers
    torch.manual_seed(2)
    net = getattr(dgl_nets, cls)(params)
    PU.bn_randomize(net, 3)
    net = net.to(DEV).eval()
    k = params["pos_enc_dim"]
    a = synth.make_batch(24, seed=11)
    b = _permuted(a, list(reversed(range(24))))
    ga, ha, pa, ea, sa = _inputs(a, k)
    gb, hb, pb, eb, sb = _inputs(b, k)
    snorm = name == "pna"

    def eager(g, h, p, e, s):
        with torch.no_grad():
            q = net.sign_inv_net(g, p).squeeze(-1)
            return net(g, h, q, e, s if snorm else None)[0].clone()

    ya, yb = eager(ga, ha, pa, ea, sa), eager(gb, hb, pb, eb, sb)
    assert not torch.equal(ya, yb[: ya.shape[0]])            # a different batch indeed
    gf = GraphedDGLForward(net, ga, ha, pa, ea, sa if snorm else None)
    assert torch.equal(gf().clone(), ya)
    assert torch.equal(gf(gb, hb, pb, eb, sb if snorm else None).clone(), yb)
    assert torch.equal(gf(ga, ha, pa, ea, sa if snorm else None).clone(), ya)
    gf.check()
    if name != "gatedgcn":                                   # (its one-launch kernel reports through check_last(); covered elsewhere)
        bad = hb.clone()
        bad[3] = 1000                                        # an atom type outside the table: flagged on the device, raised by check()
        gf(gb, bad, pb, eb, sb if snorm else None)
        with pytest.raises(IndexError):
            gf.check()
        gf(gb, hb, pb, eb, sb if snorm else None)
        gf.check()
    c = synth.make_batch(24, seed=12)
    gc, hc, pc, ec, sc = _inputs(c, k)
    with pytest.raises(ValueError, match="shape"):
        gf(gc, hc, pc, ec, sc if snorm else None)
    # a rejected call copied nothing (every argument is validated first): the recorded inputs are still batch b's
    assert torch.equal(gf(gb, hb, pb, eb, sb if snorm else None).clone(), yb)
    with pytest.raises(ValueError, match="shape"):
        gf(ga, ha, pa[:, :-1] if pa.shape[1] > 1 else pa[:-1], ea, sa if snorm else None)
    assert torch.equal(gf().clone(), yb)
    # the weights are frozen at capture: after a mode round trip (which drops the packed copies the recorded launches point into),
    # an in-place update or a re-load, a replay raises instead of reading freed or half-updated weights
    net.train()
    net.eval()
    with pytest.raises(RuntimeError, match="changed since the forward was recorded"):
        gf()
    gf2 = GraphedDGLForward(net, ga, ha, pa, ea, sa if snorm else None)
    assert torch.equal(gf2().clone(), ya)
    with torch.no_grad():
        next(net.parameters()).mul_(1.0)
    with pytest.raises(RuntimeError, match="changed since the forward was recorded"):
        gf2()


def test_transformer_net_fused_eval_layers_equal_the_op_by_op_path():
    """TransformerNet eval (round 4): Q | K | V as one [3d, d] Linear read in place by the attention, every layer's E projection in one
    Linear up front, BatchNorm(x + Linear(h)) as the Linear's epilogue — 5 launches per layer instead of 12, bit-identical."""
    from signnet_basisnet_amd import dgl_configs, dgl_nets, ops, synth
    import parity_util as PU
    cls, params = dgl_configs.net_params("transformer", DEV)
    params.update(L=4)
    torch.manual_seed(2)
    net = getattr(dgl_nets, cls)(params)
    PU.bn_randomize(net, 3)
    net = net.to(DEV).eval()
    net.fused_stages = False          # (the layer path's own fusions are compared here; the one-launch stage kernel has its own test)
    a = synth.make_batch(24, seed=11)
    g, h, pe, e, _ = _inputs(a, params["pos_enc_dim"])
    with torch.no_grad():
        p = net.sign_inv_net(g, pe).squeeze(-1)
        rec = ops.KernelTimer()
        with rec:
            y_fused = net(g, h, p, e, None)[0].clone()
        n_fused = sum(v[0] for v in rec.summary().values())
        net.fused_layers = False
        rec = ops.KernelTimer()
        with rec:
            y_ops = net(g, h, p, e, None)[0].clone()
        n_ops = sum(v[0] for v in rec.summary().values())
    assert torch.equal(y_fused, y_ops)
    assert n_fused <= n_ops - 2 * params["L"], (n_fused, n_ops)      # (the timer sees the launches that go through ops._span: not the
                                                                      #  four pointwise passes per layer the fused epilogues also replace)


@pytest.mark.parametrize("hidden,readout", [(95, "mean"), (64, "sum"), (128, "mean"), (40, "sum")])
def test_gin_net_one_launch_eval_equals_the_layer_path(hidden, readout):
    """GINNet eval (round 4): embeddings, the L GIN layers, the readout and MLPReadout in ONE launch (sn_gin_net_fused_f32: a workgroup
    per graph on the PyG tree's GINE stage kernel) against the layer path (fused_stages = False) — widths that pad to 64 / 96 / 128,
    both readouts; the same values up to the rounding of different GEMM orders."""
    from signnet_basisnet_amd import dgl_configs, dgl_nets, ops, synth
    import parity_util as PU
    cls, params = dgl_configs.net_params("gin", DEV)
    params.update(hidden_dim=hidden, out_dim=hidden, readout=readout, L=5)
    torch.manual_seed(4)
    net = getattr(dgl_nets, cls)(params)
    PU.bn_randomize(net, 5)
    net = net.to(DEV).eval()
    a = synth.make_batch(40, seed=21)
    g, h, pe, e, _ = _inputs(a, params["pos_enc_dim"])
    with torch.no_grad():
        p = net.sign_inv_net(g, pe).squeeze(-1)
        rec = ops.KernelTimer()
        with rec:
            y_one = net(g, h, p, e, None)[0].clone()
        assert "sn_gin_net_fused_f32" in rec.summary(), rec.summary().keys()
        net.check_last()
        net.fused_stages = False
        y_lay = net(g, h, p, e, None)[0].clone()
        net.fused_stages = True
        assert torch.isfinite(y_one).all()
        scale = y_lay.abs().max().clamp_min(1e-6)
        assert ((y_one - y_lay).abs().max() / scale).item() < 2e-5, ((y_one - y_lay).abs().max() / scale).item()
        bad = h.clone()
        bad[3] = 1000                       # an atom type outside the table: NaN score for that graph, IndexError from check_last()
        y_bad = net(g, bad, p, e, None)[0]
        with pytest.raises(IndexError):
            net.check_last()
        assert torch.isnan(y_bad).any()


def test_pna_net_fused_eval_layers_equal_the_op_by_op_path():
    """PNANet eval (round 4): every layer's edge term W_e e + b as ONE [E, L*C] Linear read in place by the aggregation, and the mixing
    FCLayer's LeakyReLU + residual as its Linear's epilogue (SN_EPI_LEAKY) — 4 launches per layer instead of 6, the same bits."""
    from signnet_basisnet_amd import dgl_configs, dgl_nets, ops, synth
    import parity_util as PU
    cls, params = dgl_configs.net_params("pna", DEV)
    params.update(L=4)
    torch.manual_seed(2)
    net = getattr(dgl_nets, cls)(params)
    PU.bn_randomize(net, 3)
    net = net.to(DEV).eval()
    a = synth.make_batch(24, seed=11)
    g, h, pe, e, sn = _inputs(a, params["pos_enc_dim"])
    with torch.no_grad():
        p = net.sign_inv_net(g, pe).squeeze(-1)
        rec = ops.KernelTimer()
        with rec:
            y_fused = net(g, h, p, e, sn)[0].clone()
        n_fused = sum(v[0] for v in rec.summary().values())
        net.fused_layers = False
        rec = ops.KernelTimer()
        with rec:
            y_ops = net(g, h, p, e, sn)[0].clone()
        n_ops = sum(v[0] for v in rec.summary().values())
    assert torch.equal(y_fused, y_ops)
    assert n_fused <= n_ops - (params["L"] - 1), (n_fused, n_ops)    # (the timer sees the L - 1 edge-term Linears that went away; the pointwise
                                                                      #  pass per layer that the epilogue also replaces is not one of its spans)


@pytest.mark.parametrize("readout,L", [("mean", 3), ("sum", 10)])
def test_transformer_net_one_launch_eval_equals_the_layer_path(readout, L):
    """TransformerNet eval (round 4): embeddings, the L layers (Q | K | V, edge attention, O_h, FFN, both BatchNorms), readout and
    MLPReadout in ONE launch (sn_transformer_net_fused_f32: a workgroup per graph, Transformer mode of the GINE stage kernel) against the
    layer path (fused_stages = False); the same values up to the rounding of different GEMM orders."""
    from signnet_basisnet_amd import dgl_configs, dgl_nets, ops, synth
    import parity_util as PU
    cls, params = dgl_configs.net_params("transformer", DEV)
    params.update(readout=readout, L=L)
    torch.manual_seed(4)
    net = getattr(dgl_nets, cls)(params)
    PU.bn_randomize(net, 5)
    net = net.to(DEV).eval()
    a = synth.make_batch(40, seed=21)
    g, h, pe, e, _ = _inputs(a, params["pos_enc_dim"])
    with torch.no_grad():
        p = net.sign_inv_net(g, pe).squeeze(-1)
        rec = ops.KernelTimer()
        with rec:
            y_one = net(g, h, p, e, None)[0].clone()
        assert "sn_transformer_net_fused_f32" in rec.summary(), rec.summary().keys()
        net.check_last()
        net.fused_stages = False
        y_lay = net(g, h, p, e, None)[0].clone()
        net.fused_stages = True
        assert torch.isfinite(y_one).all()
        scale = y_lay.abs().max().clamp_min(1e-6)
        assert ((y_one - y_lay).abs().max() / scale).item() < 5e-5, ((y_one - y_lay).abs().max() / scale).item()
        bad = h.clone()
        bad[3] = 1000
        y_bad = net(g, bad, p, e, None)[0]
        with pytest.raises(IndexError):
            net.check_last()
        assert torch.isnan(y_bad).any()


def _odd_batch(rng, B, kind):
    """Explicit edge lists in shuffled order: molecules, stars (one node with up to 39 in-edges), random multi-edge graphs, paths of up
    to 63 nodes, single nodes."""
    import numpy as np
    from signnet_basisnet_amd import synth
    sizes, eis, off = [], [], 0
    for _ in range(B):
        k = kind if kind != "mix" else str(rng.choice(["mol", "star", "multi", "single", "path"]))
        if k == "single":
            n, ei = 1, np.zeros((2, 0), dtype=np.int64)
        elif k == "star":
            n = int(rng.integers(3, 40)); leaves = np.arange(1, n); hub = np.zeros(n - 1, dtype=np.int64)
            ei = np.concatenate([np.stack([leaves, hub]), np.stack([hub, leaves])], 1)
        elif k == "multi":
            n = int(rng.integers(2, 20)); m = int(rng.integers(n, 4 * n))
            ei = np.stack([rng.integers(0, n, m), rng.integers(0, n, m)])
        elif k == "path":
            n = int(rng.integers(2, 64)); a = np.arange(n - 1)
            ei = np.concatenate([np.stack([a, a + 1]), np.stack([a + 1, a])], 1)
        else:
            n = int(rng.integers(2, 38)); ei = synth._random_molecule(rng, n)
        eis.append(ei[:, rng.permutation(ei.shape[1])] + off); sizes.append(n); off += n
    return sizes, torch.from_numpy(np.ascontiguousarray(np.concatenate(eis, 1))).long(), off


@pytest.mark.parametrize("seed", range(12))
def test_one_launch_dgl_nets_on_random_shapes_and_odd_graphs(seed):
    """The one-launch GIN / Transformer nets against their layer paths over random widths (16 ... 128), depths, readouts, `add` / `concat`
    positional encodings and batches of molecules, stars, multi-edge graphs, long paths and single nodes with shuffled edge lists
    (60 such draws were run when the kernels were written: all within 3e-6)."""
    import numpy as np
    from signnet_basisnet_amd import dgl_configs, dgl_nets, ops, dgl_deepsigns as DS
    import parity_util as PU
    rng = np.random.default_rng(1000 + seed)
    name = ["gin", "transformer"][seed % 2]
    cls, params = dgl_configs.net_params(name, DEV)
    params.update(L=int(rng.integers(1, 7)), readout=str(rng.choice(["sum", "mean"])))
    if name == "gin":
        hid = int(rng.choice([16, 33, 64, 95, 100, 128]))
        params.update(hidden_dim=hid, out_dim=hid)
    else:
        params.update(pe_aggregate=str(rng.choice(["add", "concat"])))
    torch.manual_seed(seed)
    net = getattr(dgl_nets, cls)(params)
    PU.bn_randomize(net, seed)
    net = net.to(DEV).eval()
    sizes, ei, N = _odd_batch(rng, int(rng.integers(1, 40)), str(rng.choice(["mol", "mix", "star", "multi"])))
    szt = torch.tensor(sizes)
    bne = torch.bincount(torch.bucketize(ei[1], torch.cumsum(szt, 0), right=True), minlength=len(sizes))
    gt = torch.Generator().manual_seed(seed)
    h = torch.randint(0, 28, (N,), generator=gt).to(DEV)
    e = torch.randint(1, 4, (ei.shape[1],), generator=gt).to(DEV)
    p = torch.randn(N, params["pos_enc_dim"], generator=gt).to(DEV)
    with torch.no_grad():
        rec = ops.KernelTimer()
        with rec:
            y_one = net(DS.Graph(ei[0].to(DEV), ei[1].to(DEV), szt, bne), h, p, e, None)[0].clone()
        assert any("net_fused" in k for k in rec.summary()), rec.summary().keys()
        net.check_last()
        net.fused_stages = False
        y_lay = net(DS.Graph(ei[0].to(DEV), ei[1].to(DEV), szt, bne), h, p, e, None)[0].clone()
    assert torch.isfinite(y_one).all()
    err = ((y_one - y_lay).abs().max() / y_lay.abs().max().clamp_min(1e-6)).item()
    assert err < 2e-5, (name, params["hidden_dim"], params["L"], err)


@pytest.mark.parametrize("name", ["gin", "transformer"])
def test_one_launch_dgl_nets_route_or_flag_graphs_beyond_the_kernel_limits(name):
    """A graph with more in-edges than the stage kernel stages (192) or more than 64 nodes: with the batch object's counts the net takes
    the layer path (finite scores, no stage launch); a duck-typed graph WITHOUT edge counts reaches the kernel, which hands back NaN for
    that graph only and flags it — check_last() raises."""
    import numpy as np
    from signnet_basisnet_amd import dgl_configs, dgl_nets, ops, dgl_deepsigns as DS
    import parity_util as PU
    cls, params = dgl_configs.net_params(name, DEV)
    params.update(L=2)
    torch.manual_seed(0)
    net = getattr(dgl_nets, cls)(params)
    PU.bn_randomize(net, 1)
    net = net.to(DEV).eval()
    rng = np.random.default_rng(5)
    # graph 0: 20 nodes, 300 random edges (> 192); graph 1: a 10-node path
    e0 = np.stack([rng.integers(0, 20, 300), rng.integers(0, 20, 300)])
    a = np.arange(9)
    e1 = np.concatenate([np.stack([a, a + 1]), np.stack([a + 1, a])], 1) + 20
    ei = torch.from_numpy(np.concatenate([e0, e1], 1)).long()
    szt = torch.tensor([20, 10])
    bne = torch.tensor([300, 18])
    gt = torch.Generator().manual_seed(1)
    h = torch.randint(0, 28, (30,), generator=gt).to(DEV)
    e = torch.randint(1, 4, (ei.shape[1],), generator=gt).to(DEV)
    p = torch.randn(30, params["pos_enc_dim"], generator=gt).to(DEV)
    with torch.no_grad():
        rec = ops.KernelTimer()
        with rec:
            y_lay = net(DS.Graph(ei[0].to(DEV), ei[1].to(DEV), szt, bne), h, p, e, None)[0].clone()
        assert not any("net_fused" in k for k in rec.summary()) and torch.isfinite(y_lay).all()
        y = net(DS.Graph(ei[0].to(DEV), ei[1].to(DEV), szt, None), h, p, e, None)[0].clone()      # no edge counts: the kernel decides
        assert torch.isnan(y[0]).all() and torch.isfinite(y[1]).all()
        assert ((y[1] - y_lay[1]).abs() / y_lay.abs().max().clamp_min(1e-6)).max().item() < 2e-5
        with pytest.raises(RuntimeError, match="192 in-edges"):
            net.check_last()
