"""-m gpu: the HIP SignNetGNN modules against the golden fixtures (reference outputs) and the oracle."""
import pytest
import torch

import golden_util as G
import parity_util as PU
from parity_util import close      # max|hip - ref| <= 1e-5 * max|ref| (north_star), float64 attribution when `ref64` is given

pytestmark = pytest.mark.gpu


def build(fx, max_k=None):
    from signnet_basisnet_amd.pyg import SignNetGNN
    c = [None if v < 0 else int(v) for v in fx.meta["ctor"]]
    variant = str(fx.meta["variant"])
    m = SignNetGNN(*c, variant=variant, max_k=max_k)
    sd = G.full_state_dict(fx)
    assert sorted(m.state_dict().keys()) == sorted(sd.keys()), "state_dict key layout differs from the reference"
    for k, v in m.state_dict().items():
        assert tuple(v.shape) == tuple(sd[k].shape), k
    m.load_state_dict(sd)
    m.attn_dropout = 0.0        # the train-mode fixtures were generated with the (random) attention dropout switched off
    return m.cuda().eval()


@pytest.mark.parametrize("name", G.PYG_CASES)
def test_golden(name):
    from signnet_basisnet_amd import synth
    from oracle import pyg_signnet as O
    fx = G.load(name)
    model = build(fx)
    data = synth.batch_to(G.as_data(fx.inp), "cuda:0")
    y, st = model(data, return_stages=True)
    # the exact (float64) value of every stage, for attribution where two fp32 evaluations differ by ~1e-5 (parity_util.close)
    o64 = {}
    with torch.no_grad():
        y64 = O.signnet_gnn(PU.to_f64(fx.sd), G.pyg_cfg(fx), PU.data_f64(G.as_data(fx.inp)), training=False, out=o64)
    close(st["phi_plus"], fx.out["eval/phi_plus"], "phi(+x)", ref64=o64["phi_plus_layers"][-1])
    close(st["phi_minus"], fx.out["eval/phi_minus"], "phi(-x)", ref64=o64["phi_minus_layers"][-1])
    close(st["pos"], fx.out["eval/pos"], "sign_net output", ref64=o64["pos"])
    close(y, fx.out["eval/y"], "model output", ref64=y64)
    # fused phi kernel: same quantity as the layer path and as the reference
    assert st["bins_meta"].cpu().tolist()[1] == 0 and st["bins_meta"].cpu().tolist()[5] == 0
    close(st["phi_fused"], fx.out["eval/phi_plus"] + fx.out["eval/phi_minus"], "fused phi(x)+phi(-x)", ref64=o64["phi"])
    close(st["y_gnn_fused"], fx.out["eval/y"], "fused gnn output (from the layer-path slot sum)", ref64=y64)
    close(st["rho_sum_fused"], st["rho_sum"], "fused rho slot-sum vs layer path", ref64=o64["rho_sum"])
    # and the default forward (fused stages) gives the reference output
    close(model(data), fx.out["eval/y"], "model output (fused path)", ref64=y64)


def test_plan_bins():
    """Work-bin invariants.  phi: every graph sits in exactly one column, members do not overlap and fit 64 rows,
    column heights are max K_g, bin_col inverts col_bin0.  rho: bins per graph in closed form."""
    from signnet_basisnet_amd import ops, synth
    data = synth.make_batch(40, seed=9)
    d = synth.batch_to(data, "cuda:0")
    n = torch.tensor(data.sizes)
    for kmax in (0, 16, 5):
        kg = n.clamp(max=kmax) if kmax else n
        plan = ops.build_plan(d.batch, d.edge_index, d.num_graphs, kmax, bins=True, columns=True)
        assert plan.check()[0] == 0
        b = plan.bins
        meta = b.meta.cpu().tolist()
        nbins, err, rows, ncol = meta[:4]
        assert err == 0 and rows == int((n * kg).sum()) and nbins <= b.phi_max_bins and meta[7] == nbins
        if kmax == 0:
            # all eigenvectors on the one-launch plan: slabs of any graphs per bin, no columns (tests/test_ops_gpu.py checks the records)
            assert ncol == 0 and rows / (nbins * 64) > 0.85
        mem = b.phi_col_mem.cpu().view(-1, 8)[:ncol]
        off = b.phi_col_off.cpu().view(-1, 8)[:ncol]
        cb0 = b.phi_col_bin0.cpu()[:ncol + 1]
        seen = []
        for c in range(ncol):
            gs = [int(g) for g in mem[c] if g >= 0]
            assert gs, "empty column"
            seen += gs
            spans = sorted((int(off[c, k]), int(off[c, k]) + data.sizes[int(mem[c, k])]) for k in range(len(gs)))
            assert spans[0][0] == 0 and spans[-1][1] <= 64
            assert all(spans[i][1] <= spans[i + 1][0] for i in range(len(spans) - 1))
            assert int(cb0[c + 1] - cb0[c]) == max(int(kg[g]) for g in gs)
        if kmax != 0:
            assert sorted(seen) == list(range(len(data.sizes)))
            assert int(cb0[ncol]) == nbins
            bc = b.phi_bin_col.cpu()[:nbins]
            assert bc.tolist() == torch.repeat_interleave(torch.arange(ncol), (cb0[1:] - cb0[:-1]).long()).tolist()
            assert rows / (nbins * 64) > 0.8, "best-fit-decreasing should pack the phi bins well"
        # rho
        rnb, rerr, rrows = meta[4:7]
        pad = ((kg + 15) // 16) * 16
        per_graph = (n + (64 // pad) - 1) // (64 // pad)
        assert rerr == 0 and rrows == rows and rnb == int(per_graph.sum())
        assert b.rho_bin0.cpu().tolist() == torch.cat([torch.zeros(1, dtype=torch.long), per_graph.cumsum(0)]).tolist()


def test_full_eigenvector_mode_uses_the_lds_attention_path():
    """max_k=None with graphs of more than 16 nodes: rho units span several wave tiles."""
    from oracle import pyg_signnet as O
    from signnet_basisnet_amd import synth
    from signnet_basisnet_amd.pyg import SignNetGNN
    torch.manual_seed(1)
    ctor = (None, None, 64, 1, 2, 2)
    model = SignNetGNN(*ctor, variant="gine")
    data = synth.make_batch(10, seed=5, sizes=[3, 17, 33, 16, 40, 9, 64, 25, 1, 48])
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    yref = O.signnet_gnn(sd, O.make_cfg("gine", *ctor), data, training=False)
    close(model.cuda().eval()(synth.batch_to(data, "cuda:0")), yref, "all-eigenvector forward")


@pytest.mark.parametrize("variant,ctor,max_k,sizes", [
    # k_rho_wide<8, ONE, COLS>: the reference's ZINC default (all eigenvectors), nodes unaligned inside phi's columns
    ("gine", (None, None, 128, 1, 2, 2), None, [37, 27, 33, 31, 17, 21, 26, 9, 12, 64, 1, 2, 5, 40, 24, 16, 18, 46]),
    # k_rho_wide<8, ONE, !COLS>: 16 < max_k < n — per-graph bins, nodes padded to whole tiles
    ("gine", (None, None, 128, 1, 2, 2), 20, [37, 27, 33, 31, 17, 21, 26, 9, 12, 64, 1, 2, 5, 40, 24, 16, 18, 46]),
    ("gine", (None, None, 64, 1, 2, 2), 33, [37, 27, 33, 31, 17, 21, 26, 9, 12, 64, 1, 2, 5, 40, 24, 16, 18, 46]),
    # several encoder layers + the eigenvalue encoder (Alchemy tree at a width the wide kernel takes): !ONE instantiations
    ("alchemy", (6, 4, 64, 12, 2, 2), None, [37, 27, 33, 31, 17, 21, 26, 9, 12, 64, 1, 2, 5, 40]),
    ("alchemy", (6, 4, 128, 12, 2, 2), 24, [37, 27, 33, 9, 12, 64, 1, 2, 5, 40]),
])
def test_rho_with_more_than_16_slots_per_node(variant, ctor, max_k, sizes):
    """The one-image rho kernel (k_rho_wide, round 5): its slot sum against the layer-at-a-time kernels and the fp32 / float64 oracle,
    and the model output, in every instantiation family (columns of phi / per-graph bins, one layer / several + eigenvalue encoder)."""
    from oracle import pyg_signnet as O
    from signnet_basisnet_amd import synth
    from signnet_basisnet_amd.pyg import SignNetGNN
    torch.manual_seed(3)
    model = SignNetGNN(*ctor, variant=variant, max_k=max_k)
    PU.bn_randomize(model, 2)
    data = synth.make_batch(len(sizes), seed=11, sizes=sizes, features="alchemy" if variant == "alchemy" else "zinc")
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    cfg = O.make_cfg(variant, *ctor)
    o32, o64 = {}, {}
    with torch.no_grad():
        y32 = O.signnet_gnn(sd, cfg, data, training=False, max_k=max_k, out=o32)
        y64 = O.signnet_gnn(PU.to_f64(sd), cfg, PU.data_f64(data), training=False, max_k=max_k, out=o64)
    model = model.cuda().eval()
    dd = synth.batch_to(data, "cuda:0")
    with torch.no_grad():
        y = model(dd)
        model.check_last()
        _, st = model(dd, return_stages=True)
    assert not torch.isnan(st["rho_sum_fused"]).any()
    close(st["rho_sum_fused"], o32["rho_sum"], "fused rho slot sum vs oracle", ref64=o64["rho_sum"])
    close(st["rho_sum_fused"], st["rho_sum"], "fused rho slot sum vs layer kernels", ref64=o64["rho_sum"])
    close(y, y32, "model output", ref64=y64)


@pytest.mark.parametrize("variant,ctor,feat,max_k", [
    ("gine", (None, None, 128, 1, 4, 6), "zinc", 16),
    ("gine", (None, None, 64, 1, 4, 6), "zinc", 8),
    ("gine", (None, None, 128, 1, 4, 6), "zinc", None),
    ("alchemy", (6, 4, 108, 12, 3, 4), "alchemy", None),
    ("alchemy", (6, 4, 108, 12, 8, 16), "alchemy", None),       # main_alchemy.py:35 exactly (BASELINE configs[2])
])
def test_vs_oracle_real_widths(variant, ctor, feat, max_k):
    """Reference-sized widths on a seeded synthetic batch: HIP vs the CPU oracle (same weights)."""
    from oracle import pyg_signnet as O
    from signnet_basisnet_amd import synth
    from signnet_basisnet_amd.pyg import SignNetGNN
    torch.manual_seed(0)
    model = SignNetGNN(*ctor, variant=variant, max_k=max_k)
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
    lo, hi = (6, 14) if variant == "alchemy" else (9, 37)
    data = synth.make_batch(16, seed=77, n_lo=lo, n_hi=hi, features=feat)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    cfg = O.make_cfg(variant, *ctor)
    out = {}
    yref = O.signnet_gnn(sd, cfg, data, training=False, max_k=max_k, out=out)
    model = model.cuda().eval()
    dd = synth.batch_to(data, "cuda:0")
    y, st = model(dd, return_stages=True)
    close(st["pos"], out["pos"], "sign_net output")
    close(y, yref, "model output")
    close(st["phi_fused"], out["phi"], "fused phi(x)+phi(-x)")
    close(st["rho_sum_fused"], out["rho_sum"], "fused rho slot-sum")
    close(st["y_gnn_fused"], yref, "fused gnn output")
    close(model(dd), yref, "model output (fused path)")


def test_oversize_graph_is_flagged_and_served_by_the_layer_path():
    """A graph with more than 64 nodes cannot run in the fused kernels: strict mode re-runs it layer by layer,
    the default mode raises at the next call."""
    from oracle import pyg_signnet as O
    from signnet_basisnet_amd import synth
    from signnet_basisnet_amd.pyg import SignNetGNN
    torch.manual_seed(2)
    ctor = (None, None, 32, 1, 2, 2)
    model = SignNetGNN(*ctor, variant="gine", max_k=8)
    data = synth.make_batch(3, seed=4, sizes=[10, 70, 12])
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    yref = O.signnet_gnn(sd, O.make_cfg("gine", *ctor), data, training=False, max_k=8)
    model = model.cuda().eval()
    dd = synth.batch_to(data, "cuda:0")
    assert model.strict, "the safe mode is the default: every input the reference evaluates is evaluated"
    y_mixed = model(dd)
    close(y_mixed, yref, "strict mode (layer-path fallback)")
    # round 6: only the oversize graph goes layer by layer — the rows of the graphs around it are the stage kernels' bits
    good = synth.batch_to(synth.make_batch(3, seed=4, sizes=[10, 70, 12]), "cuda:0")
    from signnet_basisnet_amd import dist as D
    y_first, y_last = model(D.slice_graphs(good, 0, 1)), model(D.slice_graphs(good, 2, 3))
    assert torch.equal(y_mixed[0:1], y_first) and torch.equal(y_mixed[2:3], y_last)
    model.use_fused = False
    model._prep = None
    y_layer = model(dd)
    model.use_fused = True
    model._prep = None
    assert torch.equal(y_mixed[1:2], y_layer[1:2]) and not torch.equal(y_mixed[0:1], y_layer[0:1])
    model.strict = False                        # the serving mode: no host wait
    y = model(dd)                               # flags raised on the device, reported late ...
    torch.cuda.synchronize()
    assert torch.isnan(y).all(), "a batch the fused kernels cannot serve must come back as NaN, never as uninitialised memory"
    with pytest.raises(RuntimeError, match="too large for the fused"):
        model.check_last()
    ok = synth.batch_to(synth.make_batch(3, seed=4, sizes=[10, 20, 12]), "cuda:0")
    y_ok = model(ok)
    model.check_last()                          # a well-formed batch leaves nothing pending
    assert torch.isfinite(y_ok).all()
    # ... at the latest when the module changes mode (an evaluation loop's LAST batch is not lost)
    model(dd)
    with pytest.raises(RuntimeError, match="too large for the fused"):
        model.train()
    model.eval()


def test_feature_id_outside_the_embedding_table_raises_like_nn_embedding():
    """DiscreteEncoder tables have max_num_values rows (500 / 6, elements.py:22); nn.Embedding raises IndexError for an id
    outside.  The HIP kernels never dereference such an id: the affected graph's output is NaN, the others are exact, and
    IndexError is raised (immediately in strict mode and on the layer path, at the next check otherwise)."""
    from signnet_basisnet_amd import ops, synth
    from signnet_basisnet_amd.pyg import SignNetGNN
    torch.manual_seed(0)
    model = SignNetGNN(None, None, 32, 1, 2, 2, variant="gine", max_k=8).cuda().eval()
    model.strict = False
    host = synth.make_batch(5, seed=12)
    good = synth.batch_to(host, "cuda:0")
    with torch.no_grad():
        y_good = model(good).clone()
    model.check_last()
    for what in ("node", "edge"):
        bad_host = synth.make_batch(5, seed=12)
        if what == "node":
            row = int(sum(host.sizes[:2]))                     # first node of graph 2
            bad_host.x[row, 0] = 500                           # table rows: 500 -> valid ids 0..499
        else:
            e = int((bad_host.batch[bad_host.edge_index[0]] == 3).nonzero()[0])
            bad_host.edge_attr[e] = -1
        gi = 2 if what == "node" else 3
        bad = synth.batch_to(bad_host, "cuda:0")
        with torch.no_grad():
            y = model(bad)
        torch.cuda.synchronize()
        assert torch.isnan(y[gi]).all()
        keep = [i for i in range(5) if i != gi]
        assert torch.equal(y[keep], y_good[keep]), "graphs without a bad id are unaffected"
        with pytest.raises(IndexError, match="index out of range in embedding"):
            model.check_last()
        model.strict = True
        with pytest.raises(IndexError, match="index out of range in embedding"):
            model(bad)
        model.strict = False
        with pytest.raises(IndexError, match="index out of range in embedding"):       # layer path (one sync at the end)
            model(bad, return_stages=True)
    # the standalone op: checks itself unless the caller passes a status word
    tab = torch.randn(6, 8, device="cuda:0")
    with pytest.raises(IndexError):
        ops.embedding_sum(torch.tensor([0, 6, 1], device="cuda:0"), [tab])
    st = torch.zeros(1, dtype=torch.int32, device="cuda:0")
    out = ops.embedding_sum(torch.tensor([0, 6, 1], device="cuda:0"), [tab], st)
    assert int(st) == 1 and torch.equal(out[1], torch.zeros(8, device="cuda:0")) and torch.equal(out[0], tab[0])


def test_all_eigenvector_slot_count_from_host_sizes_or_the_plan_report():
    """max_k=None: K = the largest graph.  With host-side sizes on the batch object (`data.sizes`, PyG's `_slice_dict`, a CPU `ptr`) no
    graph size is read back; without them the plan kernel's pinned-memory report is polled; all routes give the same bits, in the
    strict (default) and the serving mode.  Host sizes that understate the batch are caught (strict) and never write out of bounds."""
    import types
    from oracle import pyg_signnet as O
    from signnet_basisnet_amd import pyg, synth
    torch.manual_seed(5)
    ctor = (None, None, 64, 1, 3, 2)
    model = pyg.SignNetGNN(*ctor, variant="gine")
    host = synth.make_batch(12, seed=21, sizes=[9, 37, 20, 33, 12, 27, 16, 30, 10, 36, 22, 17])
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    ref = O.signnet_gnn(sd, O.make_cfg("gine", *ctor), host, training=False, max_k=None)
    model = model.cuda().eval()
    with_sizes = synth.batch_to(host, "cuda:0")
    assert pyg.host_max_nodes(with_sizes) == 37
    bare = types.SimpleNamespace(**{k: v for k, v in vars(with_sizes).items() if k != "sizes"})
    assert pyg.host_max_nodes(bare) is None
    ptr = torch.tensor([0] + list(host.sizes)).cumsum(0)
    sliced = types.SimpleNamespace(**vars(bare), _slice_dict={"x": ptr})
    with_ptr = types.SimpleNamespace(**vars(bare), ptr=ptr)
    assert pyg.host_max_nodes(sliced) == 37 == pyg.host_max_nodes(with_ptr)
    outs = []
    with torch.no_grad():
        for strict in (True, False):
            model.strict = strict
            for d in (with_sizes, bare, sliced, with_ptr):
                outs.append(model(d).clone())
                model.check_last()
    close(outs[0], ref, "all eigenvectors vs oracle")
    for y in outs[1:]:
        assert torch.equal(y, outs[0])
    no_sum = types.SimpleNamespace(**vars(bare), sizes=[20] * 12)           # does not even add up to the batch: ignored
    assert pyg.host_max_nodes(no_sum) is None
    with torch.no_grad():
        assert torch.equal(model(no_sum), outs[0])
    liar = types.SimpleNamespace(**vars(bare), sizes=[22, 22, 22, 22, 22, 23, 23, 23, 23, 23, 22, 22])   # adds up (269), largest 23, not 37
    assert sum(liar.sizes) == sum(host.sizes) and pyg.host_max_nodes(liar) == 23
    model.strict = True
    with pytest.raises(ValueError, match="disagree"), torch.no_grad():
        model(liar)
    model.strict = False
    with torch.no_grad():
        model(liar)                                                           # (graphs of > 23 nodes are dropped, nothing is touched outside the tensors)
    torch.cuda.synchronize()
    with pytest.raises(ValueError, match="disagree"):                         # ... and the serving mode says so when its flags arrive
        model.check_last()                                                    # (round-5 advice: the late path compared nothing with K)
    with torch.no_grad():
        assert torch.equal(model(with_sizes), outs[0])
    model.check_last()


def test_forward_under_hip_graph_capture_never_waits_on_the_host():
    """Round-5 advice: with strict=True (the default) a forward on a capturing stream spun on a pinned ready word no kernel was going to
    write and then called torch.cuda.synchronize() — illegal during capture.  Now a capturing forward leaves the flags on the device
    whatever `strict` says; `check_captured()` reads them after a replay."""
    from signnet_basisnet_amd import synth
    from signnet_basisnet_amd.pyg import SignNetGNN
    torch.manual_seed(0)
    model = SignNetGNN(None, None, 32, 1, 2, 2, variant="gine", max_k=8).cuda().eval()
    assert model.strict
    dd = synth.batch_to(synth.make_batch(6, seed=3), "cuda:0")
    with torch.no_grad():
        y_eager = model(dd).clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            model(dd)                                  # warm-up on the capture's side stream (lazy one-time setup)
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            y_cap = model(dd)
        g.replay()
        torch.cuda.synchronize()
    assert torch.equal(y_cap, y_eager)
    model.check_captured()
    # a bad feature id written into the STATIC input buffer shows up in the captured plan's flags after the next replay
    dd.x.add_(1000)
    g.replay()
    torch.cuda.synchronize()
    with pytest.raises(IndexError):
        model.check_captured()
    dd.x.sub_(1000)


def test_a_forward_that_raises_leaves_no_stale_early_report():
    """Round-5 advice: `_early` was armed inside _forward and consumed in forward() with no try / finally — a forward that raised
    behind build_plan left the report of ITS batch on the module for a later forward to read."""
    import types
    from signnet_basisnet_amd import synth
    from signnet_basisnet_amd.pyg import SignNetGNN
    torch.manual_seed(0)
    model = SignNetGNN(None, None, 32, 1, 2, 2, variant="gine", max_k=8).cuda().eval()
    good = synth.batch_to(synth.make_batch(5, seed=8), "cuda:0")
    with torch.no_grad():
        y0 = model(good).clone()
    bad = types.SimpleNamespace(**vars(good))
    bad.x = good.x.float()                              # the GINE stage refuses float node ids — after the plan was queued and the report armed
    with pytest.raises((ValueError, TypeError, RuntimeError)), torch.no_grad():
        model(bad)
    assert getattr(model, "_early", None) is None
    with torch.no_grad():
        assert torch.equal(model(good), y0)
    model.check_last()


def test_all_eigenvectors_beyond_the_one_launch_plan():
    """max_k=None on a batch of 300 graphs (~6 900 nodes: the five-launch plan, which packs slabs up to 4096 graphs): fused forward vs
    the CPU oracle, strict and serving mode identical."""
    from oracle import pyg_signnet as O
    from signnet_basisnet_amd import synth
    from signnet_basisnet_amd.pyg import SignNetGNN
    torch.manual_seed(4)
    ctor = (None, None, 64, 1, 2, 2)
    model = SignNetGNN(*ctor, variant="gine")
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
    data = synth.make_batch(300, seed=31)
    assert data.batch.numel() > 4096
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    ref = O.signnet_gnn(sd, O.make_cfg("gine", *ctor), data, training=False, max_k=None)
    model = model.cuda().eval()
    dd = synth.batch_to(data, "cuda:0")
    with torch.no_grad():
        y = model(dd)
        model.strict = False
        y2 = model(dd)
        model.check_last()
    close(y, ref, "all eigenvectors, 300 graphs (five-launch plan)")
    assert torch.equal(y, y2)


def test_malformed_batch_comes_back_as_nan_and_raises():
    from signnet_basisnet_amd import synth
    from signnet_basisnet_amd.pyg import SignNetGNN
    torch.manual_seed(0)
    model = SignNetGNN(None, None, 32, 1, 2, 2, variant="gine", max_k=8).cuda().eval()
    host = synth.make_batch(4, seed=3)
    host.batch = host.batch.flip(0).contiguous()               # not sorted
    with pytest.raises(ValueError, match="malformed"), torch.no_grad():      # default (strict): on the spot
        model(synth.batch_to(host, "cuda:0"))
    model.strict = False
    with torch.no_grad():
        y = model(synth.batch_to(host, "cuda:0"))
    torch.cuda.synchronize()
    assert torch.isnan(y).all()
    with pytest.raises(ValueError, match="malformed"):
        model.check_last()


def test_split_gemm_path_is_fp32_accurate():
    """The fused phi / rho stages evaluate fp32 Linears as six bf16 partial products.  Against a float64 run of the
    oracle their error must be in the same class as the fp32-MFMA layer path's (not bf16-class: that would be ~1e-2)."""
    from oracle import pyg_signnet as O
    from signnet_basisnet_amd import synth
    from signnet_basisnet_amd.pyg import SignNetGNN
    torch.manual_seed(3)
    m = SignNetGNN(None, None, 128, 1, 4, 3, variant="gine", max_k=16)
    g = torch.Generator().manual_seed(4)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.running_mean.copy_(torch.randn(mod.running_mean.shape, generator=g) * 0.1)
                mod.running_var.copy_(torch.rand(mod.running_var.shape, generator=g) + 0.5)
    host = synth.make_batch(24, seed=11)
    sd64 = {k: (v.detach().double() if v.is_floating_point() else v.detach().clone()) for k, v in m.state_dict().items()}
    cfg = O.make_cfg("gine", None, None, 128, 1, 4, 3)
    h64 = synth.batch_to(host, "cpu")
    h64.eigen_vectors, h64.eigen_values = h64.eigen_vectors.double(), h64.eigen_values.double()
    out = {}
    with torch.no_grad():
        O.signnet_gnn(sd64, cfg, h64, training=False, max_k=16, out=out)
    ref = out["phi"].double()                                                # phi(x)+phi(-x), [N, K, d]
    m = m.cuda().eval()
    data = synth.batch_to(host, "cuda:0")
    with torch.no_grad():
        _, st = m(data, return_stages=True)
    scale = ref.abs().max().item()
    e_layer = (st["phi"].cpu().double() - ref).abs().max().item() / scale    # fp32-input MFMA, layer by layer
    e_fused = (st["phi_fused"].cpu().double() - ref).abs().max().item() / scale
    assert e_layer < 2e-6 and e_fused < 2e-6, (e_layer, e_fused)
    assert e_fused < 4 * e_layer + 2e-7, (e_layer, e_fused)
    rs = out["rho_sum"].double()
    r_layer = (st["rho_sum"].cpu().double() - rs).abs().max().item() / rs.abs().max().item()
    r_fused = (st["rho_sum_fused"].cpu().double() - rs).abs().max().item() / rs.abs().max().item()
    assert r_layer < 5e-6 and r_fused < 5e-6 and r_fused < 4 * r_layer + 5e-7, (r_layer, r_fused)


def _bn_randomize(m, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.running_mean.copy_(torch.randn(mod.running_mean.shape, generator=g) * 0.1)
                mod.running_var.copy_(torch.rand(mod.running_var.shape, generator=g) + 0.5)


def test_large_batch_fused_equals_layer_path():
    """BASELINE configs[3] per-rank shape and beyond: 640 graphs (N ~ 14.7k nodes: the five-launch plan path, more
    bins than resident workgroups for every stage).  Size-independent checks: the fused stages agree with the
    layer-at-a-time path on every stage, and the output of graph i does not depend on what else is in the batch."""
    from signnet_basisnet_amd import dist as D
    from signnet_basisnet_amd import synth
    from signnet_basisnet_amd.pyg import SignNetGNN
    torch.manual_seed(5)
    m = SignNetGNN(None, None, 128, 1, 4, 6, variant="gine", max_k=16)
    _bn_randomize(m, 6)
    m = m.cuda().eval()
    host = synth.make_batch(640, seed=21)
    data = synth.batch_to(host, "cuda:0")
    with torch.no_grad():
        y_fused = m(data)
        m.check_last()
        y_layer, st = m(data, return_stages=True)
        close(st["phi_fused"], st["phi"], "fused phi vs layer path (640 graphs)")
        close(st["rho_sum_fused"], st["rho_sum"], "fused rho vs layer path (640 graphs)")
        close(y_fused, y_layer, "fused forward vs layer path (640 graphs)")
        # batch independence: a shard run alone gives the same rows
        shard = synth.batch_to(D.shard_batch(host, 1, 5), "cuda:0")          # graphs [128, 256)
        y_shard = m(shard)
        m.check_last()
    close(y_shard, y_fused[128:256], "shard alone vs inside the big batch")


def test_fused_width_not_multiple_of_32():
    """d = 108 (Alchemy's hidden width): 7 output tiles, a half-filled last K block of the split GEMMs."""
    from oracle import pyg_signnet as O
    from signnet_basisnet_amd import synth
    from signnet_basisnet_amd.pyg import SignNetGNN
    torch.manual_seed(8)
    m = SignNetGNN(6, 4, 108, 12, 3, 3, variant="alchemy", max_k=12)
    _bn_randomize(m, 9)
    host = synth.make_batch(20, seed=31, features="alchemy")
    cfg = O.make_cfg("alchemy", 6, 4, 108, 12, 3, 3)
    with torch.no_grad():
        ref = O.signnet_gnn({k: v.detach().clone() for k, v in m.state_dict().items()}, cfg, synth.batch_to(host, "cpu"),
                            training=False, max_k=12)
    m = m.cuda().eval()
    with torch.no_grad():
        y = m(synth.batch_to(host, "cuda:0"))
        m.check_last()
    close(y, ref, "alchemy d=108 fused forward vs oracle")


@pytest.mark.parametrize("name", G.PYG_CASES)
def test_train_mode_forward_with_the_references_own_attention_dropout_draws(name):
    """The one dropout the reference leaves active in training (ScaledDotProductAttention(attn_dropout=0.1), transformer_module.py:
    46,55) against the reference ITSELF: the fixture's `train_do` outputs were produced with torch's own seeded Bernoulli draws, the
    keep-masks it drew are stored beside them, and the HIP train-mode forward (value path and differentiable path) is run on those
    very masks."""
    from signnet_basisnet_amd import synth
    fx = G.load(name)
    model = build(fx)
    model.train()
    data = synth.batch_to(G.as_data(fx.inp), "cuda:0")
    masks = G.attn_keep_masks(fx, device="cuda:0")
    assert len(masks) == len(model.sign_net.rho.transformer_layers) and all(0.05 < float((m == 0).float().mean()) < 0.2 for m in masks)
    model._attn_masks = masks
    try:
        with torch.no_grad():
            y, st = model(data, return_stages=True)
        yg = model(data)                                  # gradients enabled: the differentiable path (autograd.Function per op)
    finally:
        model._attn_masks = None
    for what, got, ref in (("y", y, fx.out["train_do/y"]), ("pos", st["pos"], fx.out["train_do/pos"]), ("y (differentiable path)", yg.detach(), fx.out["train_do/y"])):
        err = (got.cpu() - ref).abs().max().item()
        assert err <= 5e-4 * max(1.0, ref.abs().max().item()) + 5e-5, f"train-mode {what} with the reference's dropout draws: {err:.3e}"
    # and the draws matter: without them the output is a different one
    assert (fx.out["train_do/y"] - fx.out["train/y"]).abs().max().item() > 1e-4


@pytest.mark.parametrize("name", G.PYG_CASES)
def test_train_mode_forward_value_and_running_stats(name):
    """model.train() forward: BatchNorm with batch statistics over the valid rows (reference fixture generated with the
    attention dropout switched off), and the running-statistics side effect of every BatchNorm (momentum 0.1, unbiased
    variance) against what torch's own BatchNorm1d does on the oracle's activations."""
    from oracle import pyg_signnet as O
    from signnet_basisnet_amd import synth
    fx = G.load(name)
    model = build(fx)
    model.train()
    data = synth.batch_to(G.as_data(fx.inp), "cuda:0")
    before = {k: v.detach().clone() for k, v in model.state_dict().items() if k.endswith("running_mean") or k.endswith("running_var")}
    with torch.no_grad():
        y, st = model(data, return_stages=True)
    scale = max(1.0, fx.out["train/y"].abs().max().item())
    err = (y.cpu() - fx.out["train/y"]).abs().max().item()
    assert err <= 5e-4 * scale + 5e-5, f"train-mode y: {err:.3e} (scale {scale:.3e})"      # TOL_BS of test_oracle_golden.py
    perr = (st["pos"].cpu() - fx.out["train/pos"]).abs().max().item()
    assert perr <= 5e-4 * max(1.0, fx.out["train/pos"].abs().max().item()) + 5e-5, f"train-mode pos: {perr:.3e}"
    after = model.state_dict()
    # every BatchNorm that the forward visits moved its running statistics; the rho.out BatchNorm is checked exactly
    unchanged = [k for k, v in before.items() if torch.equal(after[k].cpu(), v.cpu())]
    # only BatchNorms the reference never calls may keep their statistics (rho.pos_encoder, the commented-out eigen_encoder1;
    # eigen_encoder2 IS called by GINESignNetPyG and discarded — its statistics move)
    # ... and the BatchNorm that MaskedMLP / MLP register for their LAST layer but skip when with_final_activation=False
    # (masked_layers.py:60-63, elements.py:63-66): `.nn.norms.1`, `output_encoder.norms.1`
    import re
    never_called = re.compile(r"pos_encoder|eigen_encoder1|\.nn\.norms\.1\.|output_encoder\.norms\.1\.")
    assert all(never_called.search(k) for k in unchanged), [k for k in unchanged if not never_called.search(k)]
    assert len(unchanged) < len(before) // 2
    out = {}
    O.signnet_gnn(fx.sd, G.pyg_cfg(fx), G.as_data(fx.inp), training=True, out=out)
    z = torch.nn.functional.linear(out["rho_sum"], fx.sd["sign_net.rho.out.0.weight"])
    bn = torch.nn.BatchNorm1d(z.shape[1])
    bn.running_mean.copy_(fx.sd["sign_net.rho.out.1.running_mean"])
    bn.running_var.copy_(fx.sd["sign_net.rho.out.1.running_var"])
    bn.train()
    bn(z)
    torch.testing.assert_close(after["sign_net.rho.out.1.running_mean"].cpu(), bn.running_mean, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(after["sign_net.rho.out.1.running_var"].cpu(), bn.running_var, rtol=1e-4, atol=1e-5)
    assert int(after["sign_net.rho.out.1.num_batches_tracked"]) == int(fx.sd["sign_net.rho.out.1.num_batches_tracked"]) + 1
    # and eval() afterwards still runs the fused stages
    model.eval()
    with torch.no_grad():
        model(data)
        model.check_last()


def test_batches_beyond_the_plan_limit_are_served_in_graph_ranges():
    """More graphs than one sn_batch_plan lays out work bins for (6144): the module runs consecutive graph ranges and
    concatenates — checked against the same graphs evaluated in one small batch (the forward never mixes graphs)."""
    import types
    from signnet_basisnet_amd import pyg, synth
    torch.manual_seed(0)
    model = pyg.SignNetGNN(None, None, 32, 1, 2, 2, variant="gine", max_k=8).cuda().eval()
    base = synth.make_batch(205, seed=9)
    reps = 31
    assert 205 * reps > pyg.MAX_FUSED_GRAPHS
    big = types.SimpleNamespace(
        x=base.x.repeat(reps, 1), edge_index=torch.cat([base.edge_index + r * base.num_nodes for r in range(reps)], 1),
        edge_attr=base.edge_attr.repeat(reps), batch=torch.cat([base.batch + r * 205 for r in range(reps)]),
        eigen_values=base.eigen_values.repeat(reps), eigen_vectors=base.eigen_vectors.repeat(reps),
        num_graphs=205 * reps, num_nodes=base.num_nodes * reps)
    big.sizes = list(base.sizes) * reps
    with torch.no_grad():
        y0 = model(synth.batch_to(base, "cuda:0"))
        y = model(synth.batch_to(big, "cuda:0"))
    model.check_last()
    assert y.shape == (205 * reps, 1)
    close(y, y0.repeat(reps, 1), "chunked large batch")


def test_stream_pipeline_returns_the_same_outputs_in_order():
    """serving.StreamPipeline: independent batches round-robin on three streams give the outputs of sequential forwards."""
    from signnet_basisnet_amd import pyg, serving, synth
    torch.manual_seed(0)
    model = pyg.SignNetGNN(None, None, 64, 1, 3, 3, variant="gine", max_k=16).cuda().eval()
    batches = [synth.batch_to(synth.make_batch(40 + 7 * i, seed=60 + i), "cuda:0") for i in range(7)]
    with torch.no_grad():
        ref = [model(b).clone() for b in batches]
    model.check_last()
    outs = serving.StreamPipeline(model, streams=3).map(batches)
    assert len(outs) == len(ref)
    for a, b in zip(outs, ref):
        assert torch.equal(a, b)


def test_overlap_front_mode_is_bit_identical_and_stream_ordered():
    """SignNetGNN.overlap_front: plan + phi and rho of a forward on the module's two side streams, GINE on the caller's stream.  Outputs of a
    loop over different resident batches equal the sequential forwards bit for bit, consumed on the caller's stream without any extra
    synchronisation (each output is read by a kernel queued right behind the forward), with batches dropped as soon as possible;
    strict mode and batches whose inputs were just produced on the current stream (overlap_inputs_ready = False) stay correct."""
    from signnet_basisnet_amd import pyg, synth
    torch.manual_seed(0)
    model = pyg.SignNetGNN(None, None, 128, 1, 4, 6, variant="gine", max_k=16).cuda().eval()
    hosts = [synth.make_batch(128, seed=70 + i, n_lo=9, n_hi=37) for i in range(6)]
    with torch.no_grad():
        ref = [model(synth.batch_to(h, "cuda:0")).clone() for h in hosts]
        model.strict = False
        model.overlap_front = True
        acc = []
        for rep in range(5):
            for i, h in enumerate(hosts):
                b = synth.batch_to(h, "cuda:0")
                torch.cuda.synchronize()                 # the batch is resident (the mode's precondition)
                y = model(b)
                acc.append((i, y * 1.0))                 # consumed on the caller's stream, no sync; `b` and `y` dropped right away
                del b, y
        model.check_last()
        torch.cuda.synchronize()
        for i, y in acc:
            assert torch.equal(y, ref[i])
        assert model._side_streams is not None
        # inputs still in flight on the caller's stream: the side stream waits for it
        model.overlap_inputs_ready = False
        for i, h in enumerate(hosts[:3]):
            b = synth.batch_to(h, "cuda:0")
            b.eigen_vectors = b.eigen_vectors * 1.0      # produced by a kernel on the current stream, not yet complete
            assert torch.equal(model(b), ref[i])
        model.check_last()
        model.strict = True                              # the default mode ignores the flag (one stream, host wait per forward)
        assert torch.equal(model(synth.batch_to(hosts[0], "cuda:0")), ref[0])
    # all-eigenvector mode (max_k None: K = the batch's largest graph, read back on side stream A), Alchemy variant with eigenvalues
    torch.manual_seed(1)
    model = pyg.SignNetGNN(6, 4, 64, 12, 3, 4, variant="alchemy").cuda().eval()
    hosts = [synth.make_batch(96, seed=80 + i, n_lo=6, n_hi=23, features="alchemy") for i in range(4)]
    with torch.no_grad():
        ref = [model(synth.batch_to(h, "cuda:0")).clone() for h in hosts]
        model.strict, model.overlap_front = False, True
        outs = []
        for rep in range(3):
            for i, h in enumerate(hosts):
                b = synth.batch_to(h, "cuda:0")
                torch.cuda.synchronize()
                outs.append((i, model(b) * 1.0))
        model.check_last()
        torch.cuda.synchronize()
    for i, y in outs:
        assert torch.equal(y, ref[i])


@pytest.mark.parametrize("variant,feats,ctor", [("gine", "zinc", (None, None, 32, 1, 2, 2)), ("alchemy", "alchemy", (6, 4, 20, 3, 2, 2))])
@pytest.mark.parametrize("sizes", [[1, 1, 1, 1], [1, 1]], ids=["four single nodes", "two single nodes"])
def test_batch_without_edges(variant, feats, ctor, sizes):
    """A batch of single-node graphs has NO edges (E = 0: empty edge_index / edge_attr, NULL data pointers).  The reference evaluates
    it (PyG's convolutions over an empty edge_index, nn.Embedding of an empty index tensor); so do the fused stages, the layer path
    and the differentiable train-mode forward (found by fuzzing the forward against the oracle, scratch work of round 3)."""
    from oracle import pyg_signnet as O
    from signnet_basisnet_amd import synth
    from signnet_basisnet_amd.pyg import SignNetGNN
    torch.manual_seed(7)
    model = SignNetGNN(*ctor, variant=variant, max_k=4)
    model.attn_dropout = 0.0
    host = synth.make_batch(len(sizes), seed=5, features=feats, sizes=sizes)
    assert host.edge_index.shape[1] == 0
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    ref = O.signnet_gnn(sd, O.make_cfg(variant, *ctor), host, training=False, max_k=4)
    m = model.cuda().eval()
    data = synth.batch_to(host, "cuda:0")
    with torch.no_grad():
        y = m(data)
        y_layer, _ = m(data, return_stages=True)
    assert m._used_fused
    close(y, ref, "fused forward, batch without edges")
    close(y_layer, ref, "layer path, batch without edges")
    # mixed: single-node graphs between ordinary ones
    host2 = synth.make_batch(4, seed=6, features=feats, sizes=[1, 7, 1, 12])
    ref2 = O.signnet_gnn(sd, O.make_cfg(variant, *ctor), host2, training=False, max_k=4)
    with torch.no_grad():
        close(m(synth.batch_to(host2, "cuda:0")), ref2, "fused forward, single-node graphs in the batch")
    # the differentiable path runs (values: train-mode statistics over identical rows are degenerate, only finiteness is asserted)
    m.train()
    yt = m(data)
    yt.sum().backward()
    assert torch.isfinite(yt).all() and all(p.grad is None or torch.isfinite(p.grad).all() for p in m.parameters())
