"""-m gpu: every C-ABI op against a plain torch fp32/fp64 restatement on the same seeded inputs."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from signnet_basisnet_amd import synth  # noqa: E402
from parity_util import close  # noqa: E402


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def batch(dev):
    from signnet_basisnet_amd import ops
    data = synth.make_batch(12, seed=5, sizes=[1, 2, 5, 9, 37, 23, 16, 17, 3, 30, 8, 12])
    d = synth.batch_to(data, dev)
    plan = ops.build_plan(d.batch, d.edge_index, d.num_graphs, 0)
    plan.check()
    return data, d, plan


def dense_adj(data, dtype=torch.float64):
    N = data.batch.numel()
    A = torch.zeros(N, N, dtype=dtype)
    A.index_put_((data.edge_index[1], data.edge_index[0]), torch.ones(data.edge_index.shape[1], dtype=dtype), accumulate=True)
    return A


def test_plan(batch):
    data, d, plan = batch
    n = torch.tensor(data.sizes)
    gp = torch.cat([torch.zeros(1, dtype=torch.long), n.cumsum(0)])
    assert plan.graph_ptr.cpu().tolist() == gp.tolist()
    assert plan.node_graph.cpu().tolist() == data.batch.tolist()
    assert plan.nvalid.cpu().tolist() == n[data.batch].tolist()
    assert plan.evoff.cpu().tolist() == torch.cat([torch.zeros(1, dtype=torch.long), (n * n).cumsum(0)]).tolist()
    st = plan.status.cpu().tolist()
    assert st[0] == 0 and st[1] == max(data.sizes)
    # CSR: in-edges of every node, ordered by edge id
    rowptr, col, eperm = plan.rowptr.cpu(), plan.col.cpu(), plan.eperm.cpu()
    src, dst = data.edge_index
    for i in range(plan.N):
        ids = torch.nonzero(dst == i).flatten()
        seg = slice(int(rowptr[i]), int(rowptr[i + 1]))
        assert eperm[seg].tolist() == ids.tolist()
        assert col[seg].tolist() == src[ids].tolist()


def test_doubled_plan_is_two_disjoint_copies(batch):
    """ops.doubled_plan (sn_plan_double_i32, round 4: one launch instead of eight torch ones): the CSR of two disjoint copies of the
    batch, nodes N .. 2N-1 the second copy."""
    from signnet_basisnet_amd import ops
    data, d, plan = batch
    d2 = ops.doubled_plan(plan)
    E = plan.col.numel()
    assert d2.N == 2 * plan.N and d2.E == 2 * E and d2.rowptr.dtype == torch.int32 and d2.col.dtype == torch.int32
    assert torch.equal(d2.rowptr, torch.cat([plan.rowptr, plan.rowptr[1:] + E]))
    assert torch.equal(d2.col, torch.cat([plan.col, plan.col + plan.N]))
    assert ops.doubled_plan(plan) is d2             # kept on the plan


def test_plan_large_batch_multikernel_path(dev):
    """N > 4096 takes the five-launch path; same contract."""
    from signnet_basisnet_amd import ops
    data = synth.make_batch(260, seed=6)
    d = synth.batch_to(data, dev)
    n = torch.tensor(data.sizes)
    assert int(n.sum()) > 4096
    plan = ops.build_plan(d.batch, d.edge_index, d.num_graphs, 16, bins=True)
    assert plan.check()[0] == 0
    gp = torch.cat([torch.zeros(1, dtype=torch.long), n.cumsum(0)])
    assert plan.graph_ptr.cpu().tolist() == gp.tolist()
    assert plan.nvalid.cpu().tolist() == n.clamp(max=16)[data.batch].tolist()
    rowptr, col, eperm = plan.rowptr.cpu(), plan.col.cpu(), plan.eperm.cpu()
    src, dst = data.edge_index
    order = torch.argsort(dst * (dst.numel() + 1) + torch.arange(dst.numel()))      # by dst, then edge id
    assert eperm.tolist() == order.tolist() and col.tolist() == src[order].tolist()
    nb, err, rows, ncol = plan.bins.meta.cpu().tolist()[:4]
    assert err == 0 and rows == int((n * n.clamp(max=16)).sum()) and 0 < ncol <= 260 and nb <= plan.bins.phi_max_bins


def _reference_columns(sizes, kmax):
    """The planner's column packing restated on the host: graphs grouped by size (ascending id inside a size class); a column takes
    the largest size still available, then repeatedly the largest size that fits the rows left (capacity 64 rows, at most 8 members);
    a column's height is its first (largest) member's slot count."""
    slots = (lambda n: -kmax) if kmax < 0 else (lambda n: min(n, kmax) if kmax > 0 else n)
    classes = {}
    for g, n in enumerate(sizes):
        if 0 < n <= 64:
            classes.setdefault(n, []).append(g)
    cols = []
    while classes:
        s = max(classes)
        cap, members, off = 64, [], 0
        cls = s
        while True:
            g = classes[cls].pop(0)
            if not classes[cls]:
                del classes[cls]
            members.append((g, off))
            off += cls
            cap -= cls
            fit = [c for c in classes if c <= cap]
            if len(members) >= 8 or cap <= 0 or not fit:
                break
            cls = max(fit)
        cols.append((members, slots(s)))
    return cols


@pytest.mark.parametrize("B,kmax,seed,lo,hi", [(128, 16, 1236, 9, 37), (256, 0, 3, 6, 14), (37, 8, 5, 1, 64), (1500, 4, 7, 1, 5),
                                                (2100, -3, 9, 1, 3), (1, 16, 2, 9, 9)])
def test_plan_phi_columns_match_the_host_restatement(dev, B, kmax, seed, lo, hi):
    """The work bins of the fused phi stage (sn_batch_plan with bins): member graphs, row offsets, first bins and the bin -> column
    map equal the host restatement of the packing, on the one-launch path (N <= 4096) and the five-launch path, including more than
    1024 graphs (several rounds of the planner's record scan) and the DGL mode (kmax < 0: a fixed slot count)."""
    from signnet_basisnet_amd import ops
    data = synth.make_batch(B, seed=seed, n_lo=lo, n_hi=hi)
    d = synth.batch_to(data, dev)
    plan = ops.build_plan(d.batch, d.edge_index, d.num_graphs, kmax, bins=True, columns=True)
    assert plan.check()[0] == 0
    cols = _reference_columns(list(data.sizes), kmax)
    nb, err, rows, ncol = plan.bins.meta.cpu().tolist()[:4]
    if kmax == 0 and B <= 4096:
        # all eigenvectors: slab-level bins, no columns (test_plan_bin_member_records)
        assert err == 0 and ncol == 0 and nb == len(_reference_slab_bins(list(data.sizes)))
        return
    assert err == 0 and ncol == len(cols)
    mem = plan.bins.phi_col_mem.cpu().view(-1, 8)[:ncol].tolist()
    off = plan.bins.phi_col_off.cpu().view(-1, 8)[:ncol].tolist()
    bin0 = plan.bins.phi_col_bin0.cpu()[:ncol + 1].tolist()
    exp_bin0 = [0]
    for c, (members, h) in enumerate(cols):
        assert mem[c] == [g for g, _ in members] + [-1] * (8 - len(members)), c
        assert off[c][:len(members)] == [o for _, o in members], c
        exp_bin0.append(exp_bin0[-1] + h)
    assert bin0 == exp_bin0 and nb == exp_bin0[-1]
    slots = (lambda n: -kmax) if kmax < 0 else (lambda n: min(n, kmax) if kmax > 0 else n)
    assert rows == sum(n * slots(n) for n in data.sizes if 0 < n <= 64)
    bc = plan.bins.phi_bin_col.cpu()[:nb].tolist()
    assert bc == [c for c, (_, h) in enumerate(cols) for _ in range(h)]


def _reference_slab_bins(sizes):
    """The planner's slab-level packing of the all-eigenvector mode restated on the host (csrc/plan.hip, plan_bins_block): a graph of n
    nodes is n slabs of n rows; best-fit-decreasing over the size classes, <= 8 slabs and <= 64 rows per bin, one chain link per bin
    PATTERN (repeated while every class of the pattern still has its copies); slab q of a class = (q // n)-th graph of the class in id
    order, index q % n."""
    classes = {}
    for g, n in enumerate(sizes):
        if 0 < n <= 64:
            classes.setdefault(n, []).append(g)
    cnt = {s: s * len(gs) for s, gs in classes.items()}
    used = {s: 0 for s in classes}
    bins, npat = [], 0
    while cnt:
        single = npat >= 160 - 64
        mult, members, cap = {}, [], 64
        cls = max(cnt)
        while True:
            copies = mult.get(cls, 0)
            if cnt[cls] - copies > 0 and cls <= cap and len(members) < 8:
                members.append((cls, 64 - cap, copies))
                mult[cls] = copies + 1
                cap -= cls
                if single:
                    break
                continue
            limit = min(cap, cls - 1)
            if limit <= 0 or len(members) >= 8:
                break
            cand = [c for c in cnt if c <= limit]
            if not cand:
                break
            cls = max(cand)
        r = min(cnt[c] // m for c, m in mult.items())
        assert r >= 1
        for rep in range(r):
            b = []
            for c, off, t in members:
                rank, idx = divmod(used[c] + t + rep * mult[c], c)
                b.append((classes[c][rank], idx, off, c))
            bins.append(b)
        for c, m in mult.items():
            cnt[c] -= r * m
            used[c] += r * m
            if cnt[c] == 0:
                del cnt[c]
        npat += 1
    return bins


def _decode_bin_records(plan):
    nb = int(plan.bins.meta[7])
    rec = plan.bins.phi_bin_mem.cpu()[:nb * 16].view(nb, 8, 2).tolist()
    out = []
    for b in rec:
        out.append([(w0 & 8191, (w0 >> 13) & 63, (w0 >> 19) & 63, ((w0 >> 25) & 63) + 1, g0) for w0, g0 in b if w0 >= 0])
    return out


@pytest.mark.parametrize("B,kmax,seed,lo,hi", [(128, 0, 1236, 9, 37), (128, 16, 1236, 9, 37), (256, 0, 3, 6, 14), (40, 0, 9, 1, 64),
                                                (37, 8, 5, 1, 64), (300, 0, 11, 1, 9), (1500, 4, 7, 1, 5), (2100, -3, 9, 1, 3),
                                                (1500, 0, 13, 1, 3), (1, 0, 2, 9, 9), (300, 0, 17, 9, 37), (4200, 0, 19, 1, 2)])
def test_plan_bin_member_records(dev, B, kmax, seed, lo, hi):
    """The per-bin member records the stage kernels walk (sn_plan_bins.phi_bin_mem): every (graph, index) slab exactly once, members
    of a bin disjoint inside 64 rows, word 1 = the graph's first node; with kmax != 0 (and on the five-launch plan) the bins of the
    columns, with all eigenvectors on the one-launch plan the slab-level packing — equal to its host restatement, at >= 95 % fill on
    the ZINC-like batch where the columns reach 92 %."""
    from signnet_basisnet_amd import ops
    data = synth.make_batch(B, seed=seed, n_lo=lo, n_hi=hi)
    d = synth.batch_to(data, dev)
    plan = ops.build_plan(d.batch, d.edge_index, d.num_graphs, kmax, bins=True, columns=True)
    assert plan.check()[0] == 0
    sizes = list(data.sizes)
    gp = [0]
    for n in sizes:
        gp.append(gp[-1] + n)
    slots = (lambda n: -kmax) if kmax < 0 else (lambda n: min(n, kmax) if kmax > 0 else n)
    meta = plan.bins.meta.cpu().tolist()
    got = _decode_bin_records(plan)
    seen = set()
    for b in got:
        spans = sorted((off, off + n) for _, _, off, n, _ in b)
        assert b and spans[-1][1] <= 64 and all(spans[i][1] <= spans[i + 1][0] for i in range(len(spans) - 1))
        for g, idx, off, n, g0 in b:
            assert n == sizes[g] and g0 == gp[g] and idx < slots(n) and (g, idx) not in seen
            seen.add((g, idx))
    assert seen == {(g, i) for g, n in enumerate(sizes) if 0 < n <= 64 for i in range(slots(n))}
    if kmax == 0 and B <= 4096:              # slab-level packing (the five-launch plan too, up to 4096 graphs)
        want = _reference_slab_bins(sizes)
        assert len(got) == len(want) == meta[0] and meta[3] == 0
        for b, (gb, wb) in enumerate(zip(got, want)):
            assert [(g, idx, off, n) for g, idx, off, n, _ in gb] == wb, b
        if (lo, hi) == (9, 37):
            assert meta[2] / (64 * len(got)) >= 0.95 > meta[2] / (64 * sum(h for _, h in _reference_columns(sizes, 0)))
    else:
        assert len(got) == meta[0]             # the columns' bins
        cols = _reference_columns(sizes, kmax)
        b = 0
        for members, h in cols:
            for j in range(h):
                assert [(g, idx, off) for g, idx, off, _, _ in got[b]] == [(g, j, off) for g, off in members if j < slots(sizes[g])], (b, j)
                b += 1


def test_plan_early_report(dev):
    """sn_batch_plan_ex: the plan kernel writes the batch's flags to pinned host memory itself (what the module's strict mode polls
    instead of waiting for the end of the forward)."""
    from signnet_basisnet_amd import ops
    data = synth.make_batch(12, seed=4, sizes=[10, 30, 12, 64, 9, 20, 33, 1, 17, 25, 40, 5])
    d = synth.batch_to(data, dev)
    assert ops.early_supported(d.batch.numel(), d.edge_index.shape[1], d.num_graphs) and not ops.early_supported(5000, 10, 10)
    E = ops.EarlyReport
    rep = E().arm(d.x, 28, d.edge_attr, 4, 192)
    plan = ops.build_plan(d.batch, d.edge_index, d.num_graphs, 0, bins=True, early=rep)
    fl = rep.wait()
    st = plan.check()
    assert fl[E.ERR] == 0 and fl[E.NMAX] == 64 == st[1] and fl[E.DEGMAX] == st[2] and fl[E.EDGES] == 0 and fl[E.PHI] == 0 and fl[E.RHO] == 0 and fl[E.IDS] == 0
    assert rep.wait_nmax() == 64
    rep.release()
    # a node id outside its table / an edge id outside its table / a graph with more in-edges than the limit given / an oversize graph
    for what in ("node", "edge", "edges", "nodes", "unsorted", "noids"):
        x, ea, batch, lim, dd = d.x, d.edge_attr, d.batch, 192, d
        if what == "node":
            x = d.x.clone(); x[17, 0] = 28
        if what == "edge":
            ea = d.edge_attr.clone(); ea[5] = -1
        if what == "edges":
            lim = 40
        if what == "nodes":
            dd = synth.batch_to(synth.make_batch(3, seed=4, sizes=[10, 70, 12]), dev)
            x, ea, batch = dd.x, dd.edge_attr, dd.batch
        if what == "unsorted":
            batch = d.batch.flip(0).contiguous()
        rep = E().arm(None if what == "noids" else x, 28, None if what == "noids" else ea, 4, lim)
        ops.build_plan(batch, dd.edge_index, dd.num_graphs, 16, bins=True, early=rep)
        fl = rep.wait()
        rep.release()
        assert bool(fl[E.ERR]) == (what == "unsorted"), what
        if what != "unsorted":               # (a malformed batch has no meaningful graph boundaries: only its error word counts)
            assert bool(fl[E.IDS]) == (what in ("node", "edge")), what
            assert bool(fl[E.EDGES]) == (what == "edges"), what
            assert bool(fl[E.PHI]) == (what == "nodes"), what
    # without the bin planner (one workgroup): every done word still arrives
    rep = E().arm(None, 0, None, 0, 192)
    ops.build_plan(d.batch, d.edge_index, d.num_graphs, 16, bins=False, early=rep)
    fl = rep.wait()
    rep.release()
    assert fl[E.NMAX] == 64 and fl[E.ERR] == 0


def test_plan_kmax_and_errors(dev):
    from signnet_basisnet_amd import ops
    data = synth.make_batch(4, seed=2, sizes=[3, 20, 7, 12])
    d = synth.batch_to(data, dev)
    plan = ops.build_plan(d.batch, d.edge_index, 4, 8)
    assert plan.nvalid.cpu().tolist() == torch.tensor(data.sizes).clamp(max=8)[data.batch].tolist()
    bad = d.batch.flip(0).contiguous()
    with pytest.raises(ValueError):
        ops.build_plan(bad, d.edge_index, 4, 0).check()
    ei = d.edge_index.clone()
    ei[0, 0] = data.batch.numel() - 1      # edge from the last graph into the first
    with pytest.raises(ValueError):
        ops.build_plan(d.batch, ei, 4, 0).check()


def test_pack_eig(batch):
    from oracle import pyg_signnet as O
    from signnet_basisnet_amd import ops
    data, d, plan = batch
    K = max(data.sizes)
    x0, s0 = ops.pack_eig(plan, d.eigen_vectors, d.eigen_values, K, True)
    es, ev, mask = O.to_dense_list_evd(data.eigen_values, data.eigen_vectors, data.batch)
    assert torch.equal(x0.cpu(), ev) and torch.equal(s0.cpu(), es)     # pure data movement: bit exact


@pytest.mark.parametrize("F", [1, 37, 64, 131, 400, 1300, 16 * 128])   # every column-group width of the gather kernel, float and float4
@pytest.mark.parametrize("slab", [False, True])
def test_gin_aggregate(batch, dev, F, slab):
    from oracle import pyg_signnet as O
    from signnet_basisnet_amd import ops
    data, d, plan = batch
    g = torch.Generator().manual_seed(F)
    x = torch.randn(plan.N, F, generator=g)
    eps = torch.tensor([0.37])
    ref = O.gin_aggregate(x, data.edge_index, eps, node_dim=0)
    out = ops.gin_aggregate(x.to(dev), plan, eps.to(dev), slab=slab)
    assert torch.equal(out.cpu(), ref)                       # same summation order -> bit exact
    outn = ops.gin_aggregate(x.to(dev), plan, eps.to(dev), negate=True, slab=slab)
    assert torch.equal(outn.cpu(), O.gin_aggregate(-x, data.edge_index, eps, node_dim=0))
    # independent check: dense fp64 adjacency product
    ref64 = dense_adj(data) @ x.double() + (1 + eps.double()) * x.double()
    torch.testing.assert_close(out.cpu().double(), ref64, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("C", [20, 128])
def test_gine_aggregate(batch, dev, C):
    from oracle import pyg_signnet as O
    from signnet_basisnet_amd import ops
    data, d, plan = batch
    g = torch.Generator().manual_seed(C)
    x, e = torch.randn(plan.N, C, generator=g), torch.randn(plan.E, C, generator=g)
    eps = torch.tensor([-0.2])
    out = ops.gine_aggregate(x.to(dev), e.to(dev), plan, eps.to(dev))
    assert torch.equal(out.cpu(), O.gine_aggregate(x, data.edge_index, e, eps))


@pytest.mark.parametrize("d_in,d_out", [(1, 1), (1, 128), (128, 128), (108, 108), (6, 44), (256, 128), (95, 4), (300, 70), (2048, 20), (600, 33)])
def test_masked_linear(batch, dev, d_in, d_out):
    from signnet_basisnet_amd import ops
    data, d, plan = batch
    K = 19
    nv = torch.tensor(data.sizes).clamp(max=K)[data.batch]
    mask = torch.arange(K)[None, :] < nv[:, None]
    g = torch.Generator().manual_seed(d_in * 1000 + d_out)
    x = torch.randn(plan.N, K, d_in, generator=g)
    W, b = torch.randn(d_out, d_in, generator=g) / d_in ** 0.5, torch.randn(d_out, generator=g)
    sc, sh = torch.rand(d_out, generator=g) + 0.5, torch.randn(d_out, generator=g)
    res = torch.randn(plan.N, K, d_out, generator=g)
    pl = ops.PackedLinear(ops.pack_weight(W.to(dev)), d_out, d_in, b.to(dev))
    nvd = nv.to(dev).int()
    # plain
    y = ops.masked_linear(x.to(dev), pl, use_bias=False)
    ref = (x.double() @ W.double().T)
    torch.testing.assert_close(y.cpu().double(), ref, rtol=1e-5, atol=1e-5)
    # full epilogue, PyG order: bias -> mask -> affine -> relu -> residual
    y = ops.masked_linear(x.to(dev), pl, nvd, K, scale=sc.to(dev), shift=sh.to(dev), relu=True, residual=res.to(dev))
    ref = torch.relu((x.double() @ W.double().T + b.double()) * sc.double() + sh.double()) + res.double()
    ref = ref * mask.unsqueeze(-1)
    torch.testing.assert_close(y.cpu().double(), ref, rtol=1e-5, atol=1e-5)
    # DGL order: bias -> relu -> affine
    y = ops.masked_linear(x.to(dev), pl, scale=sc.to(dev), shift=sh.to(dev), relu_pre=True)
    ref = torch.relu(x.double() @ W.double().T + b.double()) * sc.double() + sh.double()
    torch.testing.assert_close(y.cpu().double(), ref, rtol=1e-5, atol=1e-5)
    # fp32 torch agreement (the tolerance north_star states: 1e-5 relative)
    y32 = F.linear(x, W)
    yk = ops.masked_linear(x.to(dev), pl, use_bias=False).cpu()
    assert (yk - y32).abs().max() <= 1e-5 * max(1.0, y32.abs().max().item())


@pytest.mark.parametrize("d_in,d_out", [(128, 128), (64, 128), (256, 96), (128, 256), (200, 44)])
def test_masked_linear_lds_resident_weight_path_is_bit_identical(dev, d_in, d_out):
    """From 32 rows on sn_masked_linear_f32 stages the packed weight once per workgroup in LDS (k_linear_lds): the same products in
    the same order as the per-wave kernel — the first rows of a long input equal, bit for bit, the result of a short launch over
    those rows alone (which takes the other kernel) — with and without the full epilogue and the slot mask; and against float64."""
    from signnet_basisnet_amd import ops
    K, N = 16, 700                                    # 11200 rows; the short launch takes rows [0, 16)
    g = torch.Generator().manual_seed(d_in + 7 * d_out)
    x = torch.randn(N, K, d_in, generator=g)
    W, b = torch.randn(d_out, d_in, generator=g) / d_in ** 0.5, torch.randn(d_out, generator=g)
    sc, sh = torch.rand(d_out, generator=g) + 0.5, torch.randn(d_out, generator=g)
    res = torch.randn(N, K, d_out, generator=g)
    nv = torch.randint(0, K + 1, (N,), generator=g)
    nv[4:8] = 0                                       # a whole 64-row block without a valid row
    nv[0] = 7
    pl = ops.PackedLinear(ops.pack_weight(W.to(dev)), d_out, d_in, b.to(dev))
    xd, nvd, resd = x.to(dev), nv.to(dev).int(), res.to(dev)
    n0 = 1
    for kw in (dict(use_bias=False), dict(nvalid=nvd, K=K, scale=sc.to(dev), shift=sh.to(dev), relu=True, residual=resd),
               dict(nvalid=nvd, K=K)):
        kw_short = dict(kw)
        if "nvalid" in kw:
            kw_short["nvalid"] = nvd[:n0].contiguous()
        if "residual" in kw:
            kw_short["residual"] = resd[:n0].contiguous()
        full = ops.masked_linear(xd, pl, **kw)
        short = ops.masked_linear(xd[:n0].contiguous(), pl, **kw_short)
        assert torch.equal(full[:n0], short), kw.keys()
    mask = (torch.arange(K)[None, :] < nv[:, None]).unsqueeze(-1)
    y = ops.masked_linear(xd, pl, nvd, K, scale=sc.to(dev), shift=sh.to(dev), relu=True, residual=resd)
    ref = (torch.relu((x.double() @ W.double().T + b.double()) * sc.double() + sh.double()) + res.double()) * mask
    torch.testing.assert_close(y.cpu().double(), ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("N,K,d_in,d_out,masked", [(3000, 16, 128, 128, True), (800, 8, 64, 96, True), (5000, 1, 128, 128, False),
                                                     (300, 16, 256, 128, True), (40, 16, 128, 128, True), (600, 16, 128, 200, True),
                                                     (700, 8, 1, 64, True), (90, 9, 12, 12, True), (1, 16, 128, 128, True),
                                                     (2, 8, 16, 4, False)])
def test_linear_bn_train_one_pass(dev, N, K, d_in, d_out, masked):
    """sn_linear_bn_train_f32 (Linear + train-mode BatchNorm statistics from the Linear's accumulators): z equals sn_masked_linear_f32
    bit for bit; mean / var / count / folded scale, shift and the running statistics agree with sn_bn_train_stats_f32 on z and with
    float64 moments of the valid rows."""
    from signnet_basisnet_amd import ops
    g = torch.Generator().manual_seed(N + d_in)
    x = torch.randn(N, K, d_in, generator=g) * 1.5 + 0.3
    W, b = torch.randn(d_out, d_in, generator=g) / d_in ** 0.5, torch.randn(d_out, generator=g) * 3
    nv = torch.randint(0, K + 1, (N,), generator=g) if masked else None
    nvd = None if nv is None else nv.to(dev).int()
    pl = ops.PackedLinear(ops.pack_weight(W.to(dev)), d_out, d_in, b.to(dev))
    bn1, bn2 = (torch.nn.BatchNorm1d(d_out).to(dev).train() for _ in range(2))
    with torch.no_grad():
        for bn in (bn1, bn2):
            bn.weight.copy_(torch.rand(d_out, generator=torch.Generator().manual_seed(1)) + 0.5)
            bn.bias.copy_(torch.randn(d_out, generator=torch.Generator().manual_seed(2)))
    z, mean, var, rstd, scale, shift, count = ops.linear_bn_train(x.to(dev), pl, bn1, nvd, K if masked else 0)
    z_ref = ops.masked_linear(x.to(dev), pl, nvd, K if masked else 0)
    assert torch.equal(z, z_ref)
    m2, v2, r2, sc2, sh2, c2 = ops.bn_train_stats(z_ref, bn2, nvd, K if masked else 0)
    assert float(count) == float(c2)
    rows = z.cpu().double().reshape(N * K, d_out)
    if masked:
        rows = rows[(torch.arange(K)[None, :] < nv[:, None]).reshape(-1)]
    assert float(count) == rows.shape[0]
    mu, va = rows.mean(0), rows.var(0, unbiased=False)
    close(mean, mu, "mean", rel=2e-6)
    close(var, va, "biased variance", rel=1e-5)
    close(rstd, 1.0 / torch.sqrt(va + bn1.eps), "rstd", rel=1e-5)
    close(scale, sc2, "scale vs the two-call form")
    close(shift, sh2, "shift vs the two-call form")
    close(bn1.running_mean, bn2.running_mean, "running mean")
    close(bn1.running_var, bn2.running_var, "running var")
    assert int(bn1.num_batches_tracked) == 1


@pytest.mark.parametrize("rows,cols", [(128, 128), (44, 16), (7, 200), (2, 3)])
def test_pack_weight_transposed_in_place(dev, rows, cols):
    """sn_pack_weight_t_f32 (the backward's dX = dY W reads the forward weight in place) == sn_pack_weight_f32 of the transposed copy."""
    from signnet_basisnet_amd import ops
    W = torch.randn(rows, cols, generator=torch.Generator().manual_seed(rows + cols)).to(dev)
    assert torch.equal(ops.pack_weight_t(W), ops.pack_weight(W.t().contiguous()))
    Wv = torch.randn(rows, cols + 3, generator=torch.Generator().manual_seed(1)).to(dev)[:, :cols]      # a strided view
    assert torch.equal(ops.pack_weight_t(Wv), ops.pack_weight(Wv.t().contiguous()))


def test_colstats_affine(batch, dev):
    from signnet_basisnet_amd import ops
    data, d, plan = batch
    K, C = 11, 44
    nv = torch.tensor(data.sizes).clamp(max=K)[data.batch]
    mask = torch.arange(K)[None, :] < nv[:, None]
    x = torch.randn(plan.N, K, C, generator=torch.Generator().manual_seed(3)) * 2 + 5
    mean, var, cnt = ops.masked_colstats(x.to(dev), nv.to(dev).int(), K)
    rows = x[mask].double()
    assert cnt.item() == rows.shape[0]
    torch.testing.assert_close(mean.cpu().double(), rows.mean(0), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(var.cpu().double(), rows.var(0, unbiased=False), rtol=1e-5, atol=1e-5)
    sc, sh = torch.rand(C) + 0.5, torch.randn(C)
    y = ops.masked_affine(x.to(dev), nv.to(dev).int(), K, scale=sc.to(dev), shift=sh.to(dev), relu=True)
    ref = torch.relu(x * sc + sh) * mask.unsqueeze(-1)
    torch.testing.assert_close(y.cpu(), ref, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("C", [12, 108, 128])
def test_layernorm(batch, dev, C):
    from signnet_basisnet_amd import ops
    data, d, plan = batch
    K = 9
    nv = torch.tensor(data.sizes).clamp(max=K)[data.batch]
    mask = torch.arange(K)[None, :] < nv[:, None]
    g = torch.Generator().manual_seed(C)
    x, r = torch.randn(plan.N, K, C, generator=g), torch.randn(plan.N, K, C, generator=g)
    w, b = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    y = ops.masked_layernorm(x.to(dev), r.to(dev), w.to(dev), b.to(dev), 1e-6, nv.to(dev).int(), K)
    ref = F.layer_norm((x + r).double(), (C,), w.double(), b.double(), 1e-6) * mask.unsqueeze(-1)
    torch.testing.assert_close(y.cpu().double(), ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("K,D", [(9, 12), (16, 128), (37, 108)])
def test_set_attention(batch, dev, K, D):
    from signnet_basisnet_amd import ops
    data, d, plan = batch
    H, dk = 4, D // 4
    nv = torch.tensor(data.sizes).clamp(max=K)[data.batch]
    mask = (torch.arange(K)[None, :] < nv[:, None])
    g = torch.Generator().manual_seed(K * D)
    q, k, v = (torch.randn(plan.N, K, D, generator=g) * mask.unsqueeze(-1) for _ in range(3))
    out = ops.set_attention(q.to(dev), k.to(dev), v.to(dev), plan.N, K, H, nv.to(dev).int())
    pair = (mask.unsqueeze(1) * mask.unsqueeze(2)).unsqueeze(1)
    sp = lambda t: t.double().view(plan.N, K, H, dk).transpose(1, 2)
    att = torch.matmul(sp(q) / dk ** 0.5, sp(k).transpose(2, 3)).masked_fill(pair == 0, -1e10)
    att = torch.softmax(att, -1) * pair
    ref = torch.matmul(att, sp(v)).transpose(1, 2).reshape(plan.N, K, D)
    torch.testing.assert_close(out.cpu().double(), ref, rtol=1e-5, atol=1e-5)


def test_small_ops(batch, dev):
    from signnet_basisnet_amd import ops
    data, d, plan = batch
    g = torch.Generator().manual_seed(9)
    x = torch.randn(plan.N, 7, 20, generator=g)
    torch.testing.assert_close(ops.slot_sum(x.to(dev), plan.N, 7).cpu(), x.sum(1), rtol=1e-6, atol=1e-6)
    tabs = [torch.randn(30, 20, generator=g) for _ in range(3)]
    idx = torch.randint(0, 30, (plan.N, 3), generator=g)
    ref = sum(tabs[f][idx[:, f]] for f in range(3))
    torch.testing.assert_close(ops.embedding_sum(idx.to(dev), [t.to(dev) for t in tabs]).cpu(), ref, rtol=1e-6, atol=1e-6)
    assert torch.equal(ops.embedding_sum(idx[:, 0].contiguous().to(dev), [tabs[0].to(dev)]).cpu(), tabs[0][idx[:, 0]])
    h = torch.randn(plan.N, 20, generator=g)
    ref = torch.zeros(plan.B, 20).index_add_(0, data.batch, h)
    torch.testing.assert_close(ops.segment_pool(h.to(dev), plan).cpu(), ref, rtol=1e-6, atol=1e-6)
    cnt = torch.tensor(data.sizes).float().unsqueeze(1)
    torch.testing.assert_close(ops.segment_pool(h.to(dev), plan, "mean").cpu(), ref / cnt, rtol=1e-6, atol=1e-6)


def test_cpu_tensors_are_rejected():
    from signnet_basisnet_amd import ops
    with pytest.raises(RuntimeError):
        ops.pack_weight(torch.randn(4, 4))


# ---------------------------------------------------------------------------------------------- split-packed linears
def _unpack_split(buf, d_out, d_in):
    """Host-side inverse of sn_pack_split_f32: (W_h, W_m, W_l) as float32 [d_out_pad, d_in_pad32] and the three
    epilogue vectors, from the documented layout (include/signnet_hip.h, csrc/fused_common.hpp)."""
    import numpy as np
    nto, nkb = (d_out + 15) // 16, (d_in + 31) // 32
    nfe = 3 * nkb + 3
    raw = buf.cpu().numpy().reshape(nto, nfe, 1024)
    planes = np.zeros((3, nto * 16, nkb * 32), dtype=np.float32)
    vecs = np.zeros((3, nto * 16), dtype=np.float32)
    for ot in range(nto):
        for kb in range(nkb):
            for pl in range(3):
                frag = raw[ot, kb * 3 + pl].view(np.uint16).reshape(64, 8)
                f32 = (frag.astype(np.uint32) << 16).view(np.float32)
                for lane in range(64):
                    o, g = 16 * ot + (lane & 15), lane >> 4
                    for s in range(8):
                        planes[pl, o, 32 * kb + 16 * (s >> 2) + 4 * g + (s & 3)] = f32[lane, s]
        for e in range(3):
            frag = raw[ot, 3 * nkb + e].view(np.float32).reshape(64, 4)
            for lane in range(64):
                vecs[e, 16 * ot + 4 * (lane >> 4):16 * ot + 4 * (lane >> 4) + 4] = frag[lane]
    return planes, vecs


@pytest.mark.parametrize("d_out,d_in", [(128, 128), (108, 108), (44, 16), (16, 40)])
def test_pack_split_is_an_exact_three_way_split(d_out, d_in):
    """h + m + l reproduces every fp32 weight bit for bit, each piece is a bf16 value (8 significant bits), the
    layout is the documented one, padding is zero and the epilogue vectors ride along unchanged."""
    import numpy as np
    from signnet_basisnet_amd import ops
    g = torch.Generator().manual_seed(d_out * 131 + d_in)
    W = (torch.randn(d_out, d_in, generator=g) * torch.logspace(-6, 3, d_in)).cuda()   # wide dynamic range
    e0, e2 = torch.randn(d_out, generator=g).cuda(), torch.randn(d_out, generator=g).cuda()
    buf = ops.pack_split(W, e0, None, e2)
    torch.cuda.synchronize()
    assert buf.numel() == ((d_out + 15) // 16) * (3 * ((d_in + 31) // 32) + 3) * 1024
    planes, vecs = _unpack_split(buf, d_out, d_in)
    Wn = W.cpu().numpy()
    rec = (planes[0].astype(np.float64) + planes[1] + planes[2])[:d_out, :d_in]
    assert np.array_equal(rec.astype(np.float32), Wn) and np.array_equal(rec, Wn.astype(np.float64))
    assert not planes[:, d_out:, :].any() and not planes[:, :, d_in:].any()
    assert np.all(np.abs(planes[1]) <= np.abs(planes[0]) * 2.0 ** -7 + 1e-45)        # m < one bf16 ulp of h
    assert np.array_equal(vecs[0, :d_out], e0.cpu().numpy()) and not vecs[1].any()
    assert np.array_equal(vecs[2, :d_out], e2.cpu().numpy()) and not vecs[:, d_out:].any()
