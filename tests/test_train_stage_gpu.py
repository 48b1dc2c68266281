"""-m gpu: the training stage kernels of csrc/train.hip (one pass per direction for a Linear -> BatchNorm(train) -> ReLU link) against
float64 torch.autograd of the same composition, with the float32 CPU evaluation beside it: the HIP result may be as far from the
exact value as the reference's own fp32 arithmetic is (attribution, tests/parity_util.py), never more."""
import pytest
import torch

import parity_util as PU

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _mask(nvalid, K, R):
    if nvalid is None:
        return torch.ones(R, dtype=torch.bool)
    return (torch.arange(K)[None, :] < nvalid[:, None]).reshape(-1)


def _bn(z, m, gamma, beta, eps=1e-5):
    zv = z[m]
    mu, var = zv.mean(0), zv.var(0, unbiased=False)
    out = torch.zeros_like(z)
    out[m] = (zv - mu) / torch.sqrt(var + eps) * gamma + beta
    return out, mu, zv.var(0, unbiased=True)


def ref_mlp2(x, p, masks, G, residual, relu_out, has_bn2=True):
    """x [G*R, d_in] -> y; the reference composition per group (masked_layers.py:54-64 + sign_net.py:39-42)."""
    R = x.shape[0] // G
    ys, stats = [], []
    for g in range(G):
        m = masks
        xg = x[g * R:(g + 1) * R]
        z1 = torch.nn.functional.linear(xg, p["W1"], p.get("b1"))
        z1 = z1 * m[:, None]
        h, mu1, v1 = _bn(z1, m, p["g1"], p["be1"])
        h = torch.relu(h) * m[:, None]
        z2 = torch.nn.functional.linear(h, p["W2"], p.get("b2")) * m[:, None]
        if has_bn2:
            y, mu2, v2 = _bn(z2, m, p["g2"], p["be2"])
            if relu_out:
                y = torch.relu(y)
            if residual is not None:
                y = y + residual[g * R:(g + 1) * R]
            y = y * m[:, None]
            stats.append((mu1, v1, mu2, v2))
        else:
            y = z2
            stats.append((mu1, v1))
        ys.append(y)
    return torch.cat(ys, 0), stats


def attributed(hip, r32, r64, what):
    hip, r32, r64 = hip.detach().cpu().double(), r32.detach().double(), r64.detach().double()
    scale = max(r64.abs().max().item(), 1e-300)
    e_hip, e_cpu = (hip - r64).abs().max().item(), (r32 - r64).abs().max().item()
    assert e_hip <= max(PU.REL * scale, 2.0 * e_cpu + PU.ATTR * scale), \
        f"{what}: |hip - f64| {e_hip / scale:.2e} vs |cpu32 - f64| {e_cpu / scale:.2e} (relative to max |f64| {scale:.3e})"
    return e_hip / scale, e_cpu / scale


CASES = [
    # (d_in, d_hid, d_out, G, N, K, masked, residual, relu_out, bias)
    (128, 128, 128, 2, 300, 16, True, True, True, False),      # a phi layer of the headline model, both sign passes
    (128, 128, 128, 1, 333, 1, False, True, True, False),      # a GINE layer (plain rows, R not a multiple of 16)
    (64, 64, 64, 2, 90, 8, True, True, True, True),            # configs[0] widths, Alchemy-style bias
    (32, 128, 64, 1, 77, 4, True, False, True, True),          # unequal widths
    (108, 108, 108, 2, 120, 6, True, True, True, True),        # Alchemy's hidden width (not a multiple of 16)
    (8, 8, 8, 2, 40, 8, True, True, True, False),              # half a channel tile, a few hundred rows (found by scratch fuzzing)
    (8, 8, 8, 2, 211, 5, True, True, True, True),
    (20, 20, 20, 2, 64, 5, True, True, True, True),
    (64, 64, 64, 2, 739, 5, True, True, True, True),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_mlp2_bn_forward_backward_vs_float64(case):
    from signnet_basisnet_amd import train_stage as T
    d_in, d_hid, d_out, G, N, K, masked, use_res, relu_out, bias = case
    gen = torch.Generator().manual_seed(11)
    torch.manual_seed(11)          # nn.Linear's default initialisation draws from the global generator
    R = N * K
    nvalid = torch.randint(1, K + 1, (N,), generator=gen) if masked else None
    m = _mask(nvalid, K, R)
    lin1, lin2 = torch.nn.Linear(d_in, d_hid, bias=bias), torch.nn.Linear(d_hid, d_out, bias=bias)
    bn1, bn2 = torch.nn.BatchNorm1d(d_hid), torch.nn.BatchNorm1d(d_out)
    with torch.no_grad():
        for bn in (bn1, bn2):
            bn.weight.copy_(torch.rand(bn.weight.shape, generator=gen) + 0.5)
            bn.bias.copy_(torch.randn(bn.bias.shape, generator=gen) * 0.3)
    x = torch.randn(G * R, d_in, generator=gen) * m.repeat(G)[:, None]
    res = torch.randn(G * R, d_out, generator=gen) * m.repeat(G)[:, None] if use_res else None
    # (convention of autograd.py: gradients flowing into an op are zero on invalid rows — every producer masks them)
    cot = torch.randn(G * R, d_out, generator=gen) * m.repeat(G)[:, None]

    def run_ref(dt):
        p = {"W1": lin1.weight, "W2": lin2.weight, "g1": bn1.weight, "be1": bn1.bias, "g2": bn2.weight, "be2": bn2.bias}
        if bias:
            p["b1"], p["b2"] = lin1.bias, lin2.bias
        p = {k: v.detach().to(dt).requires_grad_(True) for k, v in p.items()}
        xx = x.detach().clone().to(dt).requires_grad_(True)
        rr = None if res is None else res.detach().clone().to(dt).requires_grad_(True)
        y, stats = ref_mlp2(xx, p, m, G, rr, relu_out)
        (y * cot.to(dt)).sum().backward()
        grads = {k: v.grad for k, v in p.items()}
        grads["x"] = xx.grad
        if rr is not None:
            grads["res"] = rr.grad
        return y, grads, stats

    y64, g64, st64 = run_ref(torch.float64)
    y32, g32, _ = run_ref(torch.float32)

    mods = torch.nn.ModuleList([lin1, bn1, lin2, bn2]).to(DEV).train()
    xd = x.to(DEV).requires_grad_(True)
    rd = None if res is None else res.to(DEV).requires_grad_(True)
    nv = None if nvalid is None else nvalid.to(DEV).int()
    y = T.mlp2_bn(xd, lin1, bn1, lin2, bn2, nv, K, G, residual=rd, relu_out=relu_out)
    (y * cot.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    attributed(y, y32, y64, "y")
    assert torch.equal(y.detach().cpu()[~m.repeat(G)], torch.zeros_like(y64[~m.repeat(G)]).float()), "invalid rows must be exactly 0"
    hip = {"W1": lin1.weight.grad, "W2": lin2.weight.grad, "g1": bn1.weight.grad, "be1": bn1.bias.grad, "g2": bn2.weight.grad,
           "be2": bn2.bias.grad, "x": xd.grad}
    if bias:
        hip["b1"], hip["b2"] = lin1.bias.grad, lin2.bias.grad
    if rd is not None:
        hip["res"] = rd.grad
    gmax = max(v.abs().max().item() for v in g64.values())
    for k, v in hip.items():
        assert v is not None, k
        r64 = g64[k]
        if r64.abs().max().item() <= 1e-9 * gmax:       # a bias in front of a batch-statistics BatchNorm: true gradient 0
            assert v.abs().max().item() <= 1e-5 * gmax, k
            continue
        attributed(v, g32[k], r64, "d" + k)
    # running statistics: G sequential updates with momentum 0.1 (the reference calls the module once per sign pass)
    for bn, (i_mu, i_var) in ((bn1, (0, 1)), (bn2, (2, 3))):
        rm, rv = torch.zeros(bn.num_features, dtype=torch.float64), torch.ones(bn.num_features, dtype=torch.float64)
        for g in range(G):
            rm = 0.9 * rm + 0.1 * st64[g][i_mu].detach()
            rv = 0.9 * rv + 0.1 * st64[g][i_var].detach()
        PU.close(bn.running_mean, rm, "running_mean", rel=2e-5)
        PU.close(bn.running_var, rv, "running_var", rel=2e-5)
        assert int(bn.num_batches_tracked) == G
    # bitwise reproducible (no atomics)
    for pth in (lin1.weight, lin2.weight, bn1.weight, bn2.bias):
        pth.grad = None
    xd2 = x.to(DEV).requires_grad_(True)
    y2 = T.mlp2_bn(xd2, lin1, bn1, lin2, bn2, nv, K, G, residual=None if res is None else res.to(DEV), relu_out=relu_out)
    (y2 * cot.to(DEV)).sum().backward()
    assert torch.equal(lin1.weight.grad, hip["W1"]) and torch.equal(lin2.weight.grad, hip["W2"]) and torch.equal(xd2.grad, hip["x"])


@pytest.mark.parametrize("relu,bias,masked", [(False, False, True), (True, True, True), (False, True, False)])
def test_stage_linear_vs_float64(relu, bias, masked):
    from signnet_basisnet_amd import train_stage as T
    gen = torch.Generator().manual_seed(5)
    torch.manual_seed(5)
    N, K, d_in, d_out = 211, 16, 128, 128
    R = N * K
    nvalid = torch.randint(1, K + 1, (N,), generator=gen) if masked else None
    m = _mask(nvalid, K, R)
    lin = torch.nn.Linear(d_in, d_out, bias=bias)
    x = torch.randn(R, d_in, generator=gen) * m[:, None]
    cot = torch.randn(R, d_out, generator=gen)

    def run_ref(dt):
        W = lin.weight.detach().to(dt).requires_grad_(True)
        b = lin.bias.detach().to(dt).requires_grad_(True) if bias else None
        xx = x.detach().clone().to(dt).requires_grad_(True)
        y = torch.nn.functional.linear(xx, W, b)
        if relu:
            y = torch.relu(y)
        y = y * m[:, None]
        (y * cot.to(dt)).sum().backward()
        return y, {"x": xx.grad, "W": W.grad, **({"b": b.grad} if bias else {})}

    y64, g64 = run_ref(torch.float64)
    y32, g32 = run_ref(torch.float32)
    lin = lin.to(DEV)
    xd = x.to(DEV).requires_grad_(True)
    y = T.linear(xd, lin.weight, lin.bias, None if nvalid is None else nvalid.to(DEV).int(), K, relu=relu)
    (y * cot.to(DEV)).sum().backward()
    attributed(y, y32, y64, "y")
    attributed(xd.grad, g32["x"], g64["x"], "dx")
    attributed(lin.weight.grad, g32["W"], g64["W"], "dW")
    if bias:
        attributed(lin.bias.grad, g32["b"], g64["b"], "db")


def test_headline_size_link_both_signs():
    """The phi link at the size of BASELINE configs[1] / [3] per rank: 2 x 47 200 rows x 128, k = 16 slots per node."""
    from signnet_basisnet_amd import train_stage as T
    gen = torch.Generator().manual_seed(2)
    torch.manual_seed(2)
    N, K, d, G = 2950, 16, 128, 2
    R = N * K
    nvalid = torch.clamp(torch.randint(9, 38, (N,), generator=gen), max=K)
    m = _mask(nvalid, K, R)
    lin1, lin2 = torch.nn.Linear(d, d, bias=False), torch.nn.Linear(d, d, bias=False)
    bn1, bn2 = torch.nn.BatchNorm1d(d), torch.nn.BatchNorm1d(d)
    x = torch.randn(G * R, d, generator=gen) * m.repeat(G)[:, None]
    res = torch.randn(G * R, d, generator=gen) * m.repeat(G)[:, None]
    cot = torch.randn(G * R, d, generator=gen) * m.repeat(G)[:, None]

    def run_ref(dt):
        p = {"W1": lin1.weight, "W2": lin2.weight, "g1": bn1.weight, "be1": bn1.bias, "g2": bn2.weight, "be2": bn2.bias}
        p = {k: v.detach().to(dt).requires_grad_(True) for k, v in p.items()}
        xx = x.detach().clone().to(dt).requires_grad_(True)
        y, _ = ref_mlp2(xx, p, m, G, res.to(dt), True)
        (y * cot.to(dt)).sum().backward()
        return y, {**{k: v.grad for k, v in p.items()}, "x": xx.grad}

    y64, g64 = run_ref(torch.float64)
    y32, g32 = run_ref(torch.float32)
    torch.nn.ModuleList([lin1, bn1, lin2, bn2]).to(DEV).train()
    xd = x.to(DEV).requires_grad_(True)
    y = T.mlp2_bn(xd, lin1, bn1, lin2, bn2, nvalid.to(DEV).int(), K, G, residual=res.to(DEV))
    (y * cot.to(DEV)).sum().backward()
    print("y", attributed(y, y32, y64, "y"))
    for k, v in {"W1": lin1.weight.grad, "W2": lin2.weight.grad, "g1": bn1.weight.grad, "be1": bn1.bias.grad, "g2": bn2.weight.grad,
                 "be2": bn2.bias.grad, "x": xd.grad}.items():
        print(k, attributed(v, g32[k], g64[k], "d" + k))


@pytest.mark.parametrize("d,N,K,G", [(128, 200, 16, 2), (64, 57, 8, 1), (36, 40, 6, 2)])
def test_scalar_mlp_closed_form_vs_float64(d, N, K, G):
    """The 1 -> 1 -> d MaskedMLP on a scalar input (GINESignNetPyG's first phi layer, both sign passes; eigen_encoder2) in closed form
    against float64 autograd of the literal composition Linear(1,1).BN.ReLU.Linear(1,d).BN.ReLU over the valid rows."""
    from signnet_basisnet_amd import train_stage as T
    gen = torch.Generator().manual_seed(3)
    torch.manual_seed(3)
    M = N * K
    nvalid = torch.randint(1, K + 1, (N,), generator=gen)
    m = _mask(nvalid, K, M)
    lin1, lin2 = torch.nn.Linear(1, 1, bias=False), torch.nn.Linear(1, d, bias=False)
    bn1, bn2 = torch.nn.BatchNorm1d(1), torch.nn.BatchNorm1d(d)
    with torch.no_grad():
        lin1.weight.fill_(0.7)
        for bn in (bn1, bn2):
            bn.weight.copy_(torch.rand(bn.weight.shape, generator=gen) + 0.5)
            bn.bias.copy_(torch.randn(bn.bias.shape, generator=gen) * 0.3)
    a = torch.randn(M, generator=gen) * m
    cot = torch.randn(G * M, d, generator=gen) * m.repeat(G)[:, None]

    def run_ref(dt):
        p = {"w1": lin1.weight, "w2": lin2.weight, "ga": bn1.weight, "ba": bn1.bias, "gb": bn2.weight, "bb": bn2.bias}
        p = {k: v.detach().to(dt).requires_grad_(True) for k, v in p.items()}
        aa = a.detach().clone().to(dt).requires_grad_(True)
        ys = []
        for g in range(G):
            x = (aa if g == 0 else -aa)[:, None]
            z = (x @ p["w1"].T) * m[:, None]
            h, _, _ = _bn(z, m, p["ga"], p["ba"])
            h = torch.relu(h) * m[:, None]
            z2 = (h @ p["w2"].T) * m[:, None]
            y, _, _ = _bn(z2, m, p["gb"], p["bb"])
            ys.append(torch.relu(y) * m[:, None])
        y = torch.cat(ys, 0)
        (y * cot.to(dt)).sum().backward()
        return y, {**{k: v.grad for k, v in p.items()}, "a": aa.grad}

    y64, g64 = run_ref(torch.float64)
    y32, g32 = run_ref(torch.float32)
    torch.nn.ModuleList([lin1, bn1, lin2, bn2]).to(DEV).train()
    ad = a.to(DEV).requires_grad_(True)
    y = T.scalar_mlp(ad, lin1, bn1, lin2, bn2, nvalid.to(DEV).int(), K, G=G, negate_second=(G == 2))
    (y * cot.to(DEV)).sum().backward()
    attributed(y, y32, y64, "y")
    hip = {"w1": lin1.weight.grad, "w2": lin2.weight.grad, "ga": bn1.weight.grad, "ba": bn1.bias.grad, "gb": bn2.weight.grad,
           "bb": bn2.bias.grad, "a": ad.grad}
    gmax = max(v.abs().max().item() for v in g64.values())
    for k, v in hip.items():
        assert v is not None, k
        if g64[k].abs().max().item() <= 1e-9 * gmax:
            assert v.abs().max().item() <= 1e-5 * gmax, k
            continue
        attributed(v, g32[k], g64[k], "d" + k)
    assert int(bn1.num_batches_tracked) == G and int(bn2.num_batches_tracked) == G


# ----------------------------------------------------------------------------- round 6: the launch structures of a link's reductions
def _train_step_grads(fuse_finish, defer_dw, monkeypatch):
    """Gradients, loss and running statistics of one training step of a small ragged model under the given launch structure."""
    from signnet_basisnet_amd import optim, synth, train_stage
    from signnet_basisnet_amd.pyg import SignNetGNN
    monkeypatch.setattr(train_stage, "FUSE_FINISH", fuse_finish)
    monkeypatch.setattr(train_stage, "DEFER_DW", defer_dw)
    torch.manual_seed(7)
    m = SignNetGNN(None, None, 128, 1, 3, 2, variant="gine", max_k=8).to(DEV).train()
    m.attn_dropout = 0.0
    o = optim.FlatAdam(m.parameters(), lr=1e-3)
    d = synth.batch_to(synth.make_batch(40, seed=5), DEV)
    target = torch.randn(40, 1, generator=torch.Generator().manual_seed(2)).to(DEV)
    o.zero_grad()
    loss = (m(d) - target).abs().mean()
    loss.backward()
    return o.flat_g.clone(), loss.item(), [b.clone() for b in m.buffers()]


def test_deferred_dw_reduction_gives_the_same_bits(monkeypatch):
    """The dW / db partials of every backward link reduced by ONE launch at the end of loss.backward() (sn_train_reduce_jobs_f32, queued
    by autograd's end-of-backward callback) instead of one launch per link: the same arithmetic per parameter, so the same bits."""
    g0, l0, b0 = _train_step_grads(False, False, monkeypatch)
    g1, l1, b1 = _train_step_grads(False, True, monkeypatch)
    assert l0 == l1 and torch.equal(g0, g1) and all(torch.equal(x, y) for x, y in zip(b0, b1))
    assert g0.abs().max().item() > 0


def test_in_launch_finishes_agree_with_the_separate_launches(monkeypatch):
    """The opt-in launch structure (SN_TRAIN_FUSE_FINISH=1): batch statistics, BatchNorm-backward coefficients and the eps gradient finished
    by the LAST-ARRIVING workgroup of the link's own launch (agent-scope ticket behind write-through partials).  Same slicing and order as
    the finish kernels; the two are compiled separately, so FMA contraction may differ in the last bit: compared at 1e-5 of the largest
    gradient entry (and the running statistics at 1e-6), far below the 1e-3 two fp32 evaluations of a step differ by."""
    g0, l0, b0 = _train_step_grads(False, True, monkeypatch)
    g1, l1, b1 = _train_step_grads(True, True, monkeypatch)
    assert abs(l0 - l1) <= 1e-6 * abs(l0)
    assert (g0 - g1).abs().max().item() <= 1e-5 * g0.abs().max().item()
    for x, y in zip(b0, b1):
        if x.is_floating_point():
            assert (x - y).abs().max().item() <= 1e-6 * max(1.0, x.abs().max().item())
        else:
            assert torch.equal(x, y)
    # ... and replayed: tickets are reset by their last arriver, so a second and third step work from the same words
    g2, _, _ = _train_step_grads(True, True, monkeypatch)
    assert torch.equal(g1, g2)


def test_consumer_side_coefficient_merge_agrees_with_the_finish_launches(monkeypatch):
    """SN_TRAIN_MERGE (default on): the backward link merges the column-sum partials of its own BatchNorm in its prologue instead of reading
    coefficients a finish launch wrote; same slicing and order, separately compiled: gradients agree to 1e-6 of the largest entry, the
    forward (loss, running statistics) is untouched."""
    from signnet_basisnet_amd import train_stage
    monkeypatch.setattr(train_stage, "MERGE_COEF", False)
    g0, l0, b0 = _train_step_grads(False, True, monkeypatch)
    monkeypatch.setattr(train_stage, "MERGE_COEF", True)
    g1, l1, b1 = _train_step_grads(False, True, monkeypatch)
    assert l0 == l1 and all(torch.equal(x, y) for x, y in zip(b0, b1))
    assert (g0 - g1).abs().max().item() <= 1e-6 * g0.abs().max().item()


def test_reduce_jobs_entry_point_vs_torch():
    import ctypes as C
    from signnet_basisnet_amd import train_stage as T
    from signnet_basisnet_amd._lib import check, lib, stream
    torch.manual_seed(0)
    parts = [torch.randn(37, 300, device=DEV), torch.randn(5, 64, device=DEV), torch.randn(256, 1000, device=DEV)]
    ns = [260, 64, 1000]
    outs = [torch.randn(n, device=DEV) for n in ns]
    want = [o.double() + p[:, :n].double().sum(0) for o, p, n in zip(outs, parts, ns)]
    arr = (T._ReduceJob * 3)()
    for j, (p, n, o) in enumerate(zip(parts, ns, outs)):
        arr[j] = T._ReduceJob(p.data_ptr(), p.shape[0], p.shape[1], n, o.data_ptr(), 1)
    check(lib().sn_train_reduce_jobs_f32(arr, 3, stream()), "sn_train_reduce_jobs_f32")
    for o, w in zip(outs, want):
        assert (o.double() - w).abs().max().item() <= 1e-5 * w.abs().max().item()
    assert lib().sn_train_reduce_jobs_f32(arr, 65, stream()) == -1        # more than SN_TRAIN_MAX_REDUCE_JOBS: refused on the host


def test_clock_probe_counts_what_it_is_asked():
    from signnet_basisnet_amd._lib import check, lib, stream
    buf = torch.zeros(1, dtype=torch.int64, device=DEV)
    check(lib().sn_clock_probe(2_000_000, buf.data_ptr(), stream()), "sn_clock_probe")
    torch.cuda.synchronize()
    assert 2_000_000 <= int(buf.item()) < 2_100_000
