"""-m gpu: every adjoint kernel of csrc/backward.hip against torch.autograd on a float64 CPU restatement
of the same op (the reference obtains these gradients from torch.autograd, main_alchemy.py:108)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def close(a, b, what, rel=2e-5):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(1e-6, b.abs().max().item())
    err = (a - b).abs().max().item()
    assert err <= rel * scale, f"{what}: max|diff| {err:.3e} vs scale {scale:.3e}"


def leaf(t, dev=None):
    return t.detach().clone().to(dev or t.device).requires_grad_(True)


def setup(N=37, K=7, seed=0):
    g = torch.Generator().manual_seed(seed)
    nv = torch.randint(1, K + 1, (N,), generator=g, dtype=torch.int32)
    nv[0] = K
    mask = (torch.arange(K)[None, :] < nv[:, None]).reshape(N * K, 1).double()
    return g, nv, mask


def run_pair(fn_hip, fn_ref, inputs, what, rel=2e-5):
    """inputs: list of fp32 CPU tensors; both functions map them to one output; compares output and all input grads."""
    xs_h = [leaf(t, DEV) for t in inputs]
    xs_r = [leaf(t.double()) for t in inputs]
    yh, yr = fn_hip(*xs_h), fn_ref(*xs_r)
    close(yh, yr, what + " forward", rel)
    cot = torch.randn(yr.shape, generator=torch.Generator().manual_seed(99), dtype=torch.float64)
    yh.backward(cot.float().to(DEV))
    yr.backward(cot)
    for i, (a, b) in enumerate(zip(xs_h, xs_r)):
        close(a.grad, b.grad, f"{what} grad[{i}]", rel)


@pytest.mark.parametrize("d_in,d_out,relu,bias", [(128, 128, False, True), (1, 40, True, True), (40, 1, False, False),
                                                  (200, 72, True, True), (36, 36, False, True)])
def test_linear_backward(d_in, d_out, relu, bias):
    from signnet_basisnet_amd import autograd as AG
    g, nv, mask = setup(N=53, K=9)
    x = torch.randn(53 * 9, d_in, generator=g) * mask.float()
    W = torch.randn(d_out, d_in, generator=g) / d_in ** 0.5
    b = torch.randn(d_out, generator=g)
    nvd = nv.to(DEV)

    def ref(x, W, *b_):
        y = x @ W.t() + (b_[0] if b_ else 0)
        if relu:
            y = torch.relu(y)
        return y * mask

    def hip(x, W, *b_):
        return AG.linear(x, W, b_[0] if b_ else None, nvd, 9, relu=relu)
    run_pair(hip, ref, [x, W] + ([b] if bias else []), f"linear {d_in}->{d_out}")
    # unmasked plain rows (graph-level head): R not a multiple of anything
    x2 = torch.randn(131, d_in, generator=g)
    run_pair(lambda x, W: AG.linear(x, W, None, None, 0, relu=relu),
             lambda x, W: torch.relu(x @ W.t()) if relu else x @ W.t(), [x2, W], "linear plain rows")


def test_linear_wgrad_many_rows():
    """Row counts beyond one chunk (partials + reduction) — the N*K = 47k rows of the bench batch."""
    from signnet_basisnet_amd import autograd as AG
    g = torch.Generator().manual_seed(1)
    x, dy = torch.randn(20011, 128, generator=g), torch.randn(20011, 128, generator=g)
    dW, db = AG.linear_wgrad(x.to(DEV), dy.to(DEV), None, 0)
    close(dW, dy.double().t() @ x.double(), "dW", 1e-5)
    close(db, dy.double().sum(0), "db", 1e-5)


@pytest.mark.parametrize("relu,res", [(True, False), (True, True), (False, False)])
def test_bn_act_backward(relu, res):
    from signnet_basisnet_amd import autograd as AG
    g, nv, mask = setup(N=41, K=6, seed=2)
    Cc = 44
    z = torch.randn(41 * 6, Cc, generator=g)
    gamma, beta = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.1
    r = torch.randn(41 * 6, Cc, generator=g) * mask.float()
    bn = torch.nn.BatchNorm1d(Cc).to(DEV).train()
    nvd = nv.to(DEV)

    def ref(z, gamma, beta, *rr):
        sel = mask[:, 0] > 0
        zv = z[sel]
        mean, var = zv.mean(0), zv.var(0, unbiased=False)
        a = (z - mean) / torch.sqrt(var + bn.eps) * gamma + beta
        if relu:
            a = torch.relu(a)
        if rr:
            a = a + rr[0]
        return a * mask

    def hip(z, gamma, beta, *rr):
        bn.weight, bn.bias = torch.nn.Parameter(gamma.detach()), torch.nn.Parameter(beta.detach())
        # route the leaves through the Function directly so that their .grad is filled
        return AG._BnAct.apply(z, gamma, beta, rr[0] if rr else None, bn, nvd, 6, relu)
    run_pair(hip, ref, [z, gamma, beta] + ([r] if res else []), "bn_act", 5e-5)
    assert int(bn.num_batches_tracked) == 1


def _graph(N=40, E=150, seed=3):
    g = torch.Generator().manual_seed(seed)
    src, dst = torch.randint(0, N, (E,), generator=g), torch.randint(0, N, (E,), generator=g)
    ei = torch.stack([src, dst])
    batch = torch.zeros(N, dtype=torch.int64)
    return g, ei, batch


def test_gin_and_gine_aggregate_backward():
    from signnet_basisnet_amd import autograd as AG
    from signnet_basisnet_amd import ops
    N, E, Cc = 40, 150, 24
    g, ei, batch = _graph(N, E)
    plan = ops.build_plan(batch.to(DEV), ei.to(DEV), 1, 0)
    rplan = ops.build_plan(batch.to(DEV), ei.flip(0).contiguous().to(DEV), 1, 0)
    A = torch.zeros(N, N, dtype=torch.float64)
    A.index_put_((ei[1], ei[0]), torch.ones(E, dtype=torch.float64), accumulate=True)
    x = torch.randn(N, 3 * Cc, generator=g)
    eps = torch.tensor([0.3])
    for neg in (False, True):
        run_pair(lambda x, eps: AG.gin_aggregate(x, eps, plan, rplan, negate=neg),
                 lambda x, eps: ((1 + eps) * x + A @ x) * (-1 if neg else 1), [x, eps], f"gin negate={neg}")
    h, ee = torch.randn(N, Cc, generator=g), torch.randn(E, Cc, generator=g)

    def ref(h, ee, eps):
        m = torch.relu(h[ei[0]] + ee)
        return (1 + eps) * h + torch.zeros_like(h).index_add_(0, ei[1], m)
    run_pair(lambda h, ee, eps: AG.gine_aggregate(h, ee, eps, plan, rplan), ref, [h, ee, eps], "gine")


def test_gated_aggregate_forward_backward():
    """GatedGCN's edge-gated aggregation (gatedgcn_layer.py:51-56) and its two-pass adjoint."""
    from signnet_basisnet_amd import autograd as AG
    from signnet_basisnet_amd import ops
    N, E, Cc = 40, 150, 20
    g, ei, batch = _graph(N, E, seed=8)
    plan = ops.build_plan(batch.to(DEV), ei.to(DEV), 1, 0)
    rplan = ops.build_plan(batch.to(DEV), ei.flip(0).contiguous().to(DEV), 1, 0)
    Ah, Bh, Dh, Eh = (torch.randn(N, Cc, generator=g) for _ in range(4))
    Ce = torch.randn(E, Cc, generator=g)
    wh, we = torch.randn(N, Cc, generator=g).double(), torch.randn(E, Cc, generator=g).double()

    def ref(Ah, Bh, Dh, Eh, Ce):
        en = Dh[ei[0]] + Eh[ei[1]] + Ce
        sg = torch.sigmoid(en)
        num = torch.zeros_like(Ah).index_add_(0, ei[1], Bh[ei[0]] * sg)
        den = torch.zeros_like(Ah).index_add_(0, ei[1], sg)
        return ((Ah + num / (den + 1e-6)) * wh).sum(1, keepdim=True).sum(0, keepdim=True) + (en * we).sum()

    def hip(Ah, Bh, Dh, Eh, Ce):
        h, e = AG.gated_aggregate(Ah, Bh, Dh, Eh, Ce, plan, rplan)
        return (h * wh.float().to(DEV)).sum(1, keepdim=True).sum(0, keepdim=True) + (e * we.float().to(DEV)).sum()
    run_pair(hip, ref, [Ah, Bh, Dh, Eh, Ce], "gated_aggregate", 5e-5)


def test_attention_layernorm_slotsum_backward():
    from signnet_basisnet_amd import autograd as AG
    N, K, H, dk = 29, 7, 4, 8
    D = H * dk
    g, nv, mask = setup(N, K, seed=4)
    nvd = nv.to(DEV)
    q, k, v = (torch.randn(N * K, D, generator=g) * mask.float() for _ in range(3))

    def att_ref(q, k, v):
        qh, kh, vh = (t.view(N, K, H, dk).permute(0, 2, 1, 3) for t in (q, k, v))
        s = qh @ kh.transpose(-1, -2) / dk ** 0.5
        valid = torch.arange(K)[None, :] < nv[:, None]                        # [N, K]
        s = s.masked_fill(~valid[:, None, None, :], float("-inf"))
        o = torch.softmax(s, -1) @ vh
        return o.permute(0, 2, 1, 3).reshape(N * K, D) * mask
    run_pair(lambda q, k, v: AG.set_attention(q, k, v, N, K, H, nvd), att_ref, [q, k, v], "set_attention", 5e-5)
    # attention dropout as an explicit mask (transformer_module.py:55: dropout(softmax(.)))
    from signnet_basisnet_amd import ops
    torch.manual_seed(7)
    pm = ops.attention_dropout_mask(N, K, H, 0.25, DEV)
    kept = (pm > 0).float().mean().item()
    assert 0.6 < kept < 0.9 and abs(pm.max().item() - 1 / 0.75) < 1e-6
    pmc = pm.cpu().double()

    def att_drop_ref(q, k, v):
        qh, kh, vh = (t.view(N, K, H, dk).permute(0, 2, 1, 3) for t in (q, k, v))
        s = qh @ kh.transpose(-1, -2) / dk ** 0.5
        valid = torch.arange(K)[None, :] < nv[:, None]
        s = s.masked_fill(~valid[:, None, None, :], float("-inf"))
        o = (torch.softmax(s, -1) * pmc) @ vh
        return o.permute(0, 2, 1, 3).reshape(N * K, D) * mask
    run_pair(lambda q, k, v: AG.set_attention(q, k, v, N, K, H, nvd, pm), att_drop_ref, [q, k, v], "set_attention+dropout", 5e-5)

    x, r = torch.randn(N * K, D, generator=g), torch.randn(N * K, D, generator=g)
    gamma, beta = torch.rand(D, generator=g) + 0.5, torch.randn(D, generator=g)

    def ln_ref(x, r, gamma, beta):
        return torch.nn.functional.layer_norm(x + r, (D,), gamma, beta, 1e-6) * mask
    run_pair(lambda x, r, gamma, beta: AG.masked_layernorm(x, r, gamma, beta, 1e-6, nvd, K), ln_ref, [x, r, gamma, beta],
             "layernorm", 5e-5)
    run_pair(lambda x, gamma, beta: AG.masked_layernorm(x, None, gamma, beta, 1e-6, nvd, K),
             lambda x, gamma, beta: torch.nn.functional.layer_norm(x, (D,), gamma, beta, 1e-6) * mask, [x, gamma, beta],
             "layernorm no residual", 5e-5)
    xm = x * mask.float()
    run_pair(lambda x: AG.slot_sum(x, N, K, nvd), lambda x: (x * mask).view(N, K, D).sum(1), [xm], "slot_sum")
    run_pair(lambda a, b: AG.masked_add(a, b, nvd, K), lambda a, b: (a + b) * mask, [xm, r * mask.float()], "masked_add")
    # the largest slot count of the fused-path envelope (64 slots, head width 32): > 64 KiB of LDS in the backward
    N2, K2 = 3, 64
    g2 = torch.Generator().manual_seed(11)
    nv2 = torch.tensor([64, 40, 1], dtype=torch.int32)
    mask2 = (torch.arange(K2)[None, :] < nv2[:, None]).reshape(N2 * K2, 1).double()
    q2, k2, v2 = (torch.randn(N2 * K2, 128, generator=g2) * mask2.float() for _ in range(3))

    def att_ref2(q, k, v):
        qh, kh, vh = (t.view(N2, K2, 4, 32).permute(0, 2, 1, 3) for t in (q, k, v))
        s = qh @ kh.transpose(-1, -2) / 32 ** 0.5
        valid = torch.arange(K2)[None, :] < nv2[:, None]
        s = s.masked_fill(~valid[:, None, None, :], float("-inf"))
        return (torch.softmax(s, -1) @ vh).permute(0, 2, 1, 3).reshape(N2 * K2, 128) * mask2
    run_pair(lambda q, k, v: AG.set_attention(q, k, v, N2, K2, 4, nv2.to(DEV)), att_ref2, [q2, k2, v2], "set_attention K=64", 5e-5)


def test_embedding_and_pool_backward():
    from signnet_basisnet_amd import autograd as AG
    from signnet_basisnet_amd import ops
    g = torch.Generator().manual_seed(5)
    R, Cc = 300, 20
    idx = torch.stack([torch.randint(0, 11, (R,), generator=g), torch.randint(0, 5, (R,), generator=g)], 1)
    t0, t1, t2 = torch.randn(11, Cc, generator=g), torch.randn(5, Cc, generator=g), torch.randn(7, Cc, generator=g)
    idx_d = idx.to(DEV)
    run_pair(lambda a, b: AG.embedding_sum(idx_d, [a, b]), lambda a, b: a[idx[:, 0]] + b[idx[:, 1]], [t0, t1], "embedding_sum")
    # a third table without a feature column gets no gradient
    a, b, c = leaf(t0, DEV), leaf(t1, DEV), leaf(t2, DEV)
    AG.embedding_sum(idx_d, [a, b, c]).sum().backward()
    assert c.grad is None and a.grad is not None
    # no atomics: the table gradients are bitwise reproducible (500-row tables as in GINESignNetPyG, many repeated ids)
    big = torch.randint(0, 28, (5000, 1), generator=g).to(DEV)
    tab = torch.randn(500, 128, generator=g)
    up = torch.randn(5000, 128, generator=g).to(DEV)
    grads = []
    for _ in range(3):
        t = leaf(tab, DEV)
        (AG.embedding_sum(big, [t]) * up).sum().backward()
        grads.append(t.grad.clone())
    assert torch.equal(grads[0], grads[1]) and torch.equal(grads[0], grads[2])
    ref = torch.zeros(500, 128, dtype=torch.float64).index_add_(0, big[:, 0].cpu(), up.cpu().double())
    close(grads[0], ref, "embedding table gradient", 1e-5)
    # a table with more rows than the workgroup budget of the chunking (vocab 5000 > 4096: the rows-per-chunk loop of
    # embedding_bwd_rpc did not terminate there and divided by zero; the entry point accepts up to 65535 rows)
    bigv = torch.randint(0, 5000, (1000, 1), generator=g).to(DEV)
    tabv = torch.randn(5000, 12, generator=g)
    upv = torch.randn(1000, 12, generator=g).to(DEV)
    t = leaf(tabv, DEV)
    (AG.embedding_sum(bigv, [t]) * upv).sum().backward()
    refv = torch.zeros(5000, 12, dtype=torch.float64).index_add_(0, bigv[:, 0].cpu(), upv.cpu().double())
    close(t.grad, refv, "embedding table gradient, vocabulary 5000", 1e-5)
    # L encoders over one index block (the per-layer edge encoders of the GINE stack): one adjoint launch pair for all planes
    L, Re = 3, 700
    idx2 = torch.stack([torch.randint(0, 4, (Re,), generator=g), torch.randint(0, 9, (Re,), generator=g)], 1)
    tabs = [[torch.randn(500, 128, generator=g), torch.randn(40, 128, generator=g), torch.randn(6, 128, generator=g)] for _ in range(L)]
    ups = torch.randn(L, Re, 128, generator=g)
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    runs = []
    for _ in range(2):
        leaves = [[leaf(t, DEV) for t in ts] for ts in tabs]
        blk = AG.embedding_sum_layers(idx2.to(DEV), leaves, status)
        assert blk.shape == (L, Re, 128) and hasattr(blk, "_sn_gbuf")
        for l in range(L):
            close(blk[l], tabs[l][0][idx2[:, 0]] + tabs[l][1][idx2[:, 1]], f"embedding_sum_layers plane {l}", 1e-6)
        (blk * ups.to(DEV)).sum().backward()
        runs.append([[t.grad for t in ts] for ts in leaves])
    assert int(status.item()) == 0
    for l in range(L):
        assert runs[0][l][2] is None                       # a table without a feature column
        for f in range(2):
            ref = torch.zeros(tabs[l][f].shape, dtype=torch.float64).index_add_(0, idx2[:, f], ups[l].double())
            close(runs[0][l][f], ref, f"embedding_sum_layers table gradient {l}/{f}", 1e-5)
            assert torch.equal(runs[0][l][f], runs[1][l][f])
    sizes = [5, 1, 17, 30, 9]
    batch = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
    plan = ops.build_plan(batch.to(DEV), torch.zeros(2, 0, dtype=torch.int64, device=DEV), len(sizes), 0)
    h = torch.randn(sum(sizes), Cc, generator=g)
    for mode in ("add", "mean"):
        def ref(h):
            out = torch.zeros(len(sizes), Cc, dtype=h.dtype).index_add_(0, batch, h)
            return out / torch.tensor(sizes, dtype=h.dtype)[:, None] if mode == "mean" else out
        run_pair(lambda h: AG.segment_pool(h, plan, mode), ref, [h], f"segment_pool {mode}")


def test_adam_step_matches_torch():
    from signnet_basisnet_amd import optim
    g = torch.Generator().manual_seed(6)
    ps = [torch.randn(33, 17, generator=g), torch.randn(5, generator=g)]
    for wd in (0.0, 0.01):
        ref = [torch.nn.Parameter(p.clone().double()) for p in ps]
        mine = [torch.nn.Parameter(p.clone().to(DEV)) for p in ps]
        o_ref = torch.optim.Adam(ref, lr=1e-2, betas=(0.9, 0.99), eps=1e-8, weight_decay=wd)
        o_mine = optim.Adam(mine, lr=1e-2, betas=(0.9, 0.99), eps=1e-8, weight_decay=wd)
        for step in range(4):
            for a, b in zip(ref, mine):
                gr = torch.randn(a.shape, generator=g)
                a.grad, b.grad = gr.double(), gr.to(DEV)
            o_ref.step()
            o_mine.step()
        for a, b in zip(ref, mine):
            close(b, a, f"adam wd={wd}", 1e-5)
