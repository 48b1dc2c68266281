"""-m gpu: parity at the BASELINE.json batch sizes, asserted at north_star's 1e-5 (no slack factor).

configs[0]  ZINC-subset k=8  hidden=64  batch=32    GINESignNetPyG/core/config.py:30,53-57
configs[1]  ZINC        k=16 hidden=128 batch=128   (the headline)
configs[2]  Alchemy SignNetGNN(6,4,108,12,8,16), all eigenvectors, batch=256      Alchemy/main_alchemy.py:35,84
(+)         ZINC hidden=128 batch=128 with max_k=None — the reference's own default (all eigenvectors)
configs[4]  BasisNet on the real 32x32 grid: all three multiplicity groups (32 x mult 1, 480 x mult 2, 1 x mult 32)
            LearningFilters/training.py:47-73,119-126

Each case runs the CPU oracle twice — fp32 (what the reference computes) and float64 (the exact value) — and the HIP
modules on the same seeded inputs and weights, and asserts for every stage
    max|hip - cpu32| <= 1e-5 * max|cpu32|                                   (north_star)
    max|hip - f64|   <= max|cpu32 - f64| + 1e-6 * max|f64|                  (the HIP path is as exact as the fp32 CPU path)
plus an element-wise assert_close(rtol=1e-5, atol=1e-5*rms) on the model output.  The worst stage is printed (-s).
"""
import pytest
import torch

import parity_util as PU

pytestmark = pytest.mark.gpu

CASES = {
    "configs0_zinc_k8_h64_b32": dict(variant="gine", ctor=(None, None, 64, 1, 4, 6), feat="zinc", lo=9, hi=37, B=32, k=8, seed=1234),
    "configs1_zinc_k16_h128_b128": dict(variant="gine", ctor=(None, None, 128, 1, 4, 6), feat="zinc", lo=9, hi=37, B=128, k=16, seed=1235),
    "configs2_alchemy_b256": dict(variant="alchemy", ctor=(6, 4, 108, 12, 8, 16), feat="alchemy", lo=6, hi=14, B=256, k=None, seed=1236),
    "zinc_all_eigenvectors_h128_b128": dict(variant="gine", ctor=(None, None, 128, 1, 4, 6), feat="zinc", lo=9, hi=37, B=128, k=None, seed=1237),
}


@pytest.mark.parametrize("name", list(CASES))
def test_full_batch_vs_oracle_fp32_and_fp64(name):
    from oracle import pyg_signnet as O
    from signnet_basisnet_amd import synth
    from signnet_basisnet_amd.pyg import SignNetGNN
    c = CASES[name]
    torch.manual_seed(0)
    m = SignNetGNN(*c["ctor"], variant=c["variant"], max_k=c["k"])
    PU.bn_randomize(m, 1)
    data = synth.make_batch(c["B"], seed=c["seed"], n_lo=c["lo"], n_hi=c["hi"], features=c["feat"])
    cfg = O.make_cfg(c["variant"], *c["ctor"])
    sd32 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    o32, o64 = {}, {}
    with torch.no_grad():
        y32 = O.signnet_gnn(sd32, cfg, data, training=False, max_k=c["k"], out=o32)
        y64 = O.signnet_gnn(PU.to_f64(sd32), cfg, PU.data_f64(data), training=False, max_k=c["k"], out=o64)
    m = m.cuda().eval()
    dd = synth.batch_to(data, "cuda:0")
    with torch.no_grad():
        y_fused = m(dd)                      # the default forward: three whole-stage kernels
        m.check_last()
        y_layer, st = m(dd, return_stages=True)
    L = cfg["nl_gnn"]
    stages = [("phi (layer kernels)", st["phi"], o32["phi"], o64["phi"]),
              ("phi (fused stage)", st["phi_fused"], o32["phi"], o64["phi"]),
              ("rho slot sum (layer kernels)", st["rho_sum"], o32["rho_sum"], o64["rho_sum"]),
              ("rho slot sum (fused stage)", st["rho_sum_fused"], o32["rho_sum"], o64["rho_sum"]),
              ("sign_net output", st["pos"], o32["pos"], o64["pos"]),
              ("GINE layer 0", st["gine0"], o32["gnn_layers"][0], o64["gnn_layers"][0]),
              (f"GINE layer {L - 1}", st[f"gine{L - 1}"], o32["gnn_layers"][-1], o64["gnn_layers"][-1]),
              ("y (layer kernels)", y_layer, y32, y64),
              ("y (fused GINE stage)", st["y_gnn_fused"], y32, y64),
              ("y (default forward, all stages fused)", y_fused, y32, y64)]
    worst = ("", 0.0)
    for what, hip, r32, r64 in stages:
        hip = hip.reshape(r32.shape)
        e = PU.close(hip, r32, f"{name}: {what}")                                    # 1e-5, no attribution needed
        e_hip, e_cpu = PU.relerr(hip, r64), PU.relerr(r32, r64)
        assert e_hip <= e_cpu + PU.ATTR, f"{name}: {what}: |hip - f64| {e_hip:.2e} vs |cpu32 - f64| {e_cpu:.2e}"
        if e > worst[1]:
            worst = (what, e)
    PU.elementwise(y_fused, y32, f"{name}: y element-wise")
    # element-wise (not only in the max norm) for the intermediate stages too: phi(v) + phi(-v), the sign_net output, the layer-path y
    for what, hip, r32, r64 in (("phi (fused stage)", st["phi_fused"], o32["phi"], o64["phi"]), ("sign_net output", st["pos"], o32["pos"], o64["pos"]),
                                ("y (layer kernels)", y_layer, y32, y64)):
        n_attr = PU.elementwise(hip, r32, f"{name}: {what} element-wise", ref64=r64)
        print(f"\n{name}: {what}: {n_attr} of {r32.numel()} elements attributed to the fp32 reference")
    print(f"\n{name}: worst stage '{worst[0]}' max|hip - cpu32| / max|cpu32| = {worst[1]:.2e}")


def test_basisnet_real_grid_all_multiplicity_groups():
    """BASELINE configs[4] at full size: the 32x32 grid's 513 eigenspaces (2.15 GB of projectors on the device).  HIP
    IGNBasisInv on every group in full; the CPU oracle (fp32 and float64) on the whole mult-32 and mult-1 groups and on a
    64-projector subset of the mult-2 group — in eval mode BatchNorm uses running statistics, so every projector's output
    is independent of the rest of its group and the subset pins those rows exactly."""
    from oracle import basisnet as OB
    from signnet_basisnet_amd import basisnet as BN
    from signnet_basisnet_amd import synth
    import numpy as np
    ei, n = synth.grid_graph(32)
    A = np.zeros((n, n))
    A[ei[0], ei[1]] = 1.0
    deg = A.sum(1)
    Lap = np.eye(n) - A / np.sqrt(deg)[:, None] / np.sqrt(deg)[None, :]
    D, V = torch.linalg.eigh(torch.from_numpy(Lap))                       # float64 eigh on the host, then .float() (utils.py:72-78)
    groups64, counts = OB.group_eigenspaces(D.float(), V.float())
    assert sorted(groups64) == [1, 2, 32] and [groups64[k].shape[0] for k in (1, 2, 32)] == [32, 480, 1]
    torch.manual_seed(0)
    net = BN.IGNBasisInv([1, 2, 32], 1, hidden_channels=32)
    PU.bn_randomize(net, 2)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    net = net.cuda().eval()
    worst = 0.0
    for mult in (1, 2, 32):
        i = net.mult_to_idx[mult]
        enc = net.encs[i]
        eq = [(e.coeffs.detach().cpu(), e.bias.detach().cpu()) for e in enc.equi_layers]
        P = groups64[mult]
        with torch.no_grad():
            y = net(P.cuda(), mult).cpu()                                  # [b, mult, n], every projector of the group
        assert y.shape == (P.shape[0], mult, n)
        sel = torch.arange(P.shape[0]) if P.shape[0] <= 64 else torch.linspace(0, P.shape[0] - 1, 64).long()
        sub = {k[len(f"encs.{i}."):]: v for k, v in sd.items() if k.startswith(f"encs.{i}.")}
        with torch.no_grad():
            r32 = OB.ign2to1(sub, eq, P[sel], training=False)
            r64 = OB.ign2to1(PU.to_f64(sub), [(a.double(), b.double()) for a, b in eq], P[sel].double(), training=False)
        e = PU.close(y[sel], r32, f"IGN2to1 mult {mult}", ref64=r64)
        PU.elementwise(y[sel], r32, f"IGN2to1 mult {mult} element-wise", ref64=r64)
        e_hip, e_cpu = PU.relerr(y[sel], r64), PU.relerr(r32, r64)
        assert e_hip <= e_cpu + PU.ATTR, f"mult {mult}: |hip - f64| {e_hip:.2e} vs |cpu32 - f64| {e_cpu:.2e}"
        worst = max(worst, e)
        # rows outside the subset: batch independence (a projector evaluated alone gives the same rows)
        if P.shape[0] > 64:
            j = int(P.shape[0] // 2 + 1)
            with torch.no_grad():
                y1 = net(P[j:j + 1].cuda(), mult).cpu()
            assert torch.equal(y1[0], y[j]), "projector output depends on the rest of the group"
    print(f"\nBasisNet 32x32 grid: worst max|hip - cpu32| / max|cpu32| = {worst:.2e}")


# ---------------------------------------------------------------------------------------------------------------
# SURVEY.md §8 f3 at the SHIPPED sizes: the net_params of GraphPrediction/configs/*/*_ZINC_LapPE_signinv_GIN.json (hidden width, depth,
# heads / towers, k, readout, pe_aggregate; sign_inv_net = GINDeepSigns with 8 layers, phi_out 4) on a 128-graph ZINC-like batch
# (train_ZINC_graph_regression.py:20-25,77-80: sign_inv_net, then the base net).
from signnet_basisnet_amd.dgl_configs import COMMON as _COMMON, SHIPPED  # noqa: E402


@pytest.mark.parametrize("name", list(SHIPPED))
def test_dgl_base_nets_at_the_shipped_configs_vs_oracle_fp32_and_fp64(name):
    run_shipped_dgl_config(name)


def run_shipped_dgl_config(name, data=None, elementwise=True):
    """`data`: the batch (default: 128 ZINC-like molecules); tests/test_topology_gpu.py passes graphs the molecule generator never makes."""
    from oracle import dgl_deepsigns as OD
    from oracle import dgl_nets as ON
    from signnet_basisnet_amd import dgl_deepsigns as DS
    from signnet_basisnet_amd import dgl_nets, synth
    cfg = dict(SHIPPED[name])
    cls = getattr(dgl_nets, cfg.pop("cls"))
    params = dict(_COMMON, device="cuda:0")
    params.update(cfg)
    masked = params["sign_inv_net"] == "masked_gin"
    base = name.split("_")[0]
    torch.manual_seed(3)
    net = cls(params)
    PU.bn_randomize(net, 2)
    with torch.no_grad():                                   # GATConv's bias is zero-initialised: make it count
        for n_, p_ in net.named_parameters():
            if base == "gat" and n_.startswith("layers.") and n_.endswith(".bias") and n_.count(".") == 2:
                p_.copy_(0.1 * torch.randn(p_.shape, generator=torch.Generator().manual_seed(5)))
    sd32 = {k: v.detach().clone() for k, v in net.state_dict().items()}
    sd64 = PU.to_f64(sd32)
    if data is None:
        data = synth.make_batch(128, seed=4321)
    k, L = cfg["pos_enc_dim"], cfg["L"]
    pe = synth.dgl_pos_enc(data, k)
    src, dst = data.edge_index
    hx, ex, sizes = data.x.squeeze(-1), data.edge_attr, data.sizes
    sn = torch.cat([torch.full((n, 1), 1.0 / n) for n in sizes]).sqrt()          # data/molecules.py:307-308 (PNA's graph_norm)

    def oracle(sd, dt):
        ssd = {kk[len("sign_inv_net."):]: v for kk, v in sd.items() if kk.startswith("sign_inv_net.")}
        if masked:
            p = OD.masked_gin_deepsigns(ssd, src, dst, torch.tensor(sizes), pe.to(dt).unsqueeze(-1), 8, k).squeeze(-1)
        else:
            p = OD.gin_deepsigns(ssd, src, dst, pe.to(dt).unsqueeze(-1), 8, k).squeeze(-1)
        out = {}
        if base == "gin":
            y = ON.gin_net(sd, src, dst, sizes, hx, p, L, "mean")
        elif base == "gatedgcn":
            y = ON.gatedgcn_net(sd, src, dst, sizes, hx, p, ex, L, "concat", "mean", out=out)
        elif base == "gat":
            y = ON.gat_net(sd, src, dst, sizes, hx, p, L, cfg["n_heads"], "mean", out=out)
        elif base == "pna":
            y = ON.pna_net(sd, src, dst, sizes, hx, p, ex, sn.to(dt), L, cfg["towers"], float(cfg["avg_d"]["log"]), "sum", out=out)
        else:
            y = ON.transformer_net(sd, src, dst, sizes, hx, p, ex, L, cfg["n_heads"], "concat", "sum", out=out)
        return p, out.get("h_last"), y
    with torch.no_grad():
        p32, h32, y32 = oracle(sd32, torch.float32)
        p64, h64, y64 = oracle(sd64, torch.float64)
    net = net.cuda().eval()
    bne = torch.bincount(torch.bucketize(dst, torch.cumsum(torch.tensor(sizes), 0), right=True), minlength=len(sizes))
    g = DS.Graph(src.cuda(), dst.cuda(), sizes, bne)           # per-graph node AND edge counts, as a DGL batch carries them
    with torch.no_grad():
        p = net.sign_inv_net(g, pe.unsqueeze(-1).cuda()).squeeze(-1)
        y, _ = net(g, hx.cuda(), p, ex.cuda(), sn.cuda() if base == "pna" else None)
    if hasattr(net, "check_last"):
        net.check_last()
    stages = [("sign_inv_net output", p, p32, p64), ("scores", y, y32, y64)]
    if h32 is not None and getattr(net, "_h_last", None) is not None:
        stages.insert(1, ("node features after the last layer", net._h_last, h32, h64))
    for what, hip, r32, r64 in stages:
        e = PU.close(hip, r32, f"{name}: {what}", ref64=r64)
        if elementwise:
            PU.elementwise(hip, r32, f"{name}: {what} element-wise", ref64=r64)
        print(f"\n{name}: {what}: max|hip - cpu32| / max|cpu32| = {e:.2e}, |hip - f64| {PU.relerr(hip, r64):.2e}, |cpu32 - f64| {PU.relerr(r32, r64):.2e}")


@pytest.mark.parametrize("balance,k", [("count", 16), ("rows", None)])
def test_config4_global_batch_sharded_eight_ways_vs_the_oracle_of_the_whole_batch(balance, k):
    """BASELINE configs[3] on one GPU: the 1 024-graph ZINC batch cut into the eight shards `bench.py --gpus 8` gives its ranks
    (`dist.shard_batch`: by graph count with k = 16, by (node, slot) rows with all eigenvectors), the HIP forward of every shard, the
    concatenation against the CPU oracle of the WHOLE batch at north_star's 1e-5 — SURVEY.md section 8(e): no data-path collective,
    a rank never sees another rank's graphs; the module's default (strict) mode and the serving mode give the same bits."""
    from oracle import pyg_signnet as O
    from signnet_basisnet_amd import dist as D
    from signnet_basisnet_amd import synth
    from signnet_basisnet_amd.pyg import SignNetGNN
    ctor = (None, None, 128, 1, 4, 6)
    torch.manual_seed(0)
    m = SignNetGNN(*ctor, variant="gine", max_k=k)
    PU.bn_randomize(m, 1)
    world = 8
    data = synth.make_batch(128 * world if k else 256, seed=1234 + 2, n_lo=9, n_hi=37)    # (all eigenvectors: 256 graphs keep the CPU oracle short)
    cfg = O.make_cfg("gine", *ctor)
    sd = {kk: v.detach().clone() for kk, v in m.state_dict().items()}
    with torch.no_grad():
        y_ref = O.signnet_gnn(sd, cfg, data, training=False, max_k=k)
    m = m.cuda().eval()
    shards = [D.shard_batch(data, r, world, balance=balance, max_k=k) for r in range(world)]
    assert sum(s.num_graphs for s in shards) == data.num_graphs and all(s.num_graphs >= 1 for s in shards)
    if balance == "count":
        assert {s.num_graphs for s in shards} == {128}
    else:
        rows = [sum(n * n for n in s.sizes) for s in shards]
        assert max(rows) <= 1.25 * (sum(rows) / world), rows                # rows-balanced: no rank much above the mean
    outs = {}
    for strict in (True, False):
        m.strict = strict
        ys = []
        with torch.no_grad():
            for s in shards:
                ys.append(m(synth.batch_to(s, "cuda:0")))
            m.check_last()
        outs[strict] = torch.cat(ys, 0)
    assert torch.equal(outs[True], outs[False])
    PU.close(outs[True], y_ref, f"config 4 ({balance}): concatenated shard outputs vs the oracle of the whole batch")
    PU.elementwise(outs[True], y_ref, f"config 4 ({balance}): y element-wise")
