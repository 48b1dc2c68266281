"""-m gpu: DGL-tree DeepSigns and BasisNet modules (HIP) against the reference fixtures and the oracle."""
import pytest
import torch

import golden_util as G
import parity_util as PU
from parity_util import close      # max|hip - ref| <= 1e-5 * max|ref| (north_star); float64 attribution when `ref64` is given

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("name", ["dgl_gin_k8", "dgl_masked_k10"])
def test_deepsigns_golden(name):
    from signnet_basisnet_amd import dgl_deepsigns as DS
    fx = G.load(name)
    hidden, c, layers, k = (int(v) for v in fx.meta["params"])
    kind = str(fx.meta["kind"])
    net = DS.get_sign_inv_net(dict(sign_inv_net=kind, hidden_dim=hidden, phi_out_dim=c, sign_inv_layers=layers,
                                   pos_enc_dim=k, dropout=0.0, sign_inv_activation="relu", device=DEV))
    assert sorted(net.state_dict().keys()) == sorted(str(s) for s in fx.meta["sd_keys"])
    net.load_state_dict(fx.sd)
    net = net.to(DEV).eval()
    ei = fx.inp["edge_index"]
    g = DS.Graph(ei[0].to(DEV), ei[1].to(DEV), fx.inp["sizes"])
    y = net(g, fx.inp["pos_enc"].unsqueeze(-1).to(DEV))
    from oracle import dgl_deepsigns as OD
    x64 = fx.inp["pos_enc"].unsqueeze(-1).double()
    with torch.no_grad():
        if kind == "gin":
            r64 = OD.gin_deepsigns(PU.to_f64(fx.sd), ei[0], ei[1], x64, layers, k)
        else:
            r64 = OD.masked_gin_deepsigns(PU.to_f64(fx.sd), ei[0], ei[1], fx.inp["sizes"], x64, layers, k)
    close(y, fx.out["eval/y"], kind, ref64=r64)


@pytest.mark.parametrize("kind,k,hidden,c", [("gin", 8, 95, 4), ("gin", 16, 64, 4), ("masked_gin", 37, 67, 67)])
def test_deepsigns_shipped_sizes_vs_oracle(kind, k, hidden, c):
    """(K, hidden, phi_out) of the shipped configs, 8 layers, vs the CPU oracle."""
    from oracle import dgl_deepsigns as OD
    from signnet_basisnet_amd import dgl_deepsigns as DS
    from signnet_basisnet_amd import synth
    torch.manual_seed(0)
    net = DS.get_sign_inv_net(dict(sign_inv_net=kind, hidden_dim=hidden, phi_out_dim=c, sign_inv_layers=8, pos_enc_dim=k,
                                   dropout=0.0, sign_inv_activation="relu", device=DEV))
    gen = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=gen) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=gen) + 0.5)
    data = synth.make_batch(12, seed=41)
    pe = synth.dgl_pos_enc(data, k)
    sd = {kk: v.clone() for kk, v in net.state_dict().items()}
    ei = data.edge_index
    x = pe.unsqueeze(-1)
    with torch.no_grad():
        if kind == "gin":
            ref = OD.gin_deepsigns(sd, ei[0], ei[1], x, 8, k)
            r64 = OD.gin_deepsigns(PU.to_f64(sd), ei[0], ei[1], x.double(), 8, k)
        else:
            ref = OD.masked_gin_deepsigns(sd, ei[0], ei[1], torch.tensor(data.sizes), x, 8, k)
            r64 = OD.masked_gin_deepsigns(PU.to_f64(sd), ei[0], ei[1], torch.tensor(data.sizes), x.double(), 8, k)
    net = net.to(DEV).eval()
    y = net(DS.Graph(ei[0].to(DEV), ei[1].to(DEV), torch.tensor(data.sizes)), x.to(DEV))
    close(y, ref, kind, ref64=r64)


@pytest.mark.parametrize("kind,k,hidden,c", [("gin", 8, 95, 4), ("gin", 16, 64, 4), ("masked_gin", 37, 67, 67)])
def test_deepsigns_eval_is_three_launches_and_matches_the_layer_path(kind, k, hidden, c):
    """The eval forward of the shipped configurations is sn_batch_plan + sn_deepsigns_phi_f32 + sn_mlp_chain_f32 (VERDICT r01 item 5:
    <= 4 launches), and gives what the layer-at-a-time path gives, on a 128-graph batch (padded eigenvector columns of graphs
    smaller than k included)."""
    from signnet_basisnet_amd import dgl_deepsigns as DS
    from signnet_basisnet_amd import ops, synth
    torch.manual_seed(1)
    net = DS.get_sign_inv_net(dict(sign_inv_net=kind, hidden_dim=hidden, phi_out_dim=c, sign_inv_layers=8, pos_enc_dim=k,
                                   dropout=0.0, sign_inv_activation="relu", device=DEV))
    gen = torch.Generator().manual_seed(4)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=gen) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=gen) + 0.5)
                m.weight.copy_(1 + 0.2 * torch.randn(m.weight.shape, generator=gen))
                m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=gen))
    net = net.to(DEV).eval()
    data = synth.make_batch(128, seed=43)
    x = synth.dgl_pos_enc(data, k).unsqueeze(-1).to(DEV)
    ei = data.edge_index
    g = DS.Graph(ei[0].to(DEV), ei[1].to(DEV), torch.tensor(data.sizes))
    net(g, x)                                            # packs the parameters
    g = DS.Graph(ei[0].to(DEV), ei[1].to(DEV), torch.tensor(data.sizes))      # a fresh batch object: its plan is built once, kept on it
    rec = ops.KernelTimer()
    with rec:
        y = net(g, x)
        y = net(g, x)
    names = [n for n, _, _ in rec.spans]
    assert names == ["sn_batch_plan", "sn_deepsigns_phi_f32", "sn_mlp_chain_f32", "sn_deepsigns_phi_f32", "sn_mlp_chain_f32"], names
    net.fused_stages = False
    y_layers = net(g, x)
    net.fused_stages = True
    assert torch.isfinite(y).all()
    close(y, y_layers, f"{kind} stage kernels vs layer path")


def test_ign_contractions_vs_fp64():
    from signnet_basisnet_amd import ops
    g = torch.Generator().manual_seed(0)
    for b, n in ((3, 36), (2, 100), (1, 1024), (5, 300)):
        X = torch.randn(b, 1, n, n, generator=g)
        o = ops.ign_contract_2to1(X.to(DEV)).cpu().double()
        Xd = X[:, 0].double()
        ref = torch.stack([torch.diagonal(Xd, dim1=1, dim2=2), Xd.diagonal(dim1=1, dim2=2).sum(1, keepdim=True).expand(-1, n) / n,
                           Xd.sum(2) / n, Xd.sum(1) / n, Xd.sum((1, 2)).unsqueeze(1).expand(-1, n) / n ** 2], dim=2)
        torch.testing.assert_close(o, ref, rtol=1e-5, atol=1e-5)


def test_basisnet_golden():
    from oracle import basisnet as OB
    from signnet_basisnet_amd import basisnet as BN
    fx = G.load("basisnet_grid6")
    D, V = fx.inp["eigvals"], fx.inp["eigvecs"]
    N = V.shape[0]
    groups = BN.group_eigenspaces(D.to(DEV), V.to(DEV))          # device op (sn_eigenspace_group + sn_eigenspace_projectors_f32)
    mults = [int(m) for m in fx.meta["mults"]]
    assert sorted(groups) == mults
    hidden = int(fx.meta["hidden"])
    net = BN.IGNBasisInv(mults, 1, hidden_channels=hidden)
    outs = []
    for m in mults:
        enc = net.encs[net.mult_to_idx[m]]
        sd = {k.split("/", 1)[1]: v for k, v in fx.sd.items() if k.startswith(f"enc{m}/")}
        assert sorted(enc.state_dict().keys()) == sorted(sd.keys())
        enc.load_state_dict(sd)
    net = net.to(DEV).eval()
    for m in mults:
        o = net(groups[m], m)
        sdm = {k.split("/", 1)[1]: v for k, v in fx.sd.items() if k.startswith(f"enc{m}/")}
        eq64 = [(fx.eq[f"enc{m}/{i}/coeffs"].double(), fx.eq[f"enc{m}/{i}/bias"].double()) for i in range(3)]
        with torch.no_grad():
            r64 = OB.ign2to1(PU.to_f64(sdm), eq64, groups[m].cpu().double(), training=False)
        close(o, fx.out[f"eval/phi_m{m}"], f"IGN2to1 mult {m}", ref64=r64)
        outs.append(o.cpu())
    # rho on the reference's own phi outputs (identical inputs on both sides; the phi stage is checked above).  Its BatchNorm has
    # track_running_stats=False: batch statistics over 36 rows are ill-conditioned, both fp32 evaluations sit ~1e-5 from the
    # exact value — hence the float64 attribution inside close()
    feats = OB.basis_inv_features([fx.out[f"eval/phi_m{m}"] for m in mults], D, N)   # reshape/concat bookkeeping of training.py:119-123
    rho = BN.EqDeepSetsEncoder(2 * N, hidden_channels=10, out_channels=8, num_layers=3, use_bn=True)
    rho_sd = {k.split("/", 1)[1]: v for k, v in fx.sd.items() if k.startswith("rho/")}
    rho.load_state_dict(rho_sd)
    y = rho.to(DEV)(feats.to(DEV))
    with torch.no_grad():
        y64 = OB.eq_deepsets(PU.to_f64(rho_sd), feats.double(), 3, True)
    close(y, fx.out["eval/rho"], "EqDeepSetsEncoder rho", ref64=y64)
    # and the chained value (HIP phi outputs -> HIP rho) stays in the same class
    y2 = rho(OB.basis_inv_features(outs, D, N).to(DEV))
    assert PU.relerr(y2, y64) <= 1e-4


def test_signplus_deepsets_golden():
    from signnet_basisnet_amd import basisnet as BN
    fx = G.load("basisnet_grid6")
    sign = BN.SignPlus(BN.EqDeepSetsEncoder(1, num_layers=3, use_bn=True))
    sign.load_state_dict({k.split("/", 1)[1]: v for k, v in fx.sd.items() if k.startswith("sign/")})
    v = fx.inp["eigvecs"].transpose(1, 0).unsqueeze(-1).contiguous()
    y = sign.to(DEV)(v.to(DEV))
    from oracle import basisnet as OB
    sgn_sd = {k.split("/", 1)[1]: v_ for k, v_ in fx.sd.items() if k.startswith("sign/")}
    with torch.no_grad():
        y64 = OB.sign_plus_deepsets(PU.to_f64(sgn_sd), v.double(), 3, True)
    close(y, fx.out["eval/signplus"], "SignPlus(DeepSets)", ref64=y64)
    # invariance to a global sign flip is exact: it swaps the two addends (the batch-statistics BatchNorm over the
    # stack of eigenvectors makes per-eigenvector flips only approximately invariant — in the reference too)
    assert torch.equal(y, sign(-v.to(DEV)))


def test_signplus_with_side_features_vs_oracle():
    """SignPlus.forward(v, x=...) — signbasisnet.py:19-20: negate v, do not negate x (unreachable from the reference's entry script,
    which is why rounds 1-3 refused it; part of the module's surface all the same)."""
    from oracle import basisnet as OB
    from signnet_basisnet_amd import basisnet as BN
    torch.manual_seed(4)
    sign = BN.SignPlus(BN.EqDeepSetsEncoder(1 + 3, num_layers=3, use_bn=True))
    sd = {k: v.detach().clone() for k, v in sign.state_dict().items()}
    g = torch.Generator().manual_seed(6)
    v, x = torch.randn(36, 36, 1, generator=g), torch.randn(36, 36, 3, generator=g)
    with torch.no_grad():
        y32 = OB.sign_plus_deepsets(sd, v, 3, True, x=x)
        y64 = OB.sign_plus_deepsets(PU.to_f64(sd), v.double(), 3, True, x=x.double())
    sign = sign.to(DEV)
    y = sign(v.to(DEV), x=x.to(DEV))
    close(y, y32, "SignPlus(DeepSets) with side features", ref64=y64)
    assert torch.equal(y, sign(-v.to(DEV), x=x.to(DEV)))          # a global flip of v swaps the two addends; x is not negated


@pytest.mark.parametrize("name", ["dgl_gin_k8", "dgl_masked_k10"])
def test_deepsigns_train_mode_forward(name):
    """net.train(): BatchNorm with batch statistics over all N*K rows (gnns.py:105-112, mlp.py:44-50) against the
    reference's train-mode fixture (tolerance of tests/test_oracle_golden.py::TOL_BS), plus the running-stat side effect."""
    from signnet_basisnet_amd import dgl_deepsigns as DS
    fx = G.load(name)
    hidden, c, layers, k = (int(v) for v in fx.meta["params"])
    kind = str(fx.meta["kind"])
    net = DS.get_sign_inv_net(dict(sign_inv_net=kind, hidden_dim=hidden, phi_out_dim=c, sign_inv_layers=layers,
                                   pos_enc_dim=k, dropout=0.0, sign_inv_activation="relu", device=DEV))
    net.load_state_dict(fx.sd)
    net = net.to(DEV).train()
    ei = fx.inp["edge_index"]
    g = DS.Graph(ei[0].to(DEV), ei[1].to(DEV), fx.inp["sizes"])
    with torch.no_grad():
        y = net(g, fx.inp["pos_enc"].unsqueeze(-1).to(DEV))
    torch.testing.assert_close(y.cpu(), fx.out["train/y"], rtol=5e-4, atol=5e-5)
    moved = [kk for kk, v in net.state_dict().items() if kk.endswith("running_mean") and not torch.equal(v.cpu(), fx.sd[kk])]
    assert len(moved) == sum(1 for kk in fx.sd if kk.endswith("running_mean")), moved
    y_eval = net.eval()(g, fx.inp["pos_enc"].unsqueeze(-1).to(DEV))      # new running statistics are picked up
    assert not torch.allclose(y_eval.cpu(), fx.out["eval/y"], rtol=1e-3, atol=1e-4)


def test_ign_train_mode_forward():
    """IGN2to1.train(): BatchNorm1d(hidden) with batch statistics over the b*n rows of [b, hidden, n] (ign.py:31-33)."""
    from signnet_basisnet_amd import basisnet as BN
    fx = G.load("basisnet_grid6")
    groups = BN.group_eigenspaces(fx.inp["eigvals"].to(DEV), fx.inp["eigvecs"].to(DEV))
    mults = [int(m) for m in fx.meta["mults"]]
    net = BN.IGNBasisInv(mults, 1, hidden_channels=int(fx.meta["hidden"]))
    for m in mults:
        enc = net.encs[net.mult_to_idx[m]]
        enc.load_state_dict({k.split("/", 1)[1]: v for k, v in fx.sd.items() if k.startswith(f"enc{m}/")})
    net = net.to(DEV).train()
    for m in mults:
        with torch.no_grad():
            o = net(groups[m], m)
        torch.testing.assert_close(o.cpu(), fx.out[f"train/phi_m{m}"], rtol=2e-3, atol=2e-4)
        enc = net.encs[net.mult_to_idx[m]]
        assert int(enc.bns[0].num_batches_tracked) == int(fx.sd[f"enc{m}/bns.0.num_batches_tracked"]) + 1


def test_ign_shared_vs_oracle():
    """IGNShared (signbasisnet.py:43-64): the shared IGN2to1(1, hidden, 1) followed by a per-multiplicity Linear(1, mult)."""
    from oracle import basisnet as OB
    from signnet_basisnet_amd import basisnet as BN
    fx = G.load("basisnet_grid6")
    groups = BN.group_eigenspaces(fx.inp["eigvals"].to(DEV), fx.inp["eigvecs"].to(DEV))
    mults = sorted(groups)
    torch.manual_seed(3)
    net = BN.IGNShared(mults, 1, hidden_channels=8)
    gen = torch.Generator().manual_seed(4)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=gen) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=gen) + 0.5)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    eq = [(net.enc.equi_layers[i].coeffs.detach().clone(), net.enc.equi_layers[i].bias.detach().clone()) for i in range(3)]
    net = net.to(DEV).eval()
    for m in mults:
        with torch.no_grad():
            ref = OB.ign_shared(sd, eq, groups[m].cpu(), net.mult_to_idx[m])
            r64 = OB.ign_shared(PU.to_f64(sd), [(a.double(), b.double()) for a, b in eq], groups[m].cpu().double(), net.mult_to_idx[m])
        y = net(groups[m], m)
        assert y.shape == ref.shape == (groups[m].shape[0], m, groups[m].shape[-1])
        close(y, ref, f"IGNShared mult {m}", ref64=r64)


def _ginnet(fx):
    from signnet_basisnet_amd import dgl_nets
    hidden, L, k = (int(v) for v in fx.meta["params"])
    params = dict(num_atom_type=28, num_bond_type=4, hidden_dim=hidden, out_dim=hidden, in_feat_dropout=0.0, dropout=0.0, L=L,
                  readout="mean", batch_norm=True, residual=True, edge_feat=True, device=DEV, pe_init="lap_pe",
                  lap_method="sign_inv", lap_lspe=False, use_lapeig_loss=False, lambda_loss=1, alpha_loss=1e-4, pos_enc_dim=k,
                  sign_inv_net="gin", sign_inv_layers=3, sign_inv_activation="relu", pe_aggregate="add", phi_out_dim=4)
    net = dgl_nets.GINNet(params)
    assert sorted(net.state_dict().keys()) == sorted(fx.sd.keys())
    net.load_state_dict(fx.sd)
    return net.to(DEV)


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_dgl_gin_base_net_golden(mode):
    """GraphPrediction tree end to end: sign_inv_net -> GINNet (gin_net.py), driven like train_ZINC_graph_regression.py:20-25,
    against the reference's own outputs (state_dict keys identical)."""
    from signnet_basisnet_amd import dgl_deepsigns as DS
    fx = G.load("dgl_ginnet_k6")
    net = _ginnet(fx).train(mode == "train")
    ei = fx.inp["edge_index"]
    g = DS.Graph(ei[0].to(DEV), ei[1].to(DEV), fx.inp["sizes"])
    with torch.no_grad():
        p = net.sign_inv_net(g, fx.inp["pos_enc"].unsqueeze(-1).to(DEV)).squeeze(-1)
        y, _ = net(g, fx.inp["x"].squeeze(-1).to(DEV), p, fx.inp["edge_attr"].to(DEV), None)
    if mode == "eval":
        close(p, fx.out["eval/p"], "sign_inv_net output")
        close(y, fx.out["eval/y"], "GINNet scores")
    else:
        torch.testing.assert_close(p.cpu(), fx.out["train/p"], rtol=5e-4, atol=5e-5)
        torch.testing.assert_close(y.cpu(), fx.out["train/y"], rtol=5e-4, atol=5e-5)


def _gatedgcn(fx):
    from signnet_basisnet_amd import dgl_nets
    hidden, L, k = (int(v) for v in fx.meta["params"])
    params = dict(num_atom_type=28, num_bond_type=4, hidden_dim=hidden, out_dim=hidden, in_feat_dropout=0.0, dropout=0.0, L=L,
                  readout="mean", batch_norm=True, residual=True, edge_feat=True, device=DEV, pe_init="lap_pe",
                  lap_method="sign_inv", lap_lspe=False, use_lapeig_loss=False, lambda_loss=1, alpha_loss=1e-4, pos_enc_dim=k,
                  sign_inv_net="masked_gin", sign_inv_layers=3, sign_inv_activation="relu",
                  pe_aggregate=str(fx.meta["pe_aggregate"]), phi_out_dim=4)
    net = dgl_nets.GatedGCNNet(params)
    assert sorted(net.state_dict().keys()) == sorted(fx.sd.keys())
    net.load_state_dict(fx.sd)
    return net.to(DEV)


@pytest.mark.parametrize("name", ["dgl_gatedgcn_concat_k6", "dgl_gatedgcn_add_k8"])
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_dgl_gatedgcn_base_net_golden(name, mode):
    """GraphPrediction tree, GatedGCN_ZINC_LapPE_signinv_GIN_mask.json's model scaled down: MaskedGINDeepSigns -> GatedGCNNet
    (gatedgcn_net.py + gatedgcn_layer.py) against the reference's own outputs."""
    from signnet_basisnet_amd import dgl_deepsigns as DS
    fx = G.load(name)
    net = _gatedgcn(fx).train(mode == "train")
    ei = fx.inp["edge_index"]
    g = DS.Graph(ei[0].to(DEV), ei[1].to(DEV), fx.inp["sizes"])
    with torch.no_grad():
        p = net.sign_inv_net(g, fx.inp["pos_enc"].unsqueeze(-1).to(DEV)).squeeze(-1)
        y, _ = net(g, fx.inp["x"].squeeze(-1).to(DEV), p, fx.inp["edge_attr"].to(DEV), None)
    if mode == "eval":
        close(p, fx.out["eval/p"], "sign_inv_net output")
        close(y, fx.out["eval/y"], "GatedGCNNet scores")
    else:
        torch.testing.assert_close(p.cpu(), fx.out["train/p"], rtol=1e-3, atol=1e-4)
        torch.testing.assert_close(y.cpu(), fx.out["train/y"], rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("pe_aggregate,hidden,readout", [("concat", 68, "mean"), ("add", 52, "sum"), ("add", 28, "mean")])
def test_gatedgcn_one_launch_matches_the_layer_path(pe_aggregate, hidden, readout):
    """Eval forward of the shipped GatedGCN shape (hidden 68, 16 layers) and two others on a 128-graph batch: layers + readout are
    ONE launch (sn_gatedgcn_fused_f32) and give what the layer-at-a-time path gives; with the sign-invariant net in front the
    whole model is 8 launches."""
    from signnet_basisnet_amd import dgl_nets as DN
    from signnet_basisnet_amd import dgl_deepsigns as DS
    from signnet_basisnet_amd import ops, synth
    k = 8
    torch.manual_seed(5)
    p = dict(num_atom_type=28, num_bond_type=4, hidden_dim=hidden, out_dim=hidden, L=16, readout=readout, batch_norm=True, residual=True,
             edge_feat=True, device=DEV, pe_init="lap_pe", lap_method="sign_inv", lap_lspe=False, use_lapeig_loss=False, lambda_loss=0.0,
             alpha_loss=0.0, pos_enc_dim=k, pe_aggregate=pe_aggregate, in_feat_dropout=0.0, dropout=0.0, sign_inv_net="gin",
             sign_inv_layers=8, phi_out_dim=4, sign_inv_activation="relu")
    net = DN.GatedGCNNet(p)
    gen = torch.Generator().manual_seed(6)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=gen) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=gen) + 0.5)
                m.weight.copy_(1 + 0.2 * torch.randn(m.weight.shape, generator=gen))
                m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=gen))
    net = net.to(DEV).eval()
    data = synth.make_batch(128, seed=44)
    ei = data.edge_index
    g = DS.Graph(ei[0].to(DEV), ei[1].to(DEV), torch.tensor(data.sizes))
    h = data.x.reshape(-1).to(DEV)
    e = data.edge_attr.reshape(-1).to(DEV)
    pe = synth.dgl_pos_enc(data, k).to(DEV)

    def run():
        pp = net.sign_inv_net(g, pe.unsqueeze(-1)).squeeze(-1)
        return net(g, h, pp, e)[0]
    run()
    rec = ops.KernelTimer()
    with rec:
        y = run()
    names = [n for n, _, _ in rec.spans]
    assert names.count("sn_gatedgcn_fused_f32") == 1 and "sn_gated_aggregate_f32" not in names and len(names) <= 9, names
    net.fused_stages = False
    y_layers = run()
    net.fused_stages = True
    assert y.shape == (128, 1) and torch.isfinite(y).all()
    close(y, y_layers, "GatedGCN one launch vs layer path")


@pytest.mark.parametrize("side", [6, 12, 32])
def test_eigenspace_grouping_device_op_vs_reference_statements(side):
    """SURVEY.md §8 a18 on the device: sn_eigenspace_group / sn_eigenspace_projectors_f32 against the fixture produced by EXECUTING
    LearningFilters/training.py:47-73 (tests/golden/make_golden.py::reference_grouping).  Integer results (multiplicities, grouping,
    stacking order) bit-exact; projector entries are fp32 dot products of <= 32 terms whose order differs from the CPU BLAS:
    1e-6 of the largest entry.  The projector-free contractions (sn_ign_contract_eigvecs_f32) against the same fixture."""
    from signnet_basisnet_amd import basisnet as BN
    from signnet_basisnet_amd import ops
    fx = G.load("basisnet_grouping")
    t = f"s{side}"
    D, V = fx.inp[f"{t}/eigvals"].to(DEV), fx.inp[f"{t}/eigvecs"].to(DEV)
    N = V.shape[0]
    groups, plan = BN.group_eigenspaces(D, V, return_plan=True)
    counts = fx.out[f"{t}/counts"]
    assert plan.mults == fx.out[f"{t}/mults"].tolist() and plan.n_spaces == counts.numel()
    assert plan.space_mult[:plan.n_spaces].cpu().tolist() == counts.tolist()
    assert plan.space_start[:plan.n_spaces + 1].cpu().tolist() == [0] + counts.cumsum(0).tolist()
    assert plan.space_of.cpu().tolist() == torch.repeat_interleave(torch.arange(counts.numel()), counts).tolist()
    contr = ops.ign_contract_eigvecs(V, plan)
    for m in plan.mults:
        P = groups[m]
        b = int((counts == m).sum())
        assert P.shape == (b, 1, N, N)
        sig = fx.out[f"{t}/sig_m{m}"].double()                                    # [b, N, 2]: diagonal, row sums (float64 of the reference projector)
        Pd = P[:, 0].double().cpu()
        if f"{t}/proj_m{m}" in fx.out:
            ref = fx.out[f"{t}/proj_m{m}"]
            scale = ref.abs().max().item()
            assert (P.cpu() - ref).abs().max().item() <= 1e-6 * scale
        dscale = sig[..., 0].abs().max().item()
        assert (torch.diagonal(Pd, dim1=1, dim2=2) - sig[..., 0]).abs().max().item() <= 2e-6 * dscale
        assert (Pd.sum(2) - sig[..., 1]).abs().max().item() <= 2e-5 * dscale * 1.0 + 1e-6      # a row sum adds N entries of size <= dscale
        # projector-free contractions: [diag, tr/n, rowsum/n, colsum/n, total/n^2]
        c = plan.group(contr, m).double().cpu()
        assert (c[..., 0] - sig[..., 0]).abs().max().item() <= 2e-6 * dscale
        assert (c[..., 2] - sig[..., 1] / N).abs().max().item() <= 2e-6 * dscale and torch.equal(c[..., 2], c[..., 3])
        assert (c[:, 0, 1] - fx.out[f"{t}/tr_m{m}"] / N).abs().max().item() <= 1e-5 * m / N
        assert (c[:, 0, 4] - fx.out[f"{t}/tot_m{m}"] / N ** 2).abs().max().item() <= 1e-6 * m / N
        # and they agree with the contraction kernel applied to the device-built projectors
        c2 = ops.ign_contract_2to1(P).double().cpu()
        assert (c - c2).abs().max().item() <= 2e-6 * dscale


def test_basisnet_from_eigenvectors_equals_projector_path():
    """IGNBasisInv.forward_eigvecs (contractions from V alone) against the reference API on the stacked projectors."""
    from signnet_basisnet_amd import basisnet as BN
    fx = G.load("basisnet_grouping")
    D, V = fx.inp["s12/eigvals"].to(DEV), fx.inp["s12/eigvecs"].to(DEV)
    groups, plan = BN.group_eigenspaces(D, V, return_plan=True)
    torch.manual_seed(0)
    net = BN.IGNBasisInv(plan.mults, 1, hidden_channels=16)
    PU.bn_randomize(net, 3)
    net = net.to(DEV).eval()
    with torch.no_grad():
        fast = net.forward_eigvecs(V, plan)
        for m in plan.mults:
            close(fast[m], net(groups[m], m), f"forward_eigvecs mult {m}", rel=1e-5)


def _base_params(fx, hidden, L, k, **kw):
    p = dict(num_atom_type=28, num_bond_type=4, hidden_dim=hidden, out_dim=hidden, in_feat_dropout=0.0, dropout=0.0, L=L, readout="sum",
             batch_norm=True, residual=True, edge_feat=True, device=DEV, pe_init="lap_pe", lap_method="sign_inv", lap_lspe=False,
             use_lapeig_loss=False, lambda_loss=1, alpha_loss=1e-4, pos_enc_dim=k, sign_inv_net="gin", sign_inv_layers=3,
             sign_inv_activation="relu", pe_aggregate="add", phi_out_dim=4)
    p.update(kw)
    return p


def _drive(net, fx, snorm=False):
    from signnet_basisnet_amd import dgl_deepsigns as DS
    ei = fx.inp["edge_index"]
    g = DS.Graph(ei[0].to(DEV), ei[1].to(DEV), fx.inp["sizes"])
    sn = fx.inp["snorm_n"].to(DEV) if snorm else None
    with torch.no_grad():
        p = net.sign_inv_net(g, fx.inp["pos_enc"].unsqueeze(-1).to(DEV)).squeeze(-1)          # handle_lap, sign_inv branch
        y, _ = net(g, fx.inp["x"].squeeze(-1).to(DEV), p, fx.inp["edge_attr"].to(DEV), sn)
    return p, y, net._h_last


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_dgl_pna_base_net_golden(mode):
    """GraphPrediction tree, PNA_ZINC_LapPE_signinv_GIN.json's model scaled down (4 aggregators x 3 scalers, 5 towers, edge features,
    graph_norm): GINDeepSigns -> PNANet (pna_net.py + pna_layer.py + pna_utils.py) against the reference's own outputs."""
    from oracle import dgl_nets as ON
    from signnet_basisnet_amd import dgl_nets
    fx = G.load("dgl_pna_k6")
    hidden, L, k, towers, edge_dim = (int(v) for v in fx.meta["params"])
    avg = fx.meta["avg_d"]
    net = dgl_nets.PNANet(_base_params(fx, hidden, L, k, graph_norm=True, aggregators="mean max min std",
                                       scalers="identity amplification attenuation",
                                       avg_d=dict(lin=float(avg[0]), exp=float(avg[1]), log=float(avg[2])), towers=towers,
                                       divide_input_first=True, divide_input_last=True, edge_dim=edge_dim, pretrans_layers=1,
                                       posttrans_layers=1, gru=False))
    assert sorted(net.state_dict().keys()) == sorted(fx.sd.keys())
    net.load_state_dict(fx.sd)
    net = net.to(DEV).train(mode == "train")
    p, y, h_last = _drive(net, fx, snorm=True)
    if mode == "eval":
        ei = fx.inp["edge_index"]
        with torch.no_grad():
            y64 = ON.pna_net(PU.to_f64(fx.sd), ei[0], ei[1], fx.inp["sizes"], fx.inp["x"].squeeze(-1), fx.out["eval/p"].double(),
                             fx.inp["edge_attr"], fx.inp["snorm_n"].double(), L, towers, float(avg[2]), "sum")
        close(p, fx.out["eval/p"], "sign_inv_net output")
        close(h_last, fx.out["eval/h_last"], "PNA node features")
        close(y, fx.out["eval/y"], "PNANet scores", ref64=y64)
    else:
        torch.testing.assert_close(p.cpu(), fx.out["train/p"], rtol=5e-4, atol=5e-5)
        torch.testing.assert_close(y.cpu(), fx.out["train/y"], rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("name", ["dgl_transformer_concat_k6", "dgl_transformer_add_k8"])
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_dgl_transformer_base_net_golden(name, mode):
    """GraphPrediction tree, Transformer_ZINC_LapPE_signinv_GIN.json's model scaled down: GINDeepSigns -> TransformerNet
    (transformer_net.py + layers/transformer.py: attention over the graph's edges with edge features) vs the reference's outputs."""
    from oracle import dgl_nets as ON
    from signnet_basisnet_amd import dgl_nets
    fx = G.load(name)
    hidden, L, k, heads = (int(v) for v in fx.meta["params"])
    agg = str(fx.meta["pe_aggregate"])
    net = dgl_nets.TransformerNet(_base_params(fx, hidden, L, k, n_heads=heads, full_graph=False, layer_norm=True, pe_aggregate=agg))
    assert sorted(net.state_dict().keys()) == sorted(fx.sd.keys())
    net.load_state_dict(fx.sd)
    net = net.to(DEV).train(mode == "train")
    p, y, h_last = _drive(net, fx)
    if mode == "eval":
        ei = fx.inp["edge_index"]
        with torch.no_grad():
            y64 = ON.transformer_net(PU.to_f64(fx.sd), ei[0], ei[1], fx.inp["sizes"], fx.inp["x"].squeeze(-1), fx.out["eval/p"].double(),
                                     fx.inp["edge_attr"], L, heads, agg, "sum")
        close(p, fx.out["eval/p"], "sign_inv_net output")
        close(h_last, fx.out["eval/h_last"], "Transformer node features")
        close(y, fx.out["eval/y"], "TransformerNet scores", ref64=y64)
    else:
        torch.testing.assert_close(p.cpu(), fx.out["train/p"], rtol=5e-4, atol=5e-5)
        torch.testing.assert_close(y.cpu(), fx.out["train/y"], rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_dgl_gat_base_net_golden(mode):
    """GraphPrediction tree, GAT_ZINC_LapPE_signinv_GIN.json's model scaled down: GINDeepSigns -> GATNet (gat_net.py on DGL's GATConv:
    per-head additive attention over the in-edges) against the reference's own outputs.  The net has no BatchNorm / dropout, so the
    train-mode value differs from eval only through the sign-invariant network's batch statistics."""
    from oracle import dgl_nets as ON
    from signnet_basisnet_amd import dgl_nets
    fx = G.load("dgl_gat_k6")
    hidden, L, k, heads = (int(v) for v in fx.meta["params"])
    net = dgl_nets.GATNet(_base_params(fx, hidden, L, k, n_heads=heads, readout="mean", pe_aggregate="concat"))
    assert sorted(net.state_dict().keys()) == sorted(fx.sd.keys())
    net.load_state_dict(fx.sd)
    net = net.to(DEV).train(mode == "train")
    p, y, h_last = _drive(net, fx)
    if mode == "eval":
        ei = fx.inp["edge_index"]
        with torch.no_grad():
            y64 = ON.gat_net(PU.to_f64(fx.sd), ei[0], ei[1], fx.inp["sizes"], fx.inp["x"].squeeze(-1), fx.out["eval/p"].double(), L, heads, "mean")
        close(p, fx.out["eval/p"], "sign_inv_net output")
        close(h_last, fx.out["eval/h_last"], "GAT node features")
        close(y, fx.out["eval/y"], "GATNet scores", ref64=y64)
    else:
        torch.testing.assert_close(p.cpu(), fx.out["train/p"], rtol=5e-4, atol=5e-5)
        torch.testing.assert_close(y.cpu(), fx.out["train/y"], rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("heads,C", [(4, 12), (1, 1), (4, 59), (3, 64), (8, 7)])
def test_gat_aggregate_vs_fp64(heads, C):
    """sn_gat_aggregate_f32 against the float64 restatement of GATConv's edge softmax + weighted sum on a random batch (in-degrees
    1..7), at the shipped head width (59), the kernel's maximum (64) and degenerate widths; lse = max + log(sum exp)."""
    from oracle import dgl_nets as ON
    from signnet_basisnet_amd import ops, synth
    data = synth.make_batch(40, seed=17)
    d = synth.batch_to(data, DEV)
    plan = ops.build_plan(d.batch, d.edge_index, d.num_graphs, 0)
    g = torch.Generator().manual_seed(heads * 100 + C)
    src, dst = data.edge_index
    N = data.num_nodes
    f = torch.randn(N, heads * C, generator=g)
    al, ar = torch.randn(1, heads, C, generator=g), torch.randn(1, heads, C, generator=g)
    b = torch.randn(heads * C, generator=g)
    out, lse = ops.gat_aggregate(f.to(DEV), al.to(DEV), ar.to(DEV), b.to(DEV), plan, heads, 0.2, relu=True, want_lse=True)
    sd = {"l.fc.weight": torch.eye(heads * C, dtype=torch.float64), "l.attn_l": al.double(), "l.attn_r": ar.double(), "l.bias": b.double()}
    ref = ON.gat_conv(sd, "l", src, dst, f.double(), heads).flatten(1)
    close(out, ref, "gat_aggregate")
    fd = f.double().view(N, heads, C)
    e = torch.nn.functional.leaky_relu((fd * al.double()).sum(-1)[src] + (fd * ar.double()).sum(-1)[dst], 0.2)
    ref_lse = torch.stack([torch.logsumexp(e[dst == n], 0) for n in range(N)])
    close(lse, ref_lse, "gat_aggregate lse")
    # without the activation and the bias: the plain attention-weighted sum
    out2 = ops.gat_aggregate(f.to(DEV), al.to(DEV), ar.to(DEV), None, plan, heads, 0.2, relu=False)
    sd["l.bias"] = torch.zeros(heads * C, dtype=torch.float64)
    z = fd.new_zeros(N, heads, C)
    w = torch.exp(e - ref_lse[dst])
    z.index_add_(0, dst, w.unsqueeze(-1) * fd[src])
    close(out2, z.flatten(1), "gat_aggregate (no bias, no relu)")


def test_gat_net_rejects_zero_in_degree_and_bad_atom_type():
    """DGL's GATConv raises on a graph with a node without in-edges (allow_zero_in_degree False, gat_net.py:62-66 keeps the default);
    an atom type outside the embedding table raises IndexError as nn.Embedding does."""
    from signnet_basisnet_amd import dgl_deepsigns as DS, dgl_nets
    fx = G.load("dgl_gat_k6")
    hidden, L, k, heads = (int(v) for v in fx.meta["params"])
    net = dgl_nets.GATNet(_base_params(fx, hidden, L, k, n_heads=heads, readout="mean", pe_aggregate="concat"))
    net.load_state_dict(fx.sd)
    net = net.to(DEV).eval()
    ei = fx.inp["edge_index"]
    x = fx.inp["x"].squeeze(-1).to(DEV)
    p = fx.out["eval/p"].to(DEV)
    keep = ei[1] != 0                                             # node 0 loses its in-edges
    g = DS.Graph(ei[0][keep].to(DEV), ei[1][keep].to(DEV), fx.inp["sizes"])
    with pytest.raises(ValueError, match="0-in-degree"):
        net(g, x, p, fx.inp["edge_attr"][keep].to(DEV))
    g = DS.Graph(ei[0].to(DEV), ei[1].to(DEV), fx.inp["sizes"])
    bad = x.clone()
    bad[3] = 28
    with pytest.raises(IndexError):
        net(g, bad, p, fx.inp["edge_attr"].to(DEV))
    y, _ = net(g, x, p, fx.inp["edge_attr"].to(DEV))
    close(y, fx.out["eval/y"], "GATNet scores after the rejected calls")


def test_pna_aggregate_and_edge_attention_vs_fp64():
    """The two new message-passing kernels against float64 restatements on a larger random batch (degrees 1..7)."""
    from oracle import dgl_nets as ON
    from signnet_basisnet_amd import ops, synth
    data = synth.make_batch(40, seed=13)
    d = synth.batch_to(data, DEV)
    plan = ops.build_plan(d.batch, d.edge_index, d.num_graphs, 0)
    g = torch.Generator().manual_seed(0)
    src, dst = data.edge_index
    E, N = src.numel(), data.num_nodes
    m, hs = torch.randn(E, 14, generator=g), torch.randn(N, 14, generator=g)
    out = ops.pna_aggregate(m.to(DEV), hs.to(DEV), plan, 1.3).cpu().double()
    ref = torch.cat([hs.double(), ON.pna_aggregate(m.double(), dst, N, 1.3)], dim=1)
    C = 14
    std_cols = torch.zeros(13 * C, dtype=torch.bool)
    for s_ in range(3):
        std_cols[C + (4 * s_ + 3) * C:C + (4 * s_ + 4) * C] = True
    torch.testing.assert_close(out[:, ~std_cols], ref[:, ~std_cols], rtol=1e-5, atol=1e-5)
    # std = sqrt(relu(E[x^2] - E[x]^2) + 1e-5) (pna_utils.py:28-36) cancels in fp32 — in the reference too: the variance carries an
    # absolute error of a few fp32 roundings of E[x^2], which the square root near 1e-5 amplifies; compare at that level
    deg = torch.bincount(dst, minlength=N).clamp(min=1).double().unsqueeze(1)
    ex2 = torch.zeros(N, C, dtype=torch.float64).index_add_(0, dst, m.double() ** 2) / deg
    tol = (4e-7 * ex2 / (2 * ref[:, 4 * C:5 * C]) + 1e-6).repeat(1, 3) * 2.0
    assert ((out[:, std_cols] - ref[:, std_cols]).abs() <= tol).all()
    H, dk = 8, 8
    Q, K, V = (torch.randn(N, H * dk, generator=g) for _ in range(3))
    Ee = torch.randn(E, H * dk, generator=g)
    a = ops.edge_attention(Q.to(DEV), K.to(DEV), V.to(DEV), Ee.to(DEV), plan, H).cpu().double()
    Qd, Kd, Vd, Ed = (t.double().view(-1, H, dk) for t in (Q, K, V, Ee))
    s = torch.exp(((Kd[src] * Qd[dst]) / dk ** 0.5 * Ed).sum(-1, keepdim=True).clamp(-5, 5))
    wV = torch.zeros(N, H, dk, dtype=torch.float64).index_add_(0, dst, Vd[src] * s)
    z = torch.zeros(N, H, 1, dtype=torch.float64).index_add_(0, dst, s)
    torch.testing.assert_close(a, (wV / (z + 1e-6)).reshape(N, -1), rtol=1e-5, atol=1e-5)


def test_gatedgcn_one_launch_bad_type_id_gives_nan_and_raises_at_check():
    from signnet_basisnet_amd import dgl_nets as DN
    from signnet_basisnet_amd import dgl_deepsigns as DS
    from signnet_basisnet_amd import synth
    k = 8
    p = dict(num_atom_type=28, num_bond_type=4, hidden_dim=52, out_dim=52, L=4, readout="mean", batch_norm=True, residual=True,
             edge_feat=True, device=DEV, pe_init="lap_pe", lap_method="sign_inv", lap_lspe=False, use_lapeig_loss=False, lambda_loss=0.0,
             alpha_loss=0.0, pos_enc_dim=k, pe_aggregate="add", in_feat_dropout=0.0, dropout=0.0, sign_inv_net="gin",
             sign_inv_layers=3, phi_out_dim=4, sign_inv_activation="relu")
    net = DN.GatedGCNNet(p).to(DEV).eval()
    data = synth.make_batch(16, seed=45)
    ei = data.edge_index
    g = DS.Graph(ei[0].to(DEV), ei[1].to(DEV), torch.tensor(data.sizes))
    h = data.x.reshape(-1).to(DEV).clone()
    e = data.edge_attr.reshape(-1).to(DEV)
    pe = synth.dgl_pos_enc(data, k).to(DEV)
    pp = net.sign_inv_net(g, pe.unsqueeze(-1)).squeeze(-1)
    y = net(g, h, pp, e)[0]
    assert torch.isfinite(y).all()
    net.check_last()                                   # nothing flagged
    h[3] = 28                                          # one past the atom-type table
    g2 = DS.Graph(ei[0].to(DEV), ei[1].to(DEV), torch.tensor(data.sizes))
    y = net(g2, h, pp, e)[0]
    assert torch.isnan(y).all()
    with pytest.raises(IndexError):
        net.check_last()
    net.fused_stages = False                           # the layer path raises on the spot
    with pytest.raises(IndexError):
        net(g2, h, pp, e)


# ------------------------------------------------------------------------------------------ adjoints of the f3 message-passing ops
def test_pna_and_edge_attention_adjoints_vs_fp64_autograd():
    """sn_pna_aggregate_bwd_f32, sn_edge_attention_bwd_f32, sn_edge_rows_sum_f32, sn_act_bwd_f32 against torch.autograd on the float64
    oracle restatements (the reference obtains these gradients from autograd through DGL's message passing), and reproducibility."""
    from oracle import dgl_nets as ON
    from signnet_basisnet_amd import autograd as AG
    from signnet_basisnet_amd import ops, synth
    data = synth.make_batch(24, seed=17)
    d = synth.batch_to(data, DEV)
    plan = ops.build_plan(d.batch, d.edge_index, d.num_graphs, 0)
    rplan = ops.build_plan(d.batch, d.edge_index.flip(0).contiguous(), d.num_graphs, 0)
    gen = torch.Generator().manual_seed(1)
    src, dst = data.edge_index
    E, N, C = src.numel(), data.num_nodes, 10
    # ---- PNA aggregation
    m, hs = torch.randn(E, C, generator=gen), torch.randn(N, C, generator=gen)
    cot = torch.randn(N, 13 * C, generator=gen)
    hm, hh = m.clone().to(DEV).requires_grad_(True), hs.clone().to(DEV).requires_grad_(True)
    AG.pna_aggregate(hm, hh, plan, 1.3).backward(cot.to(DEV))
    rm, rh = m.double().requires_grad_(True), hs.double().requires_grad_(True)
    torch.cat([rh, ON.pna_aggregate(rm, dst, N, 1.3)], dim=1).backward(cot.double())
    close(hh.grad, rh.grad, "pna d hself", rel=1e-6)
    # the std columns divide by 2*std with std^2 = relu(E[x^2]-E[x]^2)+1e-5 computed in fp32: compare at the level that cancellation allows
    scale = rm.grad.abs().max().item()
    assert (hm.grad.cpu().double() - rm.grad).abs().max().item() <= 2e-3 * scale
    deg = torch.bincount(dst, minlength=N)
    well = (deg[dst] >= 3)                                   # edges into nodes whose variance is not a difference of near-equal numbers
    if well.any():
        assert (hm.grad.cpu().double() - rm.grad)[well].abs().max().item() <= 5e-4 * scale
    # ---- sparse edge attention
    H, dk = 4, 6
    Q, K, V = (torch.randn(N, H * dk, generator=gen) for _ in range(3))
    Ee = torch.randn(E, H * dk, generator=gen)
    cot = torch.randn(N, H * dk, generator=gen)
    hv = [t.clone().to(DEV).requires_grad_(True) for t in (Q, K, V, Ee)]
    out = AG.edge_attention(*hv, plan, rplan, H)
    out.backward(cot.to(DEV))
    rv = [t.double().requires_grad_(True) for t in (Q, K, V, Ee)]
    Qh, Kh, Vh, Eh = (t.view(-1, H, dk) for t in rv)
    score = ((Kh[src] * Qh[dst]) / dk ** 0.5 * Eh).sum(-1)
    s_ = torch.exp(score.clamp(-5, 5))
    z = torch.zeros(N, H, dtype=torch.float64).index_add_(0, dst, s_)
    wV = torch.zeros(N, H, dk, dtype=torch.float64).index_add_(0, dst, s_.unsqueeze(-1) * Vh[src])
    ref = (wV / (z.unsqueeze(-1) + 1e-6)).reshape(N, H * dk)
    close(out, ref.detach(), "edge attention forward", rel=1e-5)
    ref.backward(cot.double())
    for a, b, n in zip(hv, rv, ("dQ", "dK", "dV", "dE")):
        close(a.grad, b.grad, "edge attention " + n, rel=2e-5)
    hv2 = [t.clone().to(DEV).requires_grad_(True) for t in (Q, K, V, Ee)]
    AG.edge_attention(*hv2, plan, rplan, H).backward(cot.to(DEV))
    assert all(torch.equal(a.grad, b.grad) for a, b in zip(hv, hv2))            # no atomics: bitwise reproducible
    # ---- rows gathered onto edges, LeakyReLU + residual, row scaling
    h0 = torch.randn(N, C, generator=gen)
    cot = torch.randn(E, C, generator=gen)
    for idx, pl in ((dst, plan), (src, rplan)):
        a = h0.clone().to(DEV).requires_grad_(True)
        AG.gather_rows(a, idx.to(DEV), pl).backward(cot.to(DEV))
        b = h0.double().requires_grad_(True)
        b[idx].backward(cot.double())
        close(a.grad, b.grad, "gather adjoint", rel=1e-6)
    x, r, sn = torch.randn(N, C, generator=gen), torch.randn(N, C, generator=gen), torch.rand(N, generator=gen) + 0.5
    cot = torch.randn(N, C, generator=gen)
    a, ar = x.clone().to(DEV).requires_grad_(True), r.clone().to(DEV).requires_grad_(True)
    AG.act_residual(a, residual=ar, act="leaky", slope=0.01).backward(cot.to(DEV))
    b, br = x.double().requires_grad_(True), r.double().requires_grad_(True)
    (torch.nn.functional.leaky_relu(b, 0.01) + br).backward(cot.double())
    close(a.grad, b.grad, "leaky adjoint", rel=1e-6)
    close(ar.grad, br.grad, "residual adjoint", rel=1e-6)
    a = x.clone().to(DEV).requires_grad_(True)
    AG.act_residual(a, rowscale=sn.to(DEV)).backward(cot.to(DEV))
    close(a.grad, cot.double() * sn.double().unsqueeze(1), "row-scale adjoint", rel=1e-6)


def _oracle_grads(fn, sd):
    sd64 = {k: (v.double().requires_grad_(True) if v.is_floating_point() and "running" not in k else v) for k, v in sd.items()}
    y = fn(sd64)
    cot = torch.randn(y.shape, generator=torch.Generator().manual_seed(5), dtype=torch.float64)
    (y * cot).sum().backward()
    return sd64, y.detach(), cot


def _check_param_grads(net, sd64, what, tol=2e-3):
    gmax = max(v.grad.abs().max().item() for v in sd64.values() if torch.is_tensor(v) and v.requires_grad and v.grad is not None)
    seen = 0
    for k, p in net.named_parameters():
        if k.startswith("sign_inv_net.") or k not in sd64:
            continue
        g64 = sd64[k].grad if sd64[k].grad is not None else torch.zeros_like(sd64[k])
        ours = p.grad if p.grad is not None else torch.zeros_like(p)
        scale = max(g64.abs().max().item(), 1e-2 * gmax)
        err = (ours.detach().cpu().double() - g64).abs().max().item()
        assert err <= tol * scale + 1e-7, (what, k, err, scale)
        seen += 1
    assert seen > 10


def test_dgl_pna_net_parameter_gradients_match_oracle_autograd():
    """Train mode with gradients enabled: every PNANet parameter gradient against torch.autograd over the float64 oracle on the
    reference's fixture (same state_dict, same batch, the fixture's own sign-invariant encoding as input)."""
    from oracle import dgl_nets as ON
    from signnet_basisnet_amd import dgl_deepsigns as DS
    from signnet_basisnet_amd import dgl_nets
    fx = G.load("dgl_pna_k6")
    hidden, L, k, towers, edge_dim = (int(v) for v in fx.meta["params"])
    avg = fx.meta["avg_d"]
    net = dgl_nets.PNANet(_base_params(fx, hidden, L, k, graph_norm=True, aggregators="mean max min std",
                                       scalers="identity amplification attenuation",
                                       avg_d=dict(lin=float(avg[0]), exp=float(avg[1]), log=float(avg[2])), towers=towers,
                                       divide_input_first=True, divide_input_last=True, edge_dim=edge_dim, pretrans_layers=1,
                                       posttrans_layers=1, gru=False))
    net.load_state_dict(fx.sd)
    net = net.to(DEV).train()
    ei = fx.inp["edge_index"]
    g = DS.Graph(ei[0].to(DEV), ei[1].to(DEV), fx.inp["sizes"])
    p_in = fx.out["train/p"]
    sd64, y64, cot = _oracle_grads(lambda sd: ON.pna_net(sd, ei[0], ei[1], fx.inp["sizes"], fx.inp["x"].squeeze(-1), p_in.double(), fx.inp["edge_attr"],
                                                          fx.inp["snorm_n"].double(), L, towers, float(avg[2]), "sum", training=True), fx.sd)
    y, _ = net(g, fx.inp["x"].squeeze(-1).to(DEV), p_in.to(DEV), fx.inp["edge_attr"].to(DEV), fx.inp["snorm_n"].to(DEV))
    assert y.requires_grad
    torch.testing.assert_close(y.detach().cpu().double(), y64, rtol=2e-3, atol=2e-4)
    (y * cot.float().to(DEV)).sum().backward()
    _check_param_grads(net, sd64, "PNANet", tol=5e-3)


def test_dgl_gat_net_parameter_gradients_match_oracle_autograd():
    """Train mode with gradients enabled: every GATNet parameter gradient (fc, attn_l, attn_r, bias of each GATConv, the embeddings,
    the readout) against torch.autograd over the float64 oracle on the reference's fixture."""
    from oracle import dgl_nets as ON
    from signnet_basisnet_amd import dgl_deepsigns as DS
    from signnet_basisnet_amd import dgl_nets
    fx = G.load("dgl_gat_k6")
    hidden, L, k, heads = (int(v) for v in fx.meta["params"])
    net = dgl_nets.GATNet(_base_params(fx, hidden, L, k, n_heads=heads, readout="mean", pe_aggregate="concat"))
    net.load_state_dict(fx.sd)
    net = net.to(DEV).train()
    ei = fx.inp["edge_index"]
    g = DS.Graph(ei[0].to(DEV), ei[1].to(DEV), fx.inp["sizes"])
    p_in = fx.out["train/p"]
    sd64, y64, cot = _oracle_grads(lambda sd: ON.gat_net(sd, ei[0], ei[1], fx.inp["sizes"], fx.inp["x"].squeeze(-1), p_in.double(), L, heads,
                                                          "mean"), fx.sd)
    y, _ = net(g, fx.inp["x"].squeeze(-1).to(DEV), p_in.to(DEV), fx.inp["edge_attr"].to(DEV))
    assert y.requires_grad
    close(y.detach(), y64, "GATNet train-mode scores")
    (y * cot.float().to(DEV)).sum().backward()
    _check_param_grads(net, sd64, "GATNet", tol=1e-4)


@pytest.mark.parametrize("heads,C,relu", [(4, 12, True), (4, 59, True), (2, 64, False), (3, 1, True)])
def test_gat_aggregate_adjoint_vs_fp64_autograd(heads, C, relu):
    """sn_gat_aggregate_bwd_f32 (+ the attn_l / attn_r / bias reductions) against torch.autograd on the float64 restatement."""
    from oracle import dgl_nets as ON
    from signnet_basisnet_amd import autograd as AG, ops, synth
    data = synth.make_batch(30, seed=19)
    d = synth.batch_to(data, DEV)
    plan = ops.build_plan(d.batch, d.edge_index, d.num_graphs, 0)
    rplan = ops.build_plan(d.batch, d.edge_index.flip(0).contiguous(), d.num_graphs, 0)
    g = torch.Generator().manual_seed(heads * 100 + C)
    src, dst = data.edge_index
    N = data.num_nodes
    f = torch.randn(N, heads * C, generator=g)
    al, ar = torch.randn(1, heads, C, generator=g), torch.randn(1, heads, C, generator=g)
    b = torch.randn(heads * C, generator=g)
    cot = torch.randn(N, heads * C, generator=g)
    ours = [t.clone().to(DEV).requires_grad_(True) for t in (f, al, ar, b)]
    AG.gat_aggregate(ours[0], ours[1], ours[2], ours[3], plan, rplan, heads, 0.2, relu).backward(cot.to(DEV))
    ref = [t.double().requires_grad_(True) for t in (f, al, ar, b)]
    sd = {"l.fc.weight": torch.eye(heads * C, dtype=torch.float64), "l.attn_l": ref[1], "l.attn_r": ref[2], "l.bias": ref[3]}
    if relu:
        out = ON.gat_conv(sd, "l", src, dst, ref[0], heads).flatten(1)
    else:
        fd = ref[0].view(N, heads, C)
        e = torch.nn.functional.leaky_relu((fd * ref[1]).sum(-1)[src] + (fd * ref[2]).sum(-1)[dst], 0.2)
        lse = torch.stack([torch.logsumexp(e[dst == n], 0) for n in range(N)])
        out = (torch.zeros(N, heads, C, dtype=torch.float64).index_add(0, dst, torch.exp(e - lse[dst]).unsqueeze(-1) * fd[src])
               + ref[3].view(1, heads, C)).flatten(1)
    out.backward(cot.double())
    for o, r, what in zip(ours, ref, ("d feat", "d attn_l", "d attn_r", "d bias")):
        close(o.grad, r.grad, f"gat {what}", rel=1e-5)


@pytest.mark.parametrize("name", ["dgl_transformer_concat_k6", "dgl_transformer_add_k8"])
def test_dgl_transformer_net_parameter_gradients_match_oracle_autograd(name):
    from oracle import dgl_nets as ON
    from signnet_basisnet_amd import dgl_deepsigns as DS
    from signnet_basisnet_amd import dgl_nets
    fx = G.load(name)
    hidden, L, k, heads = (int(v) for v in fx.meta["params"])
    agg = str(fx.meta["pe_aggregate"])
    net = dgl_nets.TransformerNet(_base_params(fx, hidden, L, k, n_heads=heads, full_graph=False, layer_norm=True, pe_aggregate=agg))
    net.load_state_dict(fx.sd)
    net = net.to(DEV).train()
    ei = fx.inp["edge_index"]
    g = DS.Graph(ei[0].to(DEV), ei[1].to(DEV), fx.inp["sizes"])
    p_in = fx.out["train/p"]
    sd64, y64, cot = _oracle_grads(lambda sd: ON.transformer_net(sd, ei[0], ei[1], fx.inp["sizes"], fx.inp["x"].squeeze(-1), p_in.double(),
                                                                  fx.inp["edge_attr"], L, heads, agg, "sum", training=True), fx.sd)
    y, _ = net(g, fx.inp["x"].squeeze(-1).to(DEV), p_in.to(DEV), fx.inp["edge_attr"].to(DEV))
    assert y.requires_grad
    torch.testing.assert_close(y.detach().cpu().double(), y64, rtol=2e-3, atol=2e-4)
    (y * cot.float().to(DEV)).sum().backward()
    _check_param_grads(net, sd64, "TransformerNet")


# ------------------------------------------------------------------------------------------ edge cases of the DGL stage kernels
def _gated_net(hidden=52, L=3, k=6):
    from signnet_basisnet_amd import dgl_nets as DN
    torch.manual_seed(7)
    p = dict(num_atom_type=28, num_bond_type=4, hidden_dim=hidden, out_dim=hidden, L=L, readout="mean", batch_norm=True, residual=True,
             edge_feat=True, device=DEV, pe_init="lap_pe", lap_method="sign_inv", lap_lspe=False, use_lapeig_loss=False, lambda_loss=0.0,
             alpha_loss=0.0, pos_enc_dim=k, pe_aggregate="add", in_feat_dropout=0.0, dropout=0.0, sign_inv_net="masked_gin",
             sign_inv_layers=3, phi_out_dim=20, sign_inv_activation="relu")
    net = DN.GatedGCNNet(p)
    PU.bn_randomize(net, 9)
    return net.to(DEV).eval()


def _run_both(net, data, k):
    from signnet_basisnet_amd import dgl_deepsigns as DS
    from signnet_basisnet_amd import synth
    ei = data.edge_index
    h, e = data.x.reshape(-1).to(DEV), data.edge_attr.reshape(-1).to(DEV)
    pe = synth.dgl_pos_enc(data, k).to(DEV)
    outs = []
    for fused in (True, False):
        net.fused_stages = net.sign_inv_net.fused_stages = fused
        g = DS.Graph(ei[0].to(DEV), ei[1].to(DEV), torch.tensor(data.sizes))
        pp = net.sign_inv_net(g, pe.unsqueeze(-1)).squeeze(-1)
        outs.append((pp, net(g, h, pp, e)[0]))
    net.fused_stages = net.sign_inv_net.fused_stages = True
    return outs


def test_dgl_stage_kernels_ragged_batch_one_node_graphs_and_the_64_node_limit():
    """Sizes 1, 2, 3 (fewer nodes than eigenvector slots: zero-padded columns, a graph without edges), the 64-node maximum of the
    one-workgroup-per-graph kernels, and ordinary sizes in one batch: stage kernels == layer path for the masked sign-invariant net
    and the GatedGCN stack."""
    from signnet_basisnet_amd import synth
    k = 6
    net = _gated_net(k=k)
    data = synth.make_batch(9, seed=51, sizes=[1, 2, 3, 64, 17, 1, 40, 5, 64])
    (p_f, y_f), (p_l, y_l) = _run_both(net, data, k)
    assert torch.isfinite(y_f).all() and y_f.shape == (9, 1)
    close(p_f, p_l, "MaskedGINDeepSigns stage kernels vs layer path (ragged)")
    close(y_f, y_l, "GatedGCN one launch vs layer path (ragged)")
    net.check_last()


def test_dgl_stage_kernels_oversize_graph_falls_back_to_the_layer_path():
    """A 70-node graph: the modules see it in batch_num_nodes and use the layer-at-a-time kernels for the whole batch."""
    from signnet_basisnet_amd import ops, synth
    from signnet_basisnet_amd import dgl_deepsigns as DS
    k = 6
    net = _gated_net(k=k)
    data = synth.make_batch(3, seed=52, sizes=[12, 70, 9])
    ei = data.edge_index
    g = DS.Graph(ei[0].to(DEV), ei[1].to(DEV), torch.tensor(data.sizes))
    pe = synth.dgl_pos_enc(data, k).to(DEV)
    rec = ops.KernelTimer()
    with rec:
        pp = net.sign_inv_net(g, pe.unsqueeze(-1)).squeeze(-1)
        y = net(g, data.x.reshape(-1).to(DEV), pp, data.edge_attr.reshape(-1).to(DEV))[0]
    names = {n for n, _, _ in rec.spans}
    assert "sn_gatedgcn_fused_f32" not in names and "sn_deepsigns_phi_f32" not in names and "sn_gated_aggregate_f32" in names
    assert torch.isfinite(y).all()


def test_gatedgcn_graph_with_too_many_edges_for_the_one_launch_kernel():
    """A dense 40-node graph (more in-edges than the kernel's LDS image holds).  The module reads the batch's largest per-graph edge
    count once per graph object and evaluates such a batch layer by layer, as the reference evaluates any graph.  The device-side
    guard stays: a batch that reaches the kernel anyway (here: the cached count overwritten) gets a NaN score for that graph only,
    the rest unaffected, and check_last() raises."""
    from signnet_basisnet_amd import dgl_deepsigns as DS
    from signnet_basisnet_amd import synth
    k = 6
    net = _gated_net(k=k)
    data = synth.make_batch(3, seed=53, sizes=[10, 40, 8])
    n0, n1 = 10, 40
    ii, jj = torch.meshgrid(torch.arange(n1), torch.arange(n1), indexing="ij")
    keep = ii != jj
    dense = torch.stack([ii[keep], jj[keep]]) + n0                       # 1560 directed edges inside graph 1
    ei = torch.cat([data.edge_index, dense], dim=1)
    e = torch.cat([data.edge_attr.reshape(-1), torch.ones(dense.shape[1], dtype=torch.long)])
    bnn = torch.tensor(data.sizes)
    bne = torch.bincount(torch.bucketize(ei[1], torch.cumsum(bnn, 0), right=True), minlength=3)       # what DGL's batch carries
    g = DS.Graph(ei[0].to(DEV), ei[1].to(DEV), bnn, bne)
    pe = synth.dgl_pos_enc(data, k).to(DEV)
    net.sign_inv_net.fused_stages = False
    pp = net.sign_inv_net(g, pe.unsqueeze(-1)).squeeze(-1)
    y_auto = net(g, data.x.reshape(-1).to(DEV), pp, e.to(DEV))[0]
    net.check_last()
    assert DS._max_in_edges(g) >= dense.shape[1] and net._fused_gated(g) is None      # (the molecule's own edges + the dense block)
    assert torch.isfinite(y_auto).all()
    g = DS.Graph(ei[0].to(DEV), ei[1].to(DEV), bnn)                      # no edge counts: the batch reaches the kernel
    assert DS._max_in_edges(g) is None
    y = net(g, data.x.reshape(-1).to(DEV), pp, e.to(DEV))[0]
    assert torch.isnan(y[1]).all() and torch.isfinite(y[0]).all() and torch.isfinite(y[2]).all()
    with pytest.raises(RuntimeError):
        net.check_last()
    net.fused_stages = False
    y_l = net(g, data.x.reshape(-1).to(DEV), pp, e.to(DEV))[0]
    assert torch.isfinite(y_l).all()
    assert torch.equal(y_auto, y_l)
    close(y[[0, 2]], y_l[[0, 2]], "graphs beside the flagged one")


def test_sign_inv_net_overlap_mode_hands_its_output_over_with_an_event():
    """GINDeepSigns.overlap: plan + stage launches on the module's side stream, the result and the graph's plan picked up by the base
    network behind an event kept on the graph object.  A loop over fresh graph objects gives the sequential outputs bit for bit."""
    from signnet_basisnet_amd import dgl_deepsigns as DS
    from signnet_basisnet_amd import dgl_nets, synth
    params = dict(num_atom_type=28, num_bond_type=4, in_feat_dropout=0.0, dropout=0.0, batch_norm=True, residual=True, edge_feat=True,
                  pe_init="lap_pe", lap_method="sign_inv", lap_lspe=False, use_lapeig_loss=False, lambda_loss=1, alpha_loss=1e-4,
                  sign_inv_net="gin", sign_inv_layers=8, sign_inv_activation="relu", phi_out_dim=4, device="cuda:0",
                  hidden_dim=68, out_dim=68, L=4, readout="mean", pos_enc_dim=8, pe_aggregate="concat")
    torch.manual_seed(0)
    nets = {"gated": dgl_nets.GatedGCNNet(params).cuda().eval(), "gin": dgl_nets.GINNet(dict(params, hidden_dim=64, out_dim=64)).cuda().eval()}
    hosts = [synth.make_batch(64 + 8 * i, seed=90 + i) for i in range(4)]
    ins = []
    for h in hosts:
        pe = synth.dgl_pos_enc(h, 8).unsqueeze(-1).cuda()
        ins.append((h.edge_index[0].cuda(), h.edge_index[1].cuda(), torch.tensor(h.sizes), h.x.squeeze(-1).cuda(), h.edge_attr.cuda(), pe))
    for name, net in nets.items():
        def run(i):
            s, d, bnn, hx, ex, pe = ins[i]
            g = DS.Graph(s, d, bnn)
            p = net.sign_inv_net(g, pe).squeeze(-1)
            return net(g, hx, p, ex, None)[0]
        with torch.no_grad():
            ref = [run(i).clone() for i in range(4)]
            torch.cuda.synchronize()
            net.sign_inv_net.overlap = True
            outs = [(i % 4, run(i % 4) * 1.0) for i in range(12)]
            if hasattr(net, "check_last"):
                net.check_last()
            torch.cuda.synchronize()
            net.sign_inv_net.overlap = False
        assert getattr(net.sign_inv_net, "_side_stream", None) is not None
        for i, y in outs:
            assert torch.equal(y, ref[i]), name


def test_round3_entry_points_against_float64():
    """sn_grouped_linear_f32, sn_masked_linear_blockbias_f32, sn_pna_aggregate_gather_f32 (both output layouts) and sn_ign_mlp_f32 on their
    own, against float64 restatements (the module tests above only see them through whole networks)."""
    from signnet_basisnet_amd import basisnet as BNm
    from signnet_basisnet_amd import ops, synth
    g = torch.Generator().manual_seed(11)
    # grouped (block-diagonal) Linear with row scale and affine epilogue
    R, G, din, dout = 777, 5, 208, 14
    x = torch.randn(R, G * din, generator=g)
    W, b = torch.randn(G, dout, din, generator=g) / din ** 0.5, torch.randn(G * dout, generator=g)
    rs, sc, sh = torch.rand(R, generator=g) + 0.5, torch.randn(G * dout, generator=g), torch.randn(G * dout, generator=g)
    y = ops.grouped_linear(x.cuda(), W.cuda().contiguous(), b.cuda(), G, rowscale=rs.cuda(), scale=sc.cuda(), shift=sh.cuda()).cpu()
    ref = torch.cat([x.double()[:, i * din:(i + 1) * din] @ W[i].double().t() for i in range(G)], 1)
    ref = ((ref + b.double()) * rs.double()[:, None]) * sc.double() + sh.double()
    close(y, ref, "grouped_linear", 2e-6)
    # Linear with a bias per block of rows
    nb, rpb, H = 7, 96, 32
    h = torch.randn(nb * rpb, H, generator=g)
    Wl, bl, bb = torch.randn(H, H, generator=g) / H ** 0.5, torch.randn(H, generator=g), torch.randn(nb, H, generator=g)
    pl = ops.PackedLinear(ops.pack_weight(Wl.cuda().contiguous()), H, H, bl.cuda())
    y = ops.linear_block_bias(h.cuda(), pl, bb.cuda(), rpb, relu_pre=True, scale=sc[:H].cuda().contiguous(), shift=sh[:H].cuda().contiguous()).cpu()
    ref = torch.relu(h.double() @ Wl.double().t() + bl.double() + bb.double().repeat_interleave(rpb, 0)) * sc[:H].double() + sh[:H].double()
    close(y, ref, "linear_block_bias", 2e-6)
    # PNA aggregation with the message formed in the kernel == the message-array form on the summed messages
    data = synth.make_batch(24, seed=8)
    plan = ops.build_plan(data.batch.cuda(), data.edge_index.cuda(), 24, 0)
    N, E, Cc = data.batch.numel(), data.edge_index.shape[1], 20
    psd, qe, hs = torch.randn(N, 2 * Cc, generator=g), torch.randn(E, Cc, generator=g), torch.randn(N, Cc, generator=g)
    src, dst = data.edge_index
    msg = psd[src, :Cc] + psd[dst, Cc:] + qe
    want = ops.pna_aggregate(msg.cuda(), hs.cuda(), plan, 1.3).cpu()
    got = ops.pna_aggregate_gather(psd.cuda(), qe.cuda(), hs.cuda(), plan, 1.3).cpu()
    close(got, want, "pna_aggregate_gather", 2e-6)
    it = 5
    tw = ops.pna_aggregate_gather(psd.cuda(), qe.cuda(), hs.cuda(), plan, 1.3, tower_width=it).cpu()
    for t in range(Cc // it):
        for j in range(13):
            assert torch.equal(tw[:, t * 13 * it + j * it:t * 13 * it + (j + 1) * it], got[:, j * Cc + t * it:j * Cc + (t + 1) * it])
    # the one-launch IGN head == the layer-at-a-time head (same module, fused_head off), several matrix sizes / widths
    for (n, H, mult, bsz) in ((1024, 32, 2, 5), (100, 16, 1, 9), (333, 32, 32, 3)):
        torch.manual_seed(n)
        enc = BNm.IGN2to1(1, H, mult).cuda().eval()
        with torch.no_grad():
            for bn in enc.bns:
                bn.running_mean.normal_(0, 0.1); bn.running_var.uniform_(0.5, 1.5)
            o = torch.randn(bsz, n, 5, device="cuda")
            y1 = enc.forward_contractions(o)
            enc.fused_head = False
            y0 = enc.forward_contractions(o)
        assert y1.shape == y0.shape == (bsz, mult, n)
        close(y1.cpu(), y0.cpu(), f"IGN head n={n} H={H}", 5e-6)


def test_deepsets_tail_kernel_equals_the_layer_path():
    """sn_deepsets_tail_f32 (one set, everything behind the first layer's Linear in one launch) == the layer-at-a-time eval path and the
    float64 restatement of EqDeepSetsEncoder.forward, with and without the batch-statistics BatchNorm."""
    from signnet_basisnet_amd import basisnet as BNm
    for (n, fin, hid, out, L, use_bn) in ((1024, 2048, 10, 32, 3, True), (300, 37, 32, 7, 4, False), (77, 12, 16, 16, 2, True)):
        torch.manual_seed(n)
        enc = BNm.EqDeepSetsEncoder(fin, hidden_channels=hid, num_layers=L, out_channels=out, use_bn=use_bn).cuda().eval()
        x = torch.randn(n, fin, device="cuda")
        with torch.no_grad():
            y1 = enc(x)
            enc.fused_tail = False
            y0 = enc(x)
            # float64 restatement (models.py:58-113)
            h = x.double().cpu()
            for i in range(L):
                l1, l2 = enc.lins1[i], enc.lins2[i]
                h = h @ l1.weight.double().cpu().t() + l1.bias.double().cpu() + (h.mean(0, keepdim=True) @ l2.weight.double().cpu().t() + l2.bias.double().cpu())
                if i < L - 1:
                    h = torch.relu(h)
                    if use_bn:
                        bn = enc.bns[i]
                        h = (h - h.mean(0)) / torch.sqrt(h.var(0, unbiased=False) + bn.eps) * bn.weight.double().cpu() + bn.bias.double().cpu()
        assert y1.shape == y0.shape == (n, out)
        close(y1.cpu(), h, f"deepsets tail vs float64 n={n}", 2e-5)
        close(y1.cpu(), y0.cpu(), f"deepsets tail vs layer path n={n}", 2e-5)
