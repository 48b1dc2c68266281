#!/usr/bin/env python
"""Generate the golden fixtures in tests/golden/*.npz.  BUILD-CONTAINER ONLY.

Imports the reference's own module code from /root/reference (unmodified) with the
third-party graph ops it needs (absent from the image) supplied by tests/golden/ref_shim/,
runs it on tiny seeded batches and stores inputs, the full state_dict and the outputs.
The fixtures are plain arrays; nothing of the reference's source travels with them.

    python tests/golden/make_golden.py          # rewrites every fixture

The GPU box never runs this (it has no /root/reference).
"""
from __future__ import annotations

import importlib
import os
import re
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
SHIM = os.path.join(HERE, "ref_shim")
sys.path.insert(0, ROOT)

from signnet_basisnet_amd import synth  # noqa: E402


def _fresh_import(tree, names):
    """Import reference modules from one tree with the shim ahead of everything."""
    for m in list(sys.modules):
        if m.split(".")[0] in ("sign_net", "core", "layers", "nets", "ign", "signbasisnet", "models"):
            del sys.modules[m]
    sys.path[:0] = [SHIM, os.path.join(REF, tree)]
    try:
        return [importlib.import_module(n) for n in names]
    finally:
        del sys.path[:2]


def randomise(model, seed):
    """Make BN running stats / affines and GIN eps non-trivial so eval-mode BN is not the identity."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, mod in model.named_modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                if mod.running_mean is not None:
                    mod.running_mean.copy_(torch.randn(mod.running_mean.shape, generator=g) * 0.1)
                    mod.running_var.copy_(torch.rand(mod.running_var.shape, generator=g) + 0.5)
                mod.weight.copy_(1 + 0.2 * torch.randn(mod.weight.shape, generator=g))
                mod.bias.copy_(0.1 * torch.randn(mod.bias.shape, generator=g))
            if isinstance(mod, torch.nn.LayerNorm):
                mod.weight.copy_(1 + 0.2 * torch.randn(mod.weight.shape, generator=g))
                mod.bias.copy_(0.1 * torch.randn(mod.bias.shape, generator=g))
        for name, p in model.named_parameters():
            if name.endswith(".eps"):
                p.copy_(0.3 * torch.randn(p.shape, generator=g))


_SKIP = re.compile(r"(embeddings\.[1-9]\.|phi\.edge_encoders\.|\.layer\.nn\.)")


def sd_arrays(model):
    """State dict as arrays.  To keep fixtures small the tensors that never influence the forward
    (embedding tables of feature columns 1..9 — ZINC has one column; GNN3d.edge_encoders — unused,
    core/sign_net.py:22,40; the second registration of each GINE MLP under `.layer.nn.`,
    pyg_gnn_wrapper.py:22-23) are listed by key+shape only (`meta/sd_keys`, `meta/sd_shapes`)."""
    sd = model.state_dict()
    out = {"sd/" + k: v.detach().cpu().clone().numpy() for k, v in sd.items() if not _SKIP.search(k)}
    out["meta/sd_keys"] = np.array(list(sd.keys()))
    out["meta/sd_shapes"] = np.array([",".join(map(str, v.shape)) for v in sd.values()])
    return out


def data_arrays(data):
    return {"in/" + k: v.numpy() for k, v in vars(data).items() if torch.is_tensor(v)} | {
        "in/sizes": np.array(data.sizes, dtype=np.int64)}


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB, {len(arrays)} arrays")


# ------------------------------------------------------------------ PyG trees
def pyg_case(name, variant, ctor_args, sizes, features, seed):
    tree, modname = ("Alchemy", "sign_net.sign_net") if variant == "alchemy" else ("GINESignNetPyG", "core.sign_net")
    (mod,) = _fresh_import(tree, [modname])
    torch.manual_seed(seed)
    model = mod.SignNetGNN(*ctor_args)
    randomise(model, seed + 1)
    data = synth.make_batch(len(sizes), seed=seed, sizes=sizes, features=features)
    cap = {}
    hooks = [
        model.sign_net.phi.register_forward_hook(lambda m, i, o: cap.setdefault("phi_calls", []).append(o.detach().clone())),
        model.sign_net.register_forward_hook(lambda m, i, o: cap.__setitem__("pos", o.detach().clone())),
    ]
    for l, conv in enumerate(model.gnn.convs):
        hooks.append(conv.register_forward_hook(
            lambda m, i, o, l=l: cap.__setitem__(f"gine_conv{l}", o.detach().clone())))
    arrays = {**sd_arrays(model), **data_arrays(data),
              "meta/ctor": np.array([-1 if a is None else a for a in ctor_args], dtype=np.int64),
              "meta/variant": np.array(variant)}
    model.eval()
    with torch.no_grad():
        y = model(data)
    arrays.update({"out/eval/y": y.numpy(), "out/eval/pos": cap["pos"].numpy(),
                   "out/eval/phi_plus": cap["phi_calls"][0].numpy(),
                   "out/eval/phi_minus": cap["phi_calls"][1].numpy()})
    for l in range(len(model.gnn.convs)):
        arrays[f"out/eval/gine_conv{l}"] = cap[f"gine_conv{l}"].numpy()
    # train-mode forward (batch-stat BN).  The attention dropout p=0.1 that survives in train mode
    # (transformer_module.py:46) is switched off so the fixture is deterministic.
    cap.clear()
    model.train()
    for layer in model.sign_net.rho.transformer_layers:
        layer.slf_attn.attention.dropout.p = 0.0
    with torch.no_grad():
        yt = model(data)
    arrays.update({"out/train/y": yt.numpy(), "out/train/pos": cap["pos"].numpy()})
    # train-mode forward WITH the attention dropout the reference leaves active (ScaledDotProductAttention(attn_dropout=0.1),
    # transformer_module.py:46,55): torch's own Bernoulli draws, seeded; the keep-masks it drew are read off the Dropout modules'
    # inputs / outputs (out != 0 wherever the softmax probability is not 0) and stored with the outputs, so that the HIP path can be
    # run on the very same masks.  (BatchNorm in train mode normalises with batch statistics: the earlier forwards do not matter.)
    cap.clear()
    rng = torch.get_rng_state()
    torch.manual_seed(seed + 1000)
    masks = {}
    mhooks = []
    for li, layer in enumerate(model.sign_net.rho.transformer_layers):
        layer.slf_attn.attention.dropout.p = 0.1
        mhooks.append(layer.slf_attn.attention.dropout.register_forward_hook(
            lambda m, i, o, li=li: masks.__setitem__(li, ((o != 0) | (i[0] == 0)).detach().clone())))
    with torch.no_grad():
        yd = model(data)
    for h in mhooks:
        h.remove()
    torch.set_rng_state(rng)
    arrays.update({"out/train_do/y": yd.numpy(), "out/train_do/pos": cap["pos"].numpy()})
    for li, m in masks.items():
        arrays[f"out/train_do/keep{li}"] = np.packbits(m.numpy().astype(np.uint8))           # [N, heads, K, K] bits, row-major
        arrays[f"out/train_do/keep{li}_shape"] = np.array(m.shape, dtype=np.int64)
    for h in hooks:
        h.remove()
    save(name, **arrays)


# ------------------------------------------------------------------ DGL tree
def dgl_case(name, kind, hidden, c, layers, k, sizes, seed):
    mods = _fresh_import("GraphPrediction", ["nets.ZINC_graph_regression.sign_inv_net"])
    import dgl  # the shim
    params = dict(sign_inv_net=kind, hidden_dim=hidden, phi_out_dim=c, sign_inv_layers=layers,
                  pos_enc_dim=k, dropout=0.0, sign_inv_activation="relu", device="cpu")
    torch.manual_seed(seed)
    net = mods[0].get_sign_inv_net(params)
    randomise(net, seed + 1)
    data = synth.make_batch(len(sizes), seed=seed, sizes=sizes)
    pe = synth.dgl_pos_enc(data, k)
    g = dgl.Graph(data.edge_index[0], data.edge_index[1], torch.tensor(data.sizes))
    x = pe.unsqueeze(-1)
    arrays = {**sd_arrays(net), **data_arrays(data), "in/pos_enc": pe.numpy(),
              "meta/params": np.array([hidden, c, layers, k], dtype=np.int64), "meta/kind": np.array(kind)}
    net.eval()
    with torch.no_grad():
        arrays["out/eval/y"] = net(g, x.clone()).numpy()
    net.train()
    with torch.no_grad():
        arrays["out/train/y"] = net(g, x.clone()).numpy()
    save(name, **arrays)


def dgl_ginnet_case(name, hidden, L, k, sizes, seed):
    """The DGL tree's GIN base network with a sign-invariant PE (nets/ZINC_graph_regression/gin_net.py, config
    GIN_ZINC_LapPE_signinv_GIN.json scaled down) driven as train_ZINC_graph_regression.py:20-25,77-80 drives it."""
    mods = _fresh_import("GraphPrediction", ["nets.ZINC_graph_regression.gin_net"])
    import dgl  # the shim
    params = dict(num_atom_type=28, num_bond_type=4, hidden_dim=hidden, out_dim=hidden, in_feat_dropout=0.0, dropout=0.0, L=L,
                  readout="mean", batch_norm=True, residual=True, edge_feat=True, device="cpu", pe_init="lap_pe",
                  lap_method="sign_inv", lap_lspe=False, use_lapeig_loss=False, lambda_loss=1, alpha_loss=1e-4, pos_enc_dim=k,
                  sign_inv_net="gin", sign_inv_layers=3, sign_inv_activation="relu", pe_aggregate="add", phi_out_dim=4)
    torch.manual_seed(seed)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        net = mods[0].GINNet(params)
    randomise(net, seed + 1)
    data = synth.make_batch(len(sizes), seed=seed, sizes=sizes)
    pe = synth.dgl_pos_enc(data, k)
    g = dgl.Graph(data.edge_index[0], data.edge_index[1], torch.tensor(data.sizes))
    arrays = {**sd_arrays(net), **data_arrays(data), "in/pos_enc": pe.numpy(),
              "meta/params": np.array([hidden, L, k], dtype=np.int64)}
    for mode in ("eval", "train"):
        net.train(mode == "train")
        with torch.no_grad():
            p = net.sign_inv_net(g, pe.unsqueeze(-1).clone()).squeeze(-1)           # handle_lap, sign_inv branch
            y, _ = net(g, data.x.squeeze(-1), p, data.edge_attr, None)
        arrays[f"out/{mode}/p"] = p.numpy()
        arrays[f"out/{mode}/y"] = y.numpy()
    save(name, **arrays)


def dgl_gatedgcn_case(name, hidden, L, k, sizes, seed, pe_aggregate):
    """The DGL tree's GatedGCN base network with a sign-invariant PE (nets/ZINC_graph_regression/gatedgcn_net.py +
    layers/gatedgcn_layer.py, config GatedGCN_ZINC_LapPE_signinv_GIN.json scaled down)."""
    mods = _fresh_import("GraphPrediction", ["nets.ZINC_graph_regression.gatedgcn_net"])
    import dgl  # the shim
    params = dict(num_atom_type=28, num_bond_type=4, hidden_dim=hidden, out_dim=hidden, in_feat_dropout=0.0, dropout=0.0, L=L,
                  readout="mean", batch_norm=True, residual=True, edge_feat=True, device="cpu", pe_init="lap_pe",
                  lap_method="sign_inv", lap_lspe=False, use_lapeig_loss=False, lambda_loss=1, alpha_loss=1e-4, pos_enc_dim=k,
                  sign_inv_net="masked_gin", sign_inv_layers=3, sign_inv_activation="relu", pe_aggregate=pe_aggregate, phi_out_dim=4)
    torch.manual_seed(seed)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        net = mods[0].GatedGCNNet(params)
    randomise(net, seed + 1)
    data = synth.make_batch(len(sizes), seed=seed, sizes=sizes)
    pe = synth.dgl_pos_enc(data, k)
    g = dgl.Graph(data.edge_index[0], data.edge_index[1], torch.tensor(data.sizes))
    arrays = {**sd_arrays(net), **data_arrays(data), "in/pos_enc": pe.numpy(),
              "meta/params": np.array([hidden, L, k], dtype=np.int64), "meta/pe_aggregate": np.array(pe_aggregate)}
    for mode in ("eval", "train"):
        net.train(mode == "train")
        with torch.no_grad():
            p = net.sign_inv_net(g, pe.unsqueeze(-1).clone()).squeeze(-1)
            y, _ = net(g, data.x.squeeze(-1), p, data.edge_attr, None)
        arrays[f"out/{mode}/p"] = p.numpy()
        arrays[f"out/{mode}/y"] = y.numpy()
        arrays[f"out/{mode}/h_last"] = g.ndata["h"].numpy()
    save(name, **arrays)


def _run_dgl_net(name, net, data, k, extra, snorm=False):
    """Shared tail of the DGL base-net cases: eval- and train-mode forward through the reference module (driven as
    train_ZINC_graph_regression.py:20-25,77-80 drives it) and the fixture arrays."""
    import dgl  # the shim
    pe = synth.dgl_pos_enc(data, k)
    g = dgl.Graph(data.edge_index[0], data.edge_index[1], torch.tensor(data.sizes))
    sn = torch.cat([torch.full((n, 1), 1.0 / n) for n in data.sizes]).sqrt() if snorm else None     # data/molecules.py:307-308
    arrays = {**sd_arrays(net), **data_arrays(data), "in/pos_enc": pe.numpy(), **extra}
    if snorm:
        arrays["in/snorm_n"] = sn.numpy()
    for mode in ("eval", "train"):
        net.train(mode == "train")
        with torch.no_grad():
            p = net.sign_inv_net(g, pe.unsqueeze(-1).clone()).squeeze(-1)
            y, _ = net(g, data.x.squeeze(-1), p, data.edge_attr, sn)
        arrays[f"out/{mode}/p"] = p.numpy()
        arrays[f"out/{mode}/y"] = y.numpy()
        arrays[f"out/{mode}/h_last"] = g.ndata["h"].numpy()
    save(name, **arrays)


def dgl_pna_case(name, hidden, L, k, towers, edge_dim, sizes, seed):
    """The DGL tree's PNA base network (nets/ZINC_graph_regression/pna_net.py + layers/pna_layer.py, pna_utils.py; config
    PNA_ZINC_LapPE_signinv_GIN.json scaled down: 4 aggregators x 3 scalers, towers, edge features, graph_norm, no GRU)."""
    mods = _fresh_import("GraphPrediction", ["nets.ZINC_graph_regression.pna_net"])
    avg_d = dict(lin=torch.tensor(2.2), exp=torch.tensor(0.6), log=torch.tensor(1.1))       # main_ZINC_graph_regression.py:400-405
    params = dict(num_atom_type=28, num_bond_type=4, hidden_dim=hidden, out_dim=hidden, in_feat_dropout=0.0, dropout=0.0, L=L,
                  readout="sum", graph_norm=True, batch_norm=True, residual=True, aggregators="mean max min std",
                  scalers="identity amplification attenuation", avg_d=avg_d, towers=towers, divide_input_first=True,
                  divide_input_last=True, edge_feat=True, edge_dim=edge_dim, pretrans_layers=1, posttrans_layers=1, gru=False,
                  device="cpu", pe_init="lap_pe", lap_method="sign_inv", lap_lspe=False, use_lapeig_loss=False, lambda_loss=1000,
                  alpha_loss=1e-4, pos_enc_dim=k, sign_inv_net="gin", sign_inv_layers=3, sign_inv_activation="relu",
                  pe_aggregate="add", phi_out_dim=4)
    torch.manual_seed(seed)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        net = mods[0].PNANet(params)
    randomise(net, seed + 1)
    data = synth.make_batch(len(sizes), seed=seed, sizes=sizes)
    extra = {"meta/params": np.array([hidden, L, k, towers, edge_dim], dtype=np.int64),
             "meta/avg_d": np.array([float(avg_d["lin"]), float(avg_d["exp"]), float(avg_d["log"])], dtype=np.float64)}
    _run_dgl_net(name, net, data, k, extra, snorm=True)


def dgl_transformer_case(name, hidden, L, k, heads, sizes, seed, pe_aggregate):
    """The DGL tree's sparse graph Transformer (nets/ZINC_graph_regression/transformer_net.py + layers/transformer.py; config
    Transformer_ZINC_LapPE_signinv_GIN.json scaled down: full_graph False, edge features, attention over the graph's edges)."""
    mods = _fresh_import("GraphPrediction", ["nets.ZINC_graph_regression.transformer_net"])
    params = dict(num_atom_type=28, num_bond_type=4, hidden_dim=hidden, out_dim=hidden, in_feat_dropout=0.0, dropout=0.0, L=L,
                  n_heads=heads, full_graph=False, readout="sum", batch_norm=True, layer_norm=True, residual=True, edge_feat=True,
                  device="cpu", pe_init="lap_pe", lap_method="sign_inv", lap_lspe=False, use_lapeig_loss=False, lambda_loss=1,
                  alpha_loss=1e-4, pos_enc_dim=k, sign_inv_net="gin", sign_inv_layers=3, sign_inv_activation="relu",
                  pe_aggregate=pe_aggregate, phi_out_dim=4)
    torch.manual_seed(seed)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        net = mods[0].TransformerNet(params)
    randomise(net, seed + 1)
    data = synth.make_batch(len(sizes), seed=seed, sizes=sizes)
    extra = {"meta/params": np.array([hidden, L, k, heads], dtype=np.int64), "meta/pe_aggregate": np.array(pe_aggregate)}
    _run_dgl_net(name, net, data, k, extra)


# ------------------------------------------------------------------ LearningFilters
def dgl_gat_case(name, hidden, L, k, heads, sizes, seed):
    """The DGL tree's GAT base network with a sign-invariant PE (nets/ZINC_graph_regression/gat_net.py on dgl GATConv, config
    GAT_ZINC_LapPE_signinv_GIN.json scaled down)."""
    mods = _fresh_import("GraphPrediction", ["nets.ZINC_graph_regression.gat_net"])
    params = dict(num_atom_type=28, num_bond_type=4, hidden_dim=hidden, out_dim=hidden, in_feat_dropout=0.0, dropout=0.0, L=L,
                  readout="mean", batch_norm=True, residual=True, edge_feat=True, device="cpu", pe_init="lap_pe",
                  lap_method="sign_inv", lap_lspe=False, use_lapeig_loss=False, lambda_loss=1, alpha_loss=1e-4, pos_enc_dim=k,
                  sign_inv_net="gin", sign_inv_layers=3, sign_inv_activation="relu", pe_aggregate="concat", phi_out_dim=4, n_heads=heads)
    torch.manual_seed(seed)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        net = mods[0].GATNet(params)
    randomise(net, seed + 1)
    with torch.no_grad():
        for n_, p_ in net.named_parameters():
            if n_.startswith("layers.") and n_.endswith(".bias") and n_.count(".") == 2:
                p_.copy_(0.1 * torch.randn(p_.shape, generator=torch.Generator().manual_seed(seed + 2)))     # GATConv.bias (zero-initialised)
    data = synth.make_batch(len(sizes), seed=seed, sizes=sizes)
    _run_dgl_net(name, net, data, k, {"meta/params": np.array([hidden, L, k, heads], dtype=np.int64)})


def reference_grouping(eigvals, eigvecs):
    """Eigenspace grouping by EXECUTING the reference's own statements: LearningFilters/training.py is a script (argparse and
    dataset loading at module level), so lines 47-73 — `around()` and the whole `if args.lap_method == 'basis_inv':` block — are read
    from the reference file at generation time and exec'd in a namespace that supplies the names the script has defined by
    then (`eigvals`, `eigvecs`, `N`, `args`, `torch`).  Nothing of that source is stored in the repo; the fixture holds the
    resulting arrays only.  Returns (counts, uniq_mults, {mult: [b,1,N,N]})."""
    import contextlib
    import io
    lines = open(os.path.join(REF, "LearningFilters", "training.py")).read().splitlines()
    block = "\n".join(lines[46:73])                    # 1-based lines 47..73
    assert block.lstrip().startswith("def around(") and "same_size_projs[mult] = torch.cat(projs, dim=0)" in block, \
        "the reference file moved: re-check the line range of the grouping block"
    ns = {"torch": torch, "eigvals": eigvals, "eigvecs": eigvecs, "N": eigvecs.shape[0],
          "args": types.SimpleNamespace(lap_method="basis_inv")}
    with contextlib.redirect_stdout(io.StringIO()):
        exec(compile(block, "reference:LearningFilters/training.py:47-73", "exec"), ns)
    return ns["counts"], ns["uniq_mults"], ns["same_size_projs"]


def grid_eig(side, normalised_f64=False):
    """Eigen-data of the side x side grid as the reference computes it (utils.py:67-78: dense sym-normalised Laplacian in float64,
    scipy/numpy eigh, then `.float()` in training.py:42-43)."""
    ei, N = synth.grid_graph(side)
    A = np.zeros((N, N))
    A[ei[0], ei[1]] = 1.0
    dis = 1.0 / np.sqrt(A.sum(1))
    L = np.eye(N) - np.diag(dis) @ A @ np.diag(dis)
    w, V = np.linalg.eigh(L)
    return ei, N, torch.from_numpy(w).float(), torch.from_numpy(V).float()


def grouping_case(name, sides):
    """Fixture of the grouping itself, pinned to the reference's statements: eigenvalue multiplicities, the
    multiplicity groups and every projector.  Small grids store the projector stacks in full; larger ones store, per projector,
    the two vectors the IGN 2->1 layer reads from it (diagonal and row sums, ign.py:344-374) plus its trace and total."""
    arrays = {"meta/sides": np.array(sides, dtype=np.int64)}
    for side in sides:
        ei, N, D, V = grid_eig(side)
        counts, uniq_mults, projs = reference_grouping(D, V)
        t = f"s{side}"
        arrays[f"in/{t}/eigvals"], arrays[f"in/{t}/eigvecs"] = D.numpy(), V.numpy()
        arrays[f"out/{t}/counts"] = counts.numpy().astype(np.int64)
        arrays[f"out/{t}/mults"] = uniq_mults.numpy().astype(np.int64)
        for m, P in projs.items():
            assert P.shape[1:] == (1, N, N)
            if N <= 64:
                arrays[f"out/{t}/proj_m{m}"] = P.numpy()
            Pd = P[:, 0].double()
            arrays[f"out/{t}/sig_m{m}"] = torch.stack([torch.diagonal(Pd, dim1=1, dim2=2), Pd.sum(2)], dim=2).float().numpy()   # [b,N,2]
            arrays[f"out/{t}/tr_m{m}"] = torch.diagonal(Pd, dim1=1, dim2=2).sum(1).numpy()
            arrays[f"out/{t}/tot_m{m}"] = Pd.sum((1, 2)).numpy()
    save(name, **arrays)


def basisnet_case(name, side, hidden, seed):
    ign, sbn, models = _fresh_import("LearningFilters", ["ign", "signbasisnet", "models"])
    from oracle import basisnet as ob
    ei, N = synth.grid_graph(side)
    D, V = synth.sym_laplacian_eigh(ei, N)
    counts, _, groups = reference_grouping(D, V)        # training.py:47-73 executed (module-level code of a script)
    mults = sorted(groups)
    torch.manual_seed(seed)
    # IGNBasisInv builds IGN2to1 with device='cuda' (signbasisnet.py:33) -> build the pieces on CPU
    encs = [ign.IGN2to1(1, hidden, m, num_layers=2, device="cpu") for m in mults]
    rho = models.EqDeepSetsEncoder(2 * N, hidden_channels=10, out_channels=8, num_layers=3, use_bn=True)
    sign = sbn.SignPlus(models.EqDeepSetsEncoder(1, num_layers=3, use_bn=True))
    for i, e in enumerate(encs):
        randomise(e, seed + 10 + i)
    randomise(rho, seed + 2)
    randomise(sign, seed + 3)
    arrays = {"in/eigvals": D.numpy(), "in/eigvecs": V.numpy(), "in/edge_index": ei,
              "meta/mults": np.array(mults, dtype=np.int64), "meta/hidden": np.array(hidden)}
    for m, e in zip(mults, encs):
        for k, v in e.state_dict().items():
            arrays[f"sd/enc{m}/{k}"] = v.clone().numpy()
        for li, layer in enumerate(e.equi_layers):
            arrays[f"eq/enc{m}/{li}/coeffs"] = layer.coeffs.detach().clone().numpy()
            arrays[f"eq/enc{m}/{li}/bias"] = layer.bias.detach().clone().numpy()
    for k, v in rho.state_dict().items():
        arrays["sd/rho/" + k] = v.clone().numpy()
    for k, v in sign.state_dict().items():
        arrays["sd/sign/" + k] = v.clone().numpy()
    for mode in ("eval", "train"):
        outs = []
        for m, e in zip(mults, encs):
            e.train(mode == "train")
            with torch.no_grad():
                outs.append(e(groups[m]))
            arrays[f"out/{mode}/phi_m{m}"] = outs[-1].numpy()
        feats = ob.basis_inv_features(outs, D, N)       # restates training.py:119-123
        rho.train(mode == "train")
        with torch.no_grad():
            arrays[f"out/{mode}/rho"] = rho(feats).numpy()
    sign.eval()
    with torch.no_grad():
        arrays["out/eval/signplus"] = sign(V.transpose(1, 0).unsqueeze(-1)).numpy()   # training.py:101-102
    save(name, **arrays)


def reference_filter_script(args, eigvals, eigvecs, data, y):
    """The LearningFilters entry script from `around()` to `gen_rho` (training.py:47-223: eigenspace grouping, PE_DIM, get_lap_feat,
    train, gen_model and the other factories) EXECUTED from the reference file at generation time, in a namespace holding what the
    script has defined by then (args, data, y, eigvals, eigvecs, N, device, the imported model classes, r2_score).  As in
    reference_grouping nothing of that source is stored here.  IGNBasisInv builds its IGN2to1 with device='cuda'
    (signbasisnet.py:33, ign.py:12): on this CPU-only box the name is bound to the same class with device='cpu'.
    Returns the namespace (gen_model, get_lap_feat, train, same_size_projs, ...)."""
    import contextlib
    import functools
    import io
    ign, sbn, models = _fresh_import("LearningFilters", ["ign", "signbasisnet", "models"])
    sbn.IGN2to1 = functools.partial(ign.IGN2to1, device="cpu")
    lines = open(os.path.join(REF, "LearningFilters", "training.py")).read().splitlines()
    block = "\n".join(lines[46:223])                  # 1-based lines 47..223
    assert block.lstrip().startswith("def around(") and block.rstrip().endswith("return rho"), \
        "the reference file moved: re-check the line range of the script block"
    from sklearn.metrics import r2_score
    ns = {"torch": torch, "np": np, "eigvals": eigvals, "eigvecs": eigvecs, "N": eigvecs.shape[0], "args": args, "data": data, "y": y,
          "device": torch.device("cpu"), "r2_score": r2_score, "SignPlus": sbn.SignPlus, "IGNBasisInv": sbn.IGNBasisInv,
          "IGNShared": sbn.IGNShared}
    for n in ("ChebNet", "BernNet", "GcnNet", "GatNet", "ARMANet", "GPRNet", "MLP", "EqDeepSetsEncoder", "Transformer"):
        ns[n] = getattr(models, n)
    with contextlib.redirect_stdout(io.StringIO()):
        exec(compile(block, "reference:LearningFilters/training.py:47-223", "exec"), ns)
    return ns


def filters_case(name, side, cases, seed):
    """the LearningFilters training workload on a side x side grid.  Per case (an argparse namespace of
    training.py:12-24): the model from the reference's gen_model, its prediction, then `train()` (training.py:132-150) called
    four times — the loss of every step, the parameter gradients of the first step (from the reference's loss.backward()) and the
    parameters after the first torch.optim.Adam step."""
    ei, N, D, V = grid_eig(side)
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, 2, generator=g)
    yv = torch.randn(N, 2, generator=g)
    m = torch.ones(N, 1)
    idx = np.arange(N)
    r, c = idx // side, idx % side
    m[(r == 0) | (c == 0) | (r == side - 1) | (c == side - 1)] = 0.0        # the boundary mask of utils.py's TwoDGrid (data.m)
    arrays = {"in/eigvals": D.numpy(), "in/eigvecs": V.numpy(), "in/x": x.numpy(), "in/y": yv.numpy(), "in/m": m.numpy(),
              "meta/side": np.array(side), "meta/cases": np.array([c["name"] for c in cases])}
    for ci, c in enumerate(cases):
        a = dict(epochs=4, lr=0.01, filter_type="band", net="DS", img_num=1, use_eig=True, lap_method="none", sign_inv_net="DS",
                 basis_inv_net="IGN", hidden_channels=32, num_layers=2)
        a.update({k: v for k, v in c.items() if k != "name"})
        args = types.SimpleNamespace(**a)
        data = types.SimpleNamespace(x=x.clone(), m=m.clone(), edge_index=torch.from_numpy(ei))
        ns = reference_filter_script(args, D, V, data, yv.clone())
        torch.manual_seed(seed + ci)
        model = ns["gen_model"](args)
        randomise(model, seed + 10 + ci)
        t = "c/" + c["name"] + "/"
        for k, v in a.items():
            arrays[t + "args/" + k] = np.array(v)
        if a["lap_method"] == "basis_inv":
            arrays[t + "mults"] = np.array(sorted(ns["same_size_projs"]), dtype=np.int64)
        for k, v in model.state_dict().items():
            arrays[t + "sd/" + k] = v.detach().clone().numpy()
        model.train()
        with torch.no_grad():
            feat = ns["get_lap_feat"](args.use_eig, D, V, x[:, 0:1], args.lap_method, model)
            arrays[t + "feat"] = feat.numpy()
            arrays[t + "pre"] = model(feat, data.edge_index).numpy()
        opt = torch.optim.Adam(model.parameters(), lr=args.lr)
        losses = []
        for step in range(4):
            loss, r2 = ns["train"](0, model, opt)
            losses.append(loss)
            if step == 0:
                arrays[t + "r2"] = np.array(r2)
                for k, p_ in model.named_parameters():
                    arrays[t + "grad/" + k] = (torch.zeros_like(p_) if p_.grad is None else p_.grad).detach().clone().numpy()
                for k, v in model.state_dict().items():
                    arrays[t + "sd1/" + k] = v.detach().clone().numpy()
        arrays[t + "losses"] = np.array(losses, dtype=np.float64)
    save(name, **arrays)


# ------------------------------------------------------------------ eigendecomposition transform
def evd_case(name, sizes, seed):
    """The reference's own EVDTransform (Alchemy/sign_net/transform.py:7-23) per sample, both normalisations.
    (GINESignNetPyG/core/transform.py holds the same code.)  Includes a self loop, a duplicated and a one-directional
    edge so that to_undirected / get_laplacian's clean-up is exercised."""
    (mod,) = _fresh_import("Alchemy", ["sign_net.transform"])
    data = synth.make_batch(len(sizes), seed=seed, sizes=sizes, features="zinc")
    ei = data.edge_index
    extra = []
    off = 0
    for n in sizes:
        if n >= 3:
            extra += [(off, off), (off, off + 2), (off + 1, off + 2), (off + 1, off + 2)]   # loop, one-way, duplicate
        off += n
    if extra:
        ei = torch.cat([ei, torch.tensor(extra, dtype=torch.int64).t()], 1)
    arrays = {"in/edge_index": ei.numpy(), "in/sizes": np.array(sizes, dtype=np.int64)}
    for norm in (None, "sym"):
        vals, vecs = [], []
        off = 0
        for n in sizes:
            sel = (ei[0] >= off) & (ei[0] < off + n)
            sample = types.SimpleNamespace(edge_index=ei[:, sel] - off, num_nodes=n)
            out = mod.EVDTransform(norm)(sample)
            vals.append(out.eigen_values.numpy())
            vecs.append(out.eigen_vectors.numpy())
            off += n
        tag = "none" if norm is None else norm
        arrays[f"out/{tag}/eigen_values"] = np.concatenate(vals)
        arrays[f"out/{tag}/eigen_vectors"] = np.concatenate(vecs)
    save(name, **arrays)


def main():
    # GINESignNetPyG: SignNetGNN(None, None, n_hid, n_out, nl_signnet, nl_gnn)
    pyg_case("gine_d16", "gine", (None, None, 16, 1, 3, 2), [5, 7, 6, 9], "zinc", 11)
    pyg_case("gine_d44_ragged", "gine", (None, None, 44, 3, 2, 2), [1, 2, 12, 4, 3], "zinc", 12)
    pyg_case("gine_d32_deep", "gine", (None, None, 32, 1, 4, 6), [9, 17, 11, 20, 10, 13], "zinc", 13)
    # Alchemy: SignNetGNN(node_feat, edge_feat, n_hid, n_out, nl_signnet, nl_gnn)  (nl_rho fixed at 4)
    pyg_case("alchemy_d12", "alchemy", (6, 4, 12, 5, 3, 3), [6, 9, 7, 8], "alchemy", 21)
    pyg_case("alchemy_d36", "alchemy", (6, 4, 36, 12, 2, 3), [10, 6, 14, 9, 11], "alchemy", 22)
    # DGL sign-inv nets
    dgl_case("dgl_gin_k8", "gin", 24, 4, 3, 8, [5, 9, 12, 7], 31)
    dgl_case("dgl_masked_k10", "masked_gin", 20, 20, 3, 10, [5, 13, 8, 11], 32)
    dgl_ginnet_case("dgl_ginnet_k6", 24, 3, 6, [5, 9, 12, 7, 3], 33)
    dgl_gatedgcn_case("dgl_gatedgcn_concat_k6", 20, 3, 6, [5, 9, 12, 7, 3], 34, "concat")
    dgl_gatedgcn_case("dgl_gatedgcn_add_k8", 28, 2, 8, [6, 4, 11, 2], 35, "add")
    dgl_pna_case("dgl_pna_k6", 20, 3, 6, 5, 8, [5, 9, 12, 7, 3], 36)
    dgl_transformer_case("dgl_transformer_concat_k6", 24, 3, 6, 4, [5, 9, 12, 7, 3], 37, "concat")
    dgl_transformer_case("dgl_transformer_add_k8", 32, 2, 8, 8, [6, 4, 11, 2], 38, "add")
    dgl_gat_case("dgl_gat_k6", 12, 3, 6, 4, [5, 9, 12, 7, 3], 39)
    # BasisNet on a small grid
    basisnet_case("basisnet_grid6", 6, 8, 41)
    grouping_case("basisnet_grouping", [6, 12, 32])
    filters_case("learning_filters_grid6", 6, [
        dict(name="ds_signinv_ds", net="DS", hidden_channels=32, num_layers=3, lap_method="sign_inv", sign_inv_net="DS"),
        dict(name="ds_basisinv_ign", net="DS", hidden_channels=16, num_layers=2, lap_method="basis_inv", basis_inv_net="IGN"),
        dict(name="tf_signinv_ds", net="Transformer", hidden_channels=16, num_layers=2, lap_method="sign_inv", sign_inv_net="DS"),
        dict(name="tf_basisinv_ign", net="Transformer", hidden_channels=12, num_layers=2, lap_method="basis_inv", basis_inv_net="IGN"),
        dict(name="mlp_signinv_mlp", net="MLP", hidden_channels=24, num_layers=3, lap_method="sign_inv", sign_inv_net="MLP"),
        dict(name="linear_signinv_tf", net="Linear", lap_method="sign_inv", sign_inv_net="Transformer"),
        dict(name="ds_basisinv_shared", net="DS", hidden_channels=16, num_layers=2, lap_method="basis_inv", basis_inv_net="IGNShared"),
        dict(name="tf_eig_none", net="Transformer", hidden_channels=20, num_layers=3, lap_method="none"),
    ], 61)
    # eigendecomposition transform: sizes across the kernel's 16 / 32 / 64-lane classes, odd and even, 1- and 2-node graphs
    evd_case("evd_transform", [1, 2, 3, 8, 15, 16, 17, 23, 31, 32, 33, 37, 48, 63, 64], 51)


if __name__ == "__main__":
    main()
