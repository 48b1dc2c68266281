"""Oracle (test infrastructure): BasisNet / LearningFilters forward pieces, restated functionally.

Reference: /root/reference/LearningFilters/ign.py:29-39 (IGN2to1.forward), :117-128 (layer_2_to_1),
:203-214 (layer_1_to_1), :344-374 / :404-417 (contractions); signbasisnet.py:11-41 (SignPlus,
IGNBasisInv); models.py:58-113 (EqDeepSetsEncoder); training.py:47-73,119-126 (eigenspace grouping
and the basis_inv feature assembly).

IGN2to1 equivariant coefficients are NOT in the module's state_dict when the module is built on a
non-default device, so they are passed explicitly: `eq` is a list of
(coeffs [D,S,B], bias [1,S,1]) for the three equivariant layers.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

BN_EPS = 1e-5


def _bn_cn(sd, pfx, x, training):
    """BatchNorm1d on [b, C, n] (stats over b*n)."""
    if training or (pfx + ".running_mean") not in sd:
        return F.batch_norm(x, None, None, sd[pfx + ".weight"], sd[pfx + ".bias"], True, 0.0, BN_EPS)
    return F.batch_norm(x, sd[pfx + ".running_mean"], sd[pfx + ".running_var"],
                        sd[pfx + ".weight"], sd[pfx + ".bias"], False, 0.0, BN_EPS)


def contractions_2_to_1(X):
    """ign.py:344-374 with normalization='inf'.  X [b,D,n,n] -> [b,D,5,n]."""
    n = X.shape[-1]
    diag = torch.diagonal(X, dim1=2, dim2=3)
    tr = diag.sum(2, keepdim=True).expand(-1, -1, n) / n
    rows = X.sum(3) / n
    cols = X.sum(2) / n
    tot = X.sum((2, 3)).unsqueeze(2).expand(-1, -1, n) / (n ** 2)
    return torch.stack([diag, tr, rows, cols, tot], dim=2)


def contractions_1_to_1(h):
    """ign.py:404-417.  h [b,D,n] -> [b,D,2,n]."""
    n = h.shape[-1]
    return torch.stack([h, h.sum(2, keepdim=True).expand(-1, -1, n) / n], dim=2)


def ign2to1(sd, eq, X, training=False, pfx=""):
    """IGN2to1.forward — ign.py:29-39 (ReLU BEFORE BatchNorm)."""
    (c0, b0), (c1, b1), (c2, b2) = eq
    h = torch.einsum("dsb,ndbi->nsi", c0, contractions_2_to_1(X)) + b0
    h = _bn_cn(sd, pfx + "bns.0", torch.relu(h), training)
    h = torch.einsum("dsb,ndbi->nsi", c1, contractions_1_to_1(h)) + b1
    h = _bn_cn(sd, pfx + "bns.1", torch.relu(h), training)
    h = torch.einsum("dsb,ndbi->nsi", c2, contractions_1_to_1(h)) + b2
    h = _bn_cn(sd, pfx + "bns.2", torch.relu(h), training)
    h = h.transpose(2, 1)
    h = torch.relu(F.linear(h, sd[pfx + "fc1.weight"], sd[pfx + "fc1.bias"]))
    h = F.linear(h, sd[pfx + "fc2.weight"], sd[pfx + "fc2.bias"])
    return h.transpose(2, 1)


def ign_shared(sd, eq, X, mult_idx, training=False):
    """IGNShared.forward — signbasisnet.py:57-64: shared IGN2to1(1,hidden,1) then Linear(1, mult) on the transposed output."""
    x = ign2to1(sd, eq, X, training, pfx="enc.")                  # [b, 1, n]
    x = x.transpose(2, 1)
    x = F.linear(x, sd[f"fcs.{mult_idx}.weight"], sd[f"fcs.{mult_idx}.bias"])
    return x.transpose(2, 1)


def eq_deepsets(sd, x, num_layers, use_bn, pfx=""):
    """EqDeepSetsEncoder.forward — models.py:91-113.  BN has track_running_stats=False, so it
    always normalises with batch statistics (models.py:74,80)."""
    for i in range(num_layers - 1):
        x1 = F.linear(x, sd[f"{pfx}lins1.{i}.weight"], sd[f"{pfx}lins1.{i}.bias"])
        x2 = F.linear(x.mean(dim=-2, keepdim=True), sd[f"{pfx}lins2.{i}.weight"], sd[f"{pfx}lins2.{i}.bias"])
        x = torch.relu(x1 + x2)
        if use_bn:
            w, b = sd[f"{pfx}bns.{i}.weight"], sd[f"{pfx}bns.{i}.bias"]
            if x.dim() == 2:
                x = F.batch_norm(x, None, None, w, b, True, 0.0, BN_EPS)
            else:
                x = F.batch_norm(x.transpose(2, 1), None, None, w, b, True, 0.0, BN_EPS).transpose(2, 1)
    i = num_layers - 1
    x1 = F.linear(x, sd[f"{pfx}lins1.{i}.weight"], sd[f"{pfx}lins1.{i}.bias"])
    x2 = F.linear(x.mean(dim=-2, keepdim=True), sd[f"{pfx}lins2.{i}.weight"], sd[f"{pfx}lins2.{i}.bias"])
    return x1 + x2


def sign_plus_deepsets(sd, v, num_layers, use_bn, pfx="model.", x=None):
    """SignPlus.forward — signbasisnet.py:16-20 with an EqDeepSetsEncoder inside (x: side features, concatenated un-negated)."""
    if x is not None:
        return (eq_deepsets(sd, torch.cat((v, x), dim=-1), num_layers, use_bn, pfx) +
                eq_deepsets(sd, torch.cat((-v, x), dim=-1), num_layers, use_bn, pfx))
    return eq_deepsets(sd, v, num_layers, use_bn, pfx) + eq_deepsets(sd, -v, num_layers, use_bn, pfx)


def group_eigenspaces(eigvals, eigvecs, decimals=5):
    """training.py:47-73: round eigenvalues, group eigenvectors by rounded value, P = V V^T,
    stack by multiplicity.  Returns {mult: [b,1,N,N]} in ascending-eigenvalue order."""
    N = eigvecs.shape[0]
    rounded = torch.round(eigvals * 10 ** decimals) / (10 ** decimals)
    _, counts = rounded.unique(return_counts=True)
    sections = torch.cumsum(counts, 0)
    spaces = torch.tensor_split(eigvecs, sections, dim=1)[:-1]
    groups = {}
    for V, c in zip(spaces, counts.tolist()):
        groups.setdefault(c, []).append((V @ V.T).reshape(1, 1, N, N))
    return {m: torch.cat(ps, 0) for m, ps in sorted(groups.items())}, counts


def basis_inv_features(phi_outs, eigvals, N):
    """training.py:119-123: `phi_out.reshape(N, -1)` (a plain reshape of [b,mult,N], as the
    reference does) concatenated, then eigvals tiled [N,N] appended -> [N, 2N]."""
    feats = torch.cat([p.reshape(N, -1) for p in phi_outs], dim=-1)
    return torch.cat([feats, eigvals.unsqueeze(0).repeat(N, 1)], dim=-1)


# --------------------------------------------------------------------------------------------- SURVEY.md §8 row f4
LN_EPS = 1e-5


def mlp(sd, x, num_layers, use_bn=False, use_ln=False, pfx=""):
    """MLP.forward — models.py:43-56: Linear, ReLU, [BatchNorm1d batch statistics], [LayerNorm] per hidden layer; last Linear."""
    for i in range(num_layers - 1):
        x = torch.relu(F.linear(x, sd[f"{pfx}lins.{i}.weight"], sd[f"{pfx}lins.{i}.bias"]))
        if use_bn:
            w, b = sd[f"{pfx}bns.{i}.weight"], sd[f"{pfx}bns.{i}.bias"]
            if x.dim() == 2:
                x = F.batch_norm(x, None, None, w, b, True, 0.0, BN_EPS)
            else:
                x = F.batch_norm(x.transpose(2, 1), None, None, w, b, True, 0.0, BN_EPS).transpose(2, 1)
        if use_ln:
            x = F.layer_norm(x, x.shape[-1:], sd[f"{pfx}lns.{i}.weight"], sd[f"{pfx}lns.{i}.bias"], LN_EPS)
    i = num_layers - 1
    return F.linear(x, sd[f"{pfx}lins.{i}.weight"], sd[f"{pfx}lins.{i}.bias"])


def dense_attention(q, k, v, heads):
    """nn.MultiheadAttention's core on [Bt, L, d]: per head softmax(q k^T / sqrt(dk)) v."""
    Bt, L, d = q.shape
    dk = d // heads
    q, k, v = (t.reshape(Bt, L, heads, dk).transpose(1, 2) for t in (q, k, v))              # [Bt, heads, L, dk]
    p = torch.softmax(q @ k.transpose(-1, -2) / dk ** 0.5, dim=-1)
    return (p @ v).transpose(1, 2).reshape(Bt, L, d)


def transformer(sd, x, num_layers, num_heads=4, pfx=""):
    """Transformer.forward — models.py:124-135, with nn.TransformerEncoderLayer(norm_first=True, batch_first=True, ReLU,
    dropout 0) written out: x = x + out_proj(attention(in_proj(norm1(x)))); x = x + linear2(relu(linear1(norm2(x))))."""
    x = F.linear(x, sd[pfx + "fc1.weight"], sd[pfx + "fc1.bias"])
    two_d = x.dim() == 2
    if two_d:
        x = x.unsqueeze(0)
    d = x.shape[-1]
    for i in range(num_layers):
        e = f"{pfx}encs.{i}."
        u = F.layer_norm(x, (d,), sd[e + "norm1.weight"], sd[e + "norm1.bias"], LN_EPS)
        qkv = F.linear(u, sd[e + "self_attn.in_proj_weight"], sd[e + "self_attn.in_proj_bias"])
        o = dense_attention(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], num_heads)
        x = x + F.linear(o, sd[e + "self_attn.out_proj.weight"], sd[e + "self_attn.out_proj.bias"])
        u = F.layer_norm(x, (d,), sd[e + "norm2.weight"], sd[e + "norm2.bias"], LN_EPS)
        x = x + F.linear(torch.relu(F.linear(u, sd[e + "linear1.weight"], sd[e + "linear1.bias"])), sd[e + "linear2.weight"],
                         sd[e + "linear2.bias"])
    if two_d:
        x = x.squeeze(0)
    return F.linear(x, sd[pfx + "fc2.weight"], sd[pfx + "fc2.bias"])


def _base_net(sd, cfg, x, kind, num_layers, pfx, use_bn=False):
    if kind == "DS":
        return eq_deepsets(sd, x, num_layers, use_bn, pfx)
    if kind in ("MLP", "Linear"):
        return mlp(sd, x, 1 if kind == "Linear" else num_layers, use_bn, False, pfx)
    if kind == "Transformer":
        return transformer(sd, x, num_layers, 4, pfx)
    raise ValueError(kind)


def _eq_of(sd, pfx):
    return [(sd[f"{pfx}equi_layers.{j}.coeffs"], sd[f"{pfx}equi_layers.{j}.bias"]) for j in range(3)]


def lap_feat(sd, cfg, feat, eigvals, eigvecs, groups=None):
    """get_lap_feat — training.py:87-130 for lap_method 'none' / 'sign_inv' / 'basis_inv'.  cfg: dict(net, hidden_channels, num_layers,
    use_eig, lap_method, sign_inv_net, basis_inv_net, mults).  groups: {mult: [b,1,N,N]} (training.py:63-73's same_size_projs)."""
    if not cfg["use_eig"]:
        return feat
    N = eigvecs.shape[0]
    eigvals_mat = eigvals.unsqueeze(0).repeat(N, 1)
    lm = cfg["lap_method"]
    if lm == "none":
        return torch.cat((feat, eigvecs, eigvals_mat), dim=-1)
    if "sign_inv" in lm:
        v = eigvecs.transpose(1, 0).unsqueeze(-1)
        kind = cfg["sign_inv_net"]                                                            # gen_sign_inv, :183-199
        nl = {"DS": 3, "MLP": cfg["num_layers"], "Transformer": 2}[kind]
        f = lambda t: _base_net(sd, cfg, t, kind, nl, "sign_inv_net.model.", use_bn=kind != "Transformer")
        ef = (f(v) + f(-v)).transpose(1, 0).reshape(N, -1)
    elif "basis_inv" in lm:
        outs = []
        for i, m in enumerate(cfg["mults"]):
            if cfg["basis_inv_net"] == "IGN":
                p = f"basis_inv_net.encs.{i}."
                outs.append(ign2to1(sd, _eq_of(sd, p), groups[m], True, pfx=p))
            else:
                outs.append(ign_shared({k[len("basis_inv_net."):]: t for k, t in sd.items() if k.startswith("basis_inv_net.")},
                                       _eq_of(sd, "basis_inv_net.enc."), groups[m], i, True))
        ef = torch.cat([o.reshape(N, -1) for o in outs], dim=-1)
    else:
        raise ValueError(lm)
    ef = torch.cat((ef, eigvals_mat), dim=-1)
    ef = eq_deepsets(sd, ef, 3, True, "rho.")                                                 # gen_rho, :212-218
    return torch.cat((feat, ef), dim=-1)


def filter_model(sd, cfg, x, eigvals, eigvecs, groups=None):
    """pre = model(get_lap_feat(...), edge_index) — training.py:136-138 with gen_model's base network (:152-181)."""
    feat = lap_feat(sd, cfg, x, eigvals, eigvecs, groups)
    return _base_net(sd, cfg, feat, cfg["net"], cfg["num_layers"], "")


def filter_loss(pre, y, m):
    """training.py:139."""
    return torch.square(m * (pre - y)).sum()
