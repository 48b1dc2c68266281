"""Oracle (test infrastructure): the PyG-tree SignNet + GINE forward, restated functionally.

Covers both near-twin trees:
  variant="alchemy"  -> /root/reference/Alchemy/sign_net/{sign_net,model,transform}.py
  variant="gine"     -> /root/reference/GINESignNetPyG/core/{sign_net,model,transform}.py
Third-party semantics restated from their published definitions (absent from the image):
  torch_geometric==2.0.1 GINConv / GINEConv, torch_scatter.scatter.

Everything is fp32 torch on the CPU.  `sd` is a reference-keyed state_dict.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

BN_EPS = 1e-5      # nn.BatchNorm1d default (masked_layers.py:10)
LN_EPS = 1e-6      # masked_layers.py:25
N_HEAD = 4         # sign_net.py:50 / core/sign_net.py:57  (TransformerEncoderLayer(nhid, n_head=4))


# --------------------------------------------------------------------------- helpers
# TERM_PROBE (tests only; None = off): the gradient of a ONE-ENTRY parameter (GIN / GINE eps, GINESignNetPyG's Linear(1, 1) and
# BatchNorm1d(1)) is one sum over ~10^5 rows; its fp32 noise scales with the TERMS of that sum, not with the (cancelling) result.
# With a dict here, every such parameter enters the arithmetic as a per-row copy of itself (same values, same result) whose .grad —
# retained — is the vector of per-row terms: tests/test_training_gpu.py prices the one-entry gradients with eps32 * sqrt(sum t_r^2).
TERM_PROBE = None


def _rows_of(key, t, lead_shape):
    """Per-row copy of the one-entry parameter `t` (shape lead_shape + (1,)); its gradient is kept for the probe."""
    r = t.reshape(()).expand(tuple(lead_shape) + (1,)) * 1.0
    if r.requires_grad:
        r.retain_grad()
        TERM_PROBE[key] = r
    return r


def _eps(sd, key, x):
    e = sd[key]
    if TERM_PROBE is not None and e.numel() == 1 and e.requires_grad:
        return _rows_of(key, e, x.shape[:-1])
    return e


def _has(sd, key):
    return key in sd


def _linear(sd, pfx, x):
    """nn.Linear with reference keys `<pfx>.weight` (+ `.bias` when registered)."""
    W = sd[pfx + ".weight"]
    if TERM_PROBE is not None and W.numel() == 1 and W.requires_grad:
        y = x * _rows_of(pfx + ".weight", W, x.shape[:-1])
        b = sd.get(pfx + ".bias")
        return y if b is None else y + _rows_of(pfx + ".bias", b, x.shape[:-1])
    return F.linear(x, W, sd.get(pfx + ".bias"))


def _bn_rows(sd, pfx, rows, training):
    """BatchNorm1d on [M, C] rows.  eval: running stats; train: batch stats (biased var),
    running stats NOT updated here (forward value only)."""
    if training and TERM_PROBE is not None and rows.shape[-1] == 1 and sd[pfx + ".weight"].requires_grad:
        xh = F.batch_norm(rows, None, None, None, None, True, 0.0, BN_EPS)
        return xh * _rows_of(pfx + ".weight", sd[pfx + ".weight"], rows.shape[:-1]) + _rows_of(pfx + ".bias", sd[pfx + ".bias"], rows.shape[:-1])
    if training:
        return F.batch_norm(rows, None, None, sd[pfx + ".weight"], sd[pfx + ".bias"], True, 0.0, BN_EPS)
    return F.batch_norm(rows, sd[pfx + ".running_mean"], sd[pfx + ".running_var"],
                        sd[pfx + ".weight"], sd[pfx + ".bias"], False, 0.0, BN_EPS)


def masked_bn(sd, pfx, x, mask, training):
    """MaskedBN.forward — masked_layers.py:13-20: `x[mask] = bn(x[mask])`."""
    x = x.clone()
    if mask is None:
        return _bn_rows(sd, pfx + ".bn", x.reshape(-1, x.shape[-1]), training).reshape(x.shape)
    x[mask] = _bn_rows(sd, pfx + ".bn", x[mask], training)
    return x


def masked_ln(sd, pfx, x, mask):
    """MaskedLN.forward — masked_layers.py:28-32 (eps 1e-6)."""
    x = x.clone()
    w, b = sd[pfx + ".ln.weight"], sd[pfx + ".ln.bias"]
    if mask is None:
        return F.layer_norm(x, (x.shape[-1],), w, b, LN_EPS)
    x[mask] = F.layer_norm(x[mask], (x.shape[-1],), w, b, LN_EPS)
    return x


def masked_mlp(sd, pfx, x, mask, nlayer, final_act, training):
    """MaskedMLP.forward — masked_layers.py:54-64."""
    for i in range(nlayer):
        x = _linear(sd, f"{pfx}.layers.{i}", x)
        if mask is not None:
            x = x.masked_fill(~mask.unsqueeze(-1), 0.0)
        if i < nlayer - 1 or final_act:
            x = masked_bn(sd, f"{pfx}.norms.{i}", x, mask, training)
            x = torch.relu(x)
    return x


def plain_mlp(sd, pfx, x, nlayer, final_act, training):
    """MLP.forward — model_utils/elements.py:58-69 (Linear -> BN -> ReLU per non-final layer)."""
    for i in range(nlayer):
        x = _linear(sd, f"{pfx}.layers.{i}", x)
        if i < nlayer - 1 or final_act:
            if _has(sd, f"{pfx}.norms.{i}.weight"):
                x = _bn_rows(sd, f"{pfx}.norms.{i}", x, training)
            x = torch.relu(x)
    return x


def discrete_encoder(sd, pfx, idx):
    """DiscreteEncoder.forward — elements.py:31-37: sum_f embeddings[f](x[:, f])."""
    if idx.dim() == 1:
        idx = idx.unsqueeze(1)
    out = 0
    for f in range(idx.shape[1]):
        out = out + F.embedding(idx[:, f], sd[f"{pfx}.embeddings.{f}.weight"])
    return out


def gin_aggregate(x, edge_index, eps, node_dim=-2):
    """PyG GINConv pre-`nn` value: (1+eps) x_i + sum_{j->i} x_j along `node_dim`
    (source = edge_index[0], target = edge_index[1])."""
    src, dst = edge_index[0], edge_index[1]
    return torch.zeros_like(x).index_add_(node_dim, dst, x.index_select(node_dim, src)) + (1 + eps) * x


def gine_aggregate(x, edge_index, e, eps):
    """PyG GINEConv pre-`nn` value: (1+eps) x_i + sum_{j->i} relu(x_j + e_ji)."""
    src, dst = edge_index[0], edge_index[1]
    return torch.zeros_like(x).index_add_(0, dst, torch.relu(x.index_select(0, src) + e)) + (1 + eps) * x


# --------------------------------------------------------------------------- a1: packing
def to_dense_list_evd(eig_s, eig_v, batch, num_graphs=None, max_k=None):
    """to_dense_list_EVD — transform.py:26-61.  Returns eigS_dense [N,K], eigV_dense [N,K],
    mask_full [N,K] (sign_net.py:100-102) with K = N_max, or min(N_max, max_k) when the
    fixed-k reading of BASELINE.json is requested (first k columns)."""
    B = int(batch.max()) + 1 if num_graphs is None else num_graphs
    n = torch.zeros(B, dtype=torch.long).index_add_(0, batch, torch.ones_like(batch))
    nmax = int(n.max())
    K = nmax if max_k is None else min(nmax, int(max_k))
    start = torch.cumsum(n, 0) - n                    # first node of each graph
    vstart = torch.cumsum(n * n, 0) - n * n           # first eigvec entry of each graph
    N = batch.numel()
    local = torch.arange(N) - start[batch]
    cols = torch.arange(K)[None, :]
    nb = n[batch][:, None]
    mask = cols < nb                                   # [N,K]
    vidx = vstart[batch][:, None] + local[:, None] * nb + cols
    sidx = start[batch][:, None] + cols
    zero = torch.zeros((), dtype=eig_v.dtype)
    eig_v_dense = torch.where(mask, eig_v[vidx.clamp(max=eig_v.numel() - 1)], zero)
    eig_s_dense = torch.where(mask, eig_s[sidx.clamp(max=eig_s.numel() - 1)], zero)
    return eig_s_dense, eig_v_dense, mask


# --------------------------------------------------------------------------- a3-a6: phi
def gnn3d(sd, pfx, x, edge_index, mask, nlayer, training, trace=None):
    """GNN3d.forward — sign_net.py:28-44 [core/sign_net.py:30-48]."""
    x = x.transpose(0, 1)                              # [K,N,c]
    m = mask.transpose(0, 1)
    prev = 0
    for l in range(nlayer):
        a = gin_aggregate(x, edge_index, _eps(sd, f"{pfx}.convs.{l}.layer.eps", x))   # masked_layers.py:75
        h = masked_mlp(sd, f"{pfx}.convs.{l}.nn", a, m, 2, False, training)     # :83 / :79
        h = h.masked_fill(~m.unsqueeze(-1), 0.0)                               # sign_net.py:39
        h = masked_bn(sd, f"{pfx}.norms.{l}", h, m, training)
        h = torch.relu(h)
        x = h + prev
        prev = x
        if trace is not None:
            trace.append(x.transpose(0, 1).clone())
    return x.transpose(0, 1)


# --------------------------------------------------------------------------- a7-a8: rho
def encoder_layer(sd, pfx, x, mask, keep=None):
    """TransformerEncoderLayer.forward — transformer_module.py:34-42.  keep: the scaled Bernoulli keep-mask [N, heads, K, K] (0 or
    1 / (1 - p)) of the attention dropout that the reference leaves active in train mode (:46,55: `self.dropout(F.softmax(attn))`);
    None = no dropout (eval, or the deterministic train-mode fixtures)."""
    N, K, d = x.shape
    dk = d // N_HEAD
    pair = (mask.unsqueeze(1) * mask.unsqueeze(2)).unsqueeze(1)              # :78,:93  [N,1,K,K]
    res = x
    q = _linear(sd, pfx + ".slf_attn.w_qs", x).view(N, K, N_HEAD, dk).transpose(1, 2)
    k = _linear(sd, pfx + ".slf_attn.w_ks", x).view(N, K, N_HEAD, dk).transpose(1, 2)
    v = _linear(sd, pfx + ".slf_attn.w_vs", x).view(N, K, N_HEAD, dk).transpose(1, 2)
    att = torch.matmul(q / (dk ** 0.5), k.transpose(2, 3))                   # :52
    att = att.masked_fill(pair == 0, -1e10)                                  # :54
    att = torch.softmax(att, dim=-1)                                         # :55
    if keep is not None:
        att = att * keep.to(att.dtype)                                       # :55  nn.Dropout on the probabilities
    att = att * pair                                                         # :56
    o = torch.matmul(att, v).transpose(1, 2).contiguous().view(N, K, d)      # :57,:98
    o = _linear(sd, pfx + ".slf_attn.fc", o) + res                           # :99-100
    o = masked_ln(sd, pfx + ".slf_attn.norm", o, mask)
    o = o.masked_fill(~mask.unsqueeze(-1), 0.0)                              # :39
    res = o
    z = torch.relu(_linear(sd, pfx + ".pos_ffn.w_1", o))                     # :118
    z = z.masked_fill(~mask.unsqueeze(-1), 0.0)
    z = _linear(sd, pfx + ".pos_ffn.w_2", z)
    z = z.masked_fill(~mask.unsqueeze(-1), 0.0)
    z = masked_ln(sd, pfx + ".pos_ffn.norm", z + res, mask)
    return z.masked_fill(~mask.unsqueeze(-1), 0.0)                           # :41


def set_transformer(sd, pfx, x, pos, mask, nlayer, training, trace=None, attn_keep=None):
    """SetTransformer.forward — sign_net.py:60-72 [core/sign_net.py:64-77]."""
    x = x + pos
    for l in range(nlayer):
        x = encoder_layer(sd, f"{pfx}.transformer_layers.{l}", x, mask, None if attn_keep is None else attn_keep[l])
        if trace is not None:
            trace.append(x.clone())
    s = x.sum(dim=1)
    return _bn_rows(sd, pfx + ".out.1", F.linear(s, sd[pfx + ".out.0.weight"]), training), s


# --------------------------------------------------------------------------- a2: SignNet
def sign_net(sd, cfg, data, training=False, max_k=None, out=None, attn_keep=None):
    """SignNet.forward — sign_net.py:96-118 [core/sign_net.py:99-120]."""
    eig_s, eig_v, mask = to_dense_list_evd(data.eigen_values, data.eigen_vectors, data.batch,
                                           getattr(data, "num_graphs", None), max_k)
    x = eig_v.unsqueeze(-1)
    if cfg["variant"] == "alchemy" and not cfg.get("ignore_eigval", False):
        pos = masked_mlp(sd, "sign_net.eigen_encoder", eig_s.unsqueeze(-1), mask, 2, True, training)
    else:
        pos = 0          # core/sign_net.py:112 — eigen_encoder2's value is discarded
    tr_p, tr_m = ([] if out is not None else None), ([] if out is not None else None)
    phi = (gnn3d(sd, "sign_net.phi", x, data.edge_index, mask, cfg["nl_signnet"], training, tr_p)
           + gnn3d(sd, "sign_net.phi", -x, data.edge_index, mask, cfg["nl_signnet"], training, tr_m))
    tr_r = [] if out is not None else None
    pe, ssum = set_transformer(sd, "sign_net.rho", phi, pos, mask, cfg["nl_rho"], training, tr_r, attn_keep)
    if out is not None:
        out.update(eigV_dense=eig_v, eigS_dense=eig_s, mask=mask, phi_plus_layers=tr_p,
                   phi_minus_layers=tr_m, phi=phi, rho_layers=tr_r, rho_sum=ssum, pos=pe)
        if torch.is_tensor(pos):
            out["eig_pos"] = pos
    return pe


# --------------------------------------------------------------------------- a10-a13: GNN
def gnn(sd, cfg, data, pe, training=False, out=None):
    """GNN.forward — model.py:36-64 [core/model.py:44-79] with GINEConv layers
    (pyg_gnn_wrapper.py:19-28) and add pooling."""
    xin = data.x.squeeze()
    if cfg["node_feat"] is None:
        h = discrete_encoder(sd, "gnn.input_encoder", xin)
    else:
        h = plain_mlp(sd, "gnn.input_encoder", xin, 1, True, training)
    h = F.linear(torch.cat([h, pe], dim=-1), sd["gnn.linear.weight"], sd["gnn.linear.bias"])
    ea = data.edge_attr
    prev = h
    layers = []
    for l in range(cfg["nl_gnn"]):
        if cfg["edge_feat"] is None:
            e = discrete_encoder(sd, f"gnn.edge_encoders.{l}", ea)
        else:
            e = plain_mlp(sd, f"gnn.edge_encoders.{l}", ea, 1, True, training)
        u = gine_aggregate(h, data.edge_index, e, _eps(sd, f"gnn.convs.{l}.layer.eps", h))
        u = plain_mlp(sd, f"gnn.convs.{l}.nn", u, 2, False, training)
        u = torch.relu(_bn_rows(sd, f"gnn.norms.{l}", u, training))
        h = u + prev
        prev = h
        layers.append(h.clone())
    B = data.num_graphs
    pooled = torch.zeros(B, h.shape[1], dtype=h.dtype).index_add_(0, data.batch, h)
    y = plain_mlp(sd, "gnn.output_encoder", pooled, 2, False, training)
    if out is not None:
        out.update(gnn_layers=layers, pooled=pooled, y=y)
    return y


def signnet_gnn(sd, cfg, data, training=False, max_k=None, out=None, attn_keep=None):
    """SignNetGNN.forward — sign_net.py:130-132 [core/sign_net.py:132-134].  attn_keep: per encoder layer, the attention dropout's scaled
    keep-mask (train mode; see encoder_layer)."""
    pe = sign_net(sd, cfg, data, training, max_k, out, attn_keep)
    return gnn(sd, cfg, data, pe, training, out)


def make_cfg(variant, node_feat, edge_feat, n_hid, n_out, nl_signnet, nl_gnn, ignore_eigval=False):
    """Constructor arguments -> the few numbers the functional oracle needs.  nl_rho is fixed by
    the reference constructors: 4 (Alchemy sign_net.py:123, argument ignored) / 1 (core/sign_net.py:125)."""
    return dict(variant=variant, node_feat=node_feat, edge_feat=edge_feat, n_hid=n_hid, n_out=n_out,
                nl_signnet=nl_signnet, nl_gnn=nl_gnn, nl_rho=4 if variant == "alchemy" else 1,
                ignore_eigval=ignore_eigval)
