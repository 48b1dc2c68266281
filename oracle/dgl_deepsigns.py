"""Oracle (test infrastructure): the DGL-tree sign-invariant nets, restated functionally.

Reference: /root/reference/GraphPrediction/layers/deepsigns.py:33-86 (GINDeepSigns,
MaskedGINDeepSigns), layers/gnns.py:81-114 (GIN), layers/mlp.py:5-56 (MLP).
Third-party op restated (dgl is absent and unpinned by the reference):
dgl.nn.pytorch.GINConv(apply_func, 'sum') = apply_func((1+eps) h_i + sum_{j->i} h_j), eps=0 buffer.
The graph is passed as (src, dst, batch_num_nodes).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

BN_EPS = 1e-5


def _bn(sd, pfx, x, training):
    """BatchNorm1d on [N,C] or, for 3-D [N,K,C] input, over N*K rows
    (mlp.py:44-47 / gnns.py:107-110 transpose(2,1) trick)."""
    shp = x.shape
    rows = x.reshape(-1, shp[-1])
    if training:
        y = F.batch_norm(rows, None, None, sd[pfx + ".weight"], sd[pfx + ".bias"], True, 0.0, BN_EPS)
    else:
        y = F.batch_norm(rows, sd[pfx + ".running_mean"], sd[pfx + ".running_var"],
                         sd[pfx + ".weight"], sd[pfx + ".bias"], False, 0.0, BN_EPS)
    return y.reshape(shp)


def _act(name):
    return {"relu": torch.relu, "elu": F.elu, "tanh": torch.tanh}[name]


def mlp(sd, pfx, x, num_layers, use_bn, activation, training):
    """MLP.forward — mlp.py:37-56: Linear -> act -> BN per hidden layer, final Linear (dropout=0)."""
    act = _act(activation)
    for i in range(num_layers - 1):
        x = act(F.linear(x, sd[f"{pfx}.lins.{i}.weight"], sd[f"{pfx}.lins.{i}.bias"]))
        if use_bn:
            x = _bn(sd, f"{pfx}.bns.{i}", x, training)
    i = num_layers - 1
    return F.linear(x, sd[f"{pfx}.lins.{i}.weight"], sd[f"{pfx}.lins.{i}.bias"])


def gin(sd, pfx, src, dst, x, num_layers, use_bn, activation, training):
    """GIN.forward — gnns.py:102-114."""
    for l in range(num_layers):
        if l != 0 and use_bn:
            x = _bn(sd, f"{pfx}.bns.{l - 1}", x, training)
        a = (1 + sd[f"{pfx}.layers.{l}.eps"]) * x + torch.zeros_like(x).index_add_(0, dst, x.index_select(0, src))
        x = mlp(sd, f"{pfx}.layers.{l}.apply_func", a, 2, use_bn, activation, training)
    return x


def gin_deepsigns(sd, src, dst, x, num_layers, k, activation="relu", training=False, out=None):
    """GINDeepSigns.forward — deepsigns.py:45-51.  x [N,k,1] -> [N,k,1]."""
    z = (gin(sd, "enc", src, dst, x, num_layers, True, activation, training)
         + gin(sd, "enc", src, dst, -x, num_layers, True, activation, training))
    if out is not None:
        out["phi"] = z
    y = mlp(sd, "rho", z.reshape(z.shape[0], -1), num_layers, True, activation, training)
    return y.reshape(z.shape[0], k, 1)


def masked_gin_deepsigns(sd, src, dst, batch_num_nodes, x, num_layers, k, activation="relu",
                         training=False, out=None):
    """MaskedGINDeepSigns.forward — deepsigns.py:72-86: mask = arange(K) < n_graph(node)."""
    z = (gin(sd, "enc", src, dst, x, num_layers, True, activation, training)
         + gin(sd, "enc", src, dst, -x, num_layers, True, activation, training))
    n_per_node = torch.repeat_interleave(batch_num_nodes, batch_num_nodes)
    mask = torch.arange(z.shape[1])[None, :] < n_per_node[:, None]
    z = z.masked_fill(~mask.unsqueeze(-1), 0.0)
    s = z.sum(dim=1)
    if out is not None:
        out["phi_sum"] = s
    y = mlp(sd, "rho", s, num_layers, True, activation, training)
    return y.reshape(z.shape[0], k, 1)
