"""Oracle (test infrastructure): the DGL tree's GIN base network consuming the sign-invariant positional encoding.

Reference: /root/reference/GraphPrediction/nets/ZINC_graph_regression/gin_net.py:19-139 (GINNet: embedding_h / embedding_p,
`h = h + p` :87-92, L x dgl GINConv(MLP(hidden, hidden, out, 2), 'sum') :58-66,99-100, mean/sum readout :126-133),
layers/mlp_readout_layer.py:9-24 (MLPReadout), train/train_ZINC_graph_regression.py:20-25 (the sign_inv call that produces p).
dgl.nn.pytorch.GINConv and dgl.mean_nodes / sum_nodes are restated (dgl is absent from the image and unpinned by the reference).
Only the lap_pe / sign_inv, lap_lspe = False configuration the shipped GIN_ZINC_LapPE_signinv_GIN.json selects is covered.

Pinned against tests/golden/dgl_ginnet_k6.npz (the reference's own GINNet run through the DGL stand-in, eval and train mode).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import dgl_deepsigns as OD


def gin_net(sd, src, dst, batch_num_nodes, h_idx, p, n_layers, readout="mean", training=False):
    """GINNet.forward (gin_net.py:80-139) for pe_init='lap_pe', lap_lspe=False.  p [N, pos_enc_dim] is the (already sign-invariant)
    positional encoding; returns scores [B, 1]."""
    h = sd["embedding_h.weight"][h_idx] + F.linear(p, sd["embedding_p.weight"], sd["embedding_p.bias"])
    for l in range(n_layers):
        a = (1 + sd[f"layers.{l}.eps"]) * h + torch.zeros_like(h).index_add_(0, dst, h.index_select(0, src))
        h = OD.mlp(sd, f"layers.{l}.apply_func", a, 2, True, "relu", training)
    bnn = torch.as_tensor(batch_num_nodes)
    seg = torch.repeat_interleave(torch.arange(len(bnn)), bnn)
    hg = torch.zeros(len(bnn), h.shape[1], dtype=h.dtype).index_add_(0, seg, h)
    if readout != "sum":
        hg = hg / bnn.to(h.dtype).clamp(min=1).unsqueeze(1)
    y = torch.relu(F.linear(hg, sd["MLP_layer.FC_layers.0.weight"], sd["MLP_layer.FC_layers.0.bias"]))
    y = torch.relu(F.linear(y, sd["MLP_layer.FC_layers.1.weight"], sd["MLP_layer.FC_layers.1.bias"]))
    return F.linear(y, sd["MLP_layer.FC_layers.2.weight"], sd["MLP_layer.FC_layers.2.bias"])


def gin_net_with_sign_inv(sd, src, dst, batch_num_nodes, h_idx, pos_enc, n_layers, sign_inv_layers, k, readout="mean", training=False):
    """handle_lap's sign_inv branch followed by the network (train_ZINC_graph_regression.py:20-25,77-80)."""
    ssd = {kk[len("sign_inv_net."):]: v for kk, v in sd.items() if kk.startswith("sign_inv_net.")}
    p = OD.gin_deepsigns(ssd, src, dst, pos_enc.unsqueeze(-1), sign_inv_layers, k, training=training).squeeze(-1)
    return gin_net(sd, src, dst, batch_num_nodes, h_idx, p, n_layers, readout, training), p


# ---------------------------------------------------------------------------------------------------------------
# GatedGCN: /root/reference/GraphPrediction/layers/gatedgcn_layer.py:12-81 (GatedGCNLayer),
# nets/ZINC_graph_regression/gatedgcn_net.py:18-148 (GatedGCNNet, lap_pe / lap_lspe = False).  dgl.function's
# u_add_v / u_mul_e / copy_e / sum restated: edge field from source / destination node fields, sum over in-edges.
# Pinned against tests/golden/dgl_gatedgcn_{concat_k6,add_k8}.npz (the reference's own modules through the DGL stand-in).
def _bn_rows(sd, pfx, x, training):
    if training:
        return F.batch_norm(x, None, None, sd[pfx + ".weight"], sd[pfx + ".bias"], True, 0.0, OD.BN_EPS)
    return F.batch_norm(x, sd[pfx + ".running_mean"], sd[pfx + ".running_var"], sd[pfx + ".weight"], sd[pfx + ".bias"], False, 0.0,
                        OD.BN_EPS)


def gatedgcn_layer(sd, pfx, src, dst, h, e, batch_norm=True, residual=True, training=False):
    """GatedGCNLayer.forward (gatedgcn_layer.py:36-77, graph_norm False, dropout 0): returns (h, e)."""
    lin = lambda n, x: F.linear(x, sd[f"{pfx}.{n}.weight"], sd[f"{pfx}.{n}.bias"])      # noqa: E731
    Ah, Bh, Dh, Eh, Ce = lin("A", h), lin("B", h), lin("D", h), lin("E", h), lin("C", e)
    e_new = Dh[src] + Eh[dst] + Ce                                                      # u_add_v('Dh','Eh') + Ce   (:51-52)
    sigma = torch.sigmoid(e_new)
    num = torch.zeros_like(Ah).index_add_(0, dst, Bh[src] * sigma)                      # u_mul_e('Bh','sigma') summed (:54)
    den = torch.zeros_like(Ah).index_add_(0, dst, sigma)                                # copy_e('sigma') summed       (:55)
    h_new = Ah + num / (den + 1e-6)                                                     # (:56)
    if batch_norm:
        h_new, e_new = _bn_rows(sd, pfx + ".bn_node_h", h_new, training), _bn_rows(sd, pfx + ".bn_node_e", e_new, training)
    h_new, e_new = torch.relu(h_new), torch.relu(e_new)
    if residual and h_new.shape[1] == h.shape[1]:
        h_new, e_new = h + h_new, e + e_new
    return h_new, e_new


def gatedgcn_net(sd, src, dst, batch_num_nodes, h_idx, p, e_idx, n_layers, pe_aggregate="concat", readout="mean", training=False,
                 out=None):
    """GatedGCNNet.forward (gatedgcn_net.py:84-148) for pe_init='lap_pe', lap_lspe=False, edge_feat=True."""
    h = sd["embedding_h.weight"][h_idx]
    pp = F.linear(p, sd["embedding_p.weight"], sd["embedding_p.bias"])
    if pe_aggregate == "concat":
        h = F.linear(torch.cat([h, pp], dim=1), sd["pe_proj.weight"], sd["pe_proj.bias"])
    else:
        h = h + pp
    e = sd["embedding_e.weight"][e_idx]
    for l in range(n_layers):
        h, e = gatedgcn_layer(sd, f"layers.{l}", src, dst, h, e, True, True, training)
    if out is not None:
        out["h_last"] = h
    bnn = torch.as_tensor(batch_num_nodes)
    seg = torch.repeat_interleave(torch.arange(len(bnn)), bnn)
    hg = torch.zeros(len(bnn), h.shape[1], dtype=h.dtype).index_add_(0, seg, h)
    if readout != "sum":
        hg = hg / bnn.to(h.dtype).clamp(min=1).unsqueeze(1)
    y = torch.relu(F.linear(hg, sd["MLP_layer.FC_layers.0.weight"], sd["MLP_layer.FC_layers.0.bias"]))
    y = torch.relu(F.linear(y, sd["MLP_layer.FC_layers.1.weight"], sd["MLP_layer.FC_layers.1.bias"]))
    return F.linear(y, sd["MLP_layer.FC_layers.2.weight"], sd["MLP_layer.FC_layers.2.bias"])
