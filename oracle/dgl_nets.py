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


# ---------------------------------------------------------------------------------------------------------------
# PNA: /root/reference/GraphPrediction/layers/pna_layer.py:16-160 (PNATower, PNALayer), layers/pna_utils.py:13-89 (aggregators,
# scalers), :245-299 (FCLayer / MLP), nets/ZINC_graph_regression/pna_net.py:19-170 (PNANet, lap_pe / lap_lspe = False, no GRU).
# DGL's apply_edges / update_all with user-defined functions restated: an edge function sees src / dst node rows and edge rows;
# the reduce function sees each node's in-edge messages [D, C] (edge-id order).  Nodes without in-edges get zeros.
# Pinned against tests/golden/dgl_pna_k6.npz (the reference's own modules through the DGL stand-in).
PNA_EPS = 1e-5        # pna_utils.py:9


def _readout(h, batch_num_nodes, readout):
    bnn = torch.as_tensor(batch_num_nodes)
    seg = torch.repeat_interleave(torch.arange(len(bnn)), bnn)
    hg = torch.zeros(len(bnn), h.shape[1], dtype=h.dtype).index_add_(0, seg, h)
    if readout != "sum":
        hg = hg / bnn.to(h.dtype).clamp(min=1).unsqueeze(1)
    return hg


def _mlp_readout(sd, hg):
    y = torch.relu(F.linear(hg, sd["MLP_layer.FC_layers.0.weight"], sd["MLP_layer.FC_layers.0.bias"]))
    y = torch.relu(F.linear(y, sd["MLP_layer.FC_layers.1.weight"], sd["MLP_layer.FC_layers.1.bias"]))
    return F.linear(y, sd["MLP_layer.FC_layers.2.weight"], sd["MLP_layer.FC_layers.2.bias"])


def pna_aggregate(m, dst, N, avg_log):
    """reduce_func_for_h (pna_layer.py:50-56) with aggregators 'mean max min std' and scalers 'identity amplification
    attenuation': m [E, C] messages -> [N, 12*C] = cat_scalers(cat_aggregators)."""
    C = m.shape[1]
    deg = torch.bincount(dst, minlength=N)
    cnt = deg.clamp(min=1).to(m.dtype).unsqueeze(1)
    s1 = torch.zeros(N, C, dtype=m.dtype).index_add_(0, dst, m)
    s2 = torch.zeros(N, C, dtype=m.dtype).index_add_(0, dst, m * m)
    idx = dst.unsqueeze(1).expand(-1, C)
    mx = torch.full((N, C), float("-inf"), dtype=m.dtype).scatter_reduce(0, idx, m, reduce="amax")
    mn = torch.full((N, C), float("inf"), dtype=m.dtype).scatter_reduce(0, idx, m, reduce="amin")
    mean = s1 / cnt
    std = torch.sqrt(torch.relu(s2 / cnt - mean * mean) + PNA_EPS)               # aggregate_std / aggregate_var (:28-36)
    agg = torch.cat([mean, mx, mn, std], dim=1)
    logd = torch.log(deg.to(m.dtype) + 1).unsqueeze(1)                          # np.log(D + 1)            (:74-81)
    out = torch.cat([agg, agg * (logd / avg_log), agg * (avg_log / logd)], dim=1)
    return torch.where((deg > 0).unsqueeze(1), out, torch.zeros_like(out))       # DGL leaves nodes without messages at zero


def pna_layer(sd, pfx, src, dst, h, e, snorm_n, towers, avg_log, residual=True, training=False):
    """PNALayer.forward (pna_layer.py:139-160) with divide_input, edge features, graph_norm and batch_norm."""
    N, d = h.shape
    it = d // towers
    outs = []
    for t in range(towers):
        ht = h[:, t * it:(t + 1) * it]
        tp = f"{pfx}.towers.{t}"
        z = torch.cat([ht[src], ht[dst], e], dim=1)                                                          # pretrans_edges (:38-44)
        m = F.linear(z, sd[f"{tp}.pretrans_h.fully_connected.0.linear.weight"], sd[f"{tp}.pretrans_h.fully_connected.0.linear.bias"])
        a = pna_aggregate(m, dst, N, avg_log)
        x = torch.cat([ht, a], dim=1)                                                                        # (:69)
        x = F.linear(x, sd[f"{tp}.posttrans_h.fully_connected.0.linear.weight"], sd[f"{tp}.posttrans_h.fully_connected.0.linear.bias"])
        x = x * snorm_n                                                                                       # graph_norm (:75-76)
        x = _bn_rows(sd, f"{tp}.batchnorm_h", x, training)
        outs.append(x)
    hc = torch.cat(outs, dim=1)
    ho = F.leaky_relu(F.linear(hc, sd[f"{pfx}.mixing_network_h.linear.weight"], sd[f"{pfx}.mixing_network_h.linear.bias"]))   # FCLayer 'LeakyReLU'
    return h + ho if residual else ho


def pna_net(sd, src, dst, batch_num_nodes, h_idx, p, e_idx, snorm_n, n_layers, towers, avg_log, readout="sum", training=False, out=None):
    """PNANet.forward (pna_net.py:117-170) for pe_init='lap_pe', lap_lspe=False, edge_feat=True, gru=False."""
    h = sd["embedding_h.weight"][h_idx] + F.linear(p, sd["embedding_p.weight"], sd["embedding_p.bias"])
    e = sd["embedding_e.weight"][e_idx]
    for l in range(n_layers):
        h = pna_layer(sd, f"layers.{l}", src, dst, h, e, snorm_n, towers, avg_log, True, training)
    if out is not None:
        out["h_last"] = h
    return _mlp_readout(sd, _readout(h, batch_num_nodes, readout))


# ---------------------------------------------------------------------------------------------------------------
# Sparse graph Transformer: /root/reference/GraphPrediction/layers/transformer.py:112-317 (MultiHeadAttentionLayer with edge
# features, full_graph False; BatchedTransformerLayer with its DEFAULT arguments layer_norm=False, batch_norm=True, residual=True,
# use_bias=False — transformer_net.py:69-70 does not forward net_params['layer_norm']), nets/ZINC_graph_regression/
# transformer_net.py:21-150.  dgl.function src_mul_edge / copy_edge / sum and the edge UDFs restated.
# Pinned against tests/golden/dgl_transformer_{concat_k6,add_k8}.npz.
def graph_attention(sd, pfx, src, dst, h, e, heads):
    """MultiHeadAttentionLayer.forward (transformer.py:197-228): per edge and head score = sum_c K[src] Q[dst] / sqrt(dk) * E[e],
    s = exp(clamp(score, -5, 5)); out[dst] = sum s V[src] / (sum s + 1e-6)."""
    N = h.shape[0]
    Q, K, V = (F.linear(h, sd[f"{pfx}.{n}.weight"]) for n in "QKV")
    Ee = F.linear(e, sd[f"{pfx}.E.weight"])
    dk = Q.shape[1] // heads
    Q, K, V, Ee = (t.view(-1, heads, dk) for t in (Q, K, V, Ee))
    score = (K[src] * Q[dst]) / (dk ** 0.5) * Ee                                    # src_dot_dst, scaling, imp_exp_attn (:159-170)
    s = torch.exp(score.sum(-1, keepdim=True).clamp(-5, 5))                         # exp() (:48-52)
    wV = torch.zeros(N, heads, dk, dtype=h.dtype).index_add_(0, dst, V[src] * s)
    z = torch.zeros(N, heads, 1, dtype=h.dtype).index_add_(0, dst, s)
    return (wV / (z + 1e-6)).reshape(N, heads * dk)


def transformer_layer(sd, pfx, src, dst, h, e, heads, training=False):
    """BatchedTransformerLayer.forward (transformer.py:268-312), defaults: residual, batch_norm, no layer_norm, dropout 0."""
    a = graph_attention(sd, f"{pfx}.attention_h", src, dst, h, e, heads)
    x = h + F.linear(a, sd[f"{pfx}.O_h.weight"], sd[f"{pfx}.O_h.bias"])
    x = _bn_rows(sd, f"{pfx}.batch_norm1_h", x, training)
    y = F.linear(torch.relu(F.linear(x, sd[f"{pfx}.FFN_h_layer1.weight"], sd[f"{pfx}.FFN_h_layer1.bias"])),
                 sd[f"{pfx}.FFN_h_layer2.weight"], sd[f"{pfx}.FFN_h_layer2.bias"])
    return _bn_rows(sd, f"{pfx}.batch_norm2_h", x + y, training)


def transformer_net(sd, src, dst, batch_num_nodes, h_idx, p, e_idx, n_layers, heads, pe_aggregate="concat", readout="sum",
                    training=False, out=None):
    """TransformerNet.forward (transformer_net.py:88-150) for pe_init='lap_pe', lap_lspe=False, edge_feat=True."""
    h = sd["embedding_h.weight"][h_idx]
    pp = F.linear(p, sd["embedding_p.weight"], sd["embedding_p.bias"])
    if pe_aggregate == "concat":
        h = F.linear(torch.cat([h, pp], dim=1), sd["pe_proj.weight"], sd["pe_proj.bias"])
    else:
        h = h + pp
    e = sd["embedding_e.weight"][e_idx]
    for l in range(n_layers):
        h = transformer_layer(sd, f"layers.{l}", src, dst, h, e, heads, training)
    if out is not None:
        out["h_last"] = h
    return _mlp_readout(sd, _readout(h, batch_num_nodes, readout))


# ---------------------------------------------------------------------------------------------------------------
# GAT: /root/reference/GraphPrediction/nets/ZINC_graph_regression/gat_net.py:19-148 (GATNet, lap_pe / lap_lspe = False) on
# dgl.nn.pytorch.GATConv — third-party, absent and unpinned by the reference; restated from its published definition:
# fc without bias, e_ij = leaky_relu(feat_j . attn_l + feat_i . attn_r, 0.2), softmax over a node's in-edges, sum, + bias, ReLU.
def gat_conv(sd, pfx, src, dst, h, heads, slope=0.2):
    N = h.shape[0]
    f = F.linear(h, sd[f"{pfx}.fc.weight"]).view(N, heads, -1)
    el = (f * sd[f"{pfx}.attn_l"]).sum(-1)
    er = (f * sd[f"{pfx}.attn_r"]).sum(-1)
    e = F.leaky_relu(el[src] + er[dst], slope)
    m = torch.full((N, heads), float("-inf"), dtype=e.dtype).scatter_reduce(0, dst.unsqueeze(1).expand(-1, heads), e, reduce="amax")
    w = torch.exp(e - m[dst])
    z = torch.zeros(N, heads, dtype=e.dtype).index_add_(0, dst, w)
    rst = torch.zeros_like(f).index_add_(0, dst, (w / z[dst]).unsqueeze(-1) * f[src])
    return torch.relu(rst + sd[f"{pfx}.bias"].view(1, heads, -1))


def gat_net(sd, src, dst, batch_num_nodes, h_idx, p, n_layers, heads, readout="mean", out=None):
    """GATNet.forward (gat_net.py:89-148) for pe_init='lap_pe', lap_lspe=False: h = embedding_h(h) + embedding_p(p); L-1 GATConv layers
    with the heads flattened, the last one averaged over the heads (:108-110); readout; MLPReadout.  (The edge embedding is computed
    and never used: 'GAT (no edge feature)', :10-13.)"""
    h = sd["embedding_h.weight"][h_idx] + F.linear(p, sd["embedding_p.weight"], sd["embedding_p.bias"])
    for l in range(n_layers - 1):
        h = gat_conv(sd, f"layers.{l}", src, dst, h, heads).flatten(1)
    h = gat_conv(sd, f"layers.{n_layers - 1}", src, dst, h, heads).mean(1)
    if out is not None:
        out["h_last"] = h
    return _mlp_readout(sd, _readout(h, batch_num_nodes, readout))
