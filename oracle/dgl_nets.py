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
