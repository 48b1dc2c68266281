"""dgl.nn.pytorch.GINConv(apply_func, 'sum', init_eps=0, learn_eps=False):
apply_func((1+eps) * h_dst + sum_{src->dst} h_src)."""
import torch
from . import glob


class GINConv(torch.nn.Module):
    def __init__(self, apply_func, aggregator_type, init_eps=0, learn_eps=False):
        super().__init__()
        assert aggregator_type == "sum"
        self.apply_func = apply_func
        if learn_eps:
            self.eps = torch.nn.Parameter(torch.FloatTensor([init_eps]))
        else:
            self.register_buffer("eps", torch.FloatTensor([init_eps]))

    def forward(self, g, feat):
        src, dst = g.edges()
        neigh = torch.zeros_like(feat).index_add_(0, dst, feat.index_select(0, src))
        rst = (1 + self.eps) * feat + neigh
        if self.apply_func is not None:
            rst = self.apply_func(rst)
        return rst


class GATConv(torch.nn.Module):
    """dgl.nn.pytorch.GATConv restated from its published definition (DGL >= 0.7, the `bias` argument included; the reference does
    not pin a DGL version): fc without bias, per-head attention vectors attn_l / attn_r, e_ij = leaky_relu(el_src + er_dst),
    edge_softmax over the in-edges of a node, sum of a_ij * feat_src, + bias, activation.  feat_drop = attn_drop = 0 and no residual
    (what gat_net.py:62-66 constructs); a zero-in-degree node raises as DGL does (allow_zero_in_degree=False)."""

    def __init__(self, in_feats, out_feats, num_heads, feat_drop=0., attn_drop=0., negative_slope=0.2, residual=False, activation=None,
                 allow_zero_in_degree=False, bias=True):
        super().__init__()
        assert feat_drop == 0. and attn_drop == 0. and not residual
        self._num_heads, self._out_feats, self._allow_zero = num_heads, out_feats, allow_zero_in_degree
        self.fc = torch.nn.Linear(in_feats, out_feats * num_heads, bias=False)
        self.attn_l = torch.nn.Parameter(torch.FloatTensor(size=(1, num_heads, out_feats)))
        self.attn_r = torch.nn.Parameter(torch.FloatTensor(size=(1, num_heads, out_feats)))
        self.bias = torch.nn.Parameter(torch.FloatTensor(size=(num_heads * out_feats,))) if bias else None
        self.negative_slope, self.activation = negative_slope, activation
        gain = torch.nn.init.calculate_gain("relu")
        torch.nn.init.xavier_normal_(self.fc.weight, gain=gain)
        torch.nn.init.xavier_normal_(self.attn_l, gain=gain)
        torch.nn.init.xavier_normal_(self.attn_r, gain=gain)
        if self.bias is not None:
            torch.nn.init.constant_(self.bias, 0)

    def forward(self, g, feat):
        src, dst = g.edges()
        N, H, C = feat.shape[0], self._num_heads, self._out_feats
        if not self._allow_zero and (torch.bincount(dst, minlength=N) == 0).any():
            raise RuntimeError("There are 0-in-degree nodes in the graph")
        f = self.fc(feat).view(N, H, C)
        el = (f * self.attn_l).sum(dim=-1)                       # [N, H]
        er = (f * self.attn_r).sum(dim=-1)
        e = torch.nn.functional.leaky_relu(el[src] + er[dst], self.negative_slope)     # [E, H]
        m = torch.full((N, H), float("-inf"), dtype=e.dtype).scatter_reduce(0, dst.unsqueeze(1).expand(-1, H), e, reduce="amax")
        w = torch.exp(e - m[dst])
        z = torch.zeros(N, H, dtype=e.dtype).index_add_(0, dst, w)
        a = w / z[dst]
        rst = torch.zeros(N, H, C, dtype=f.dtype).index_add_(0, dst, a.unsqueeze(-1) * f[src])
        if self.bias is not None:
            rst = rst + self.bias.view(1, H, C)
        if self.activation is not None:
            rst = self.activation(rst)
        return rst


class _Unused(torch.nn.Module):
    def __init__(self, *a, **k):
        super().__init__()


GraphConv = _Unused
