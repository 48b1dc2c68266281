"""GINConv / GINEConv restated from the PyG 2.0.1 documentation:
out_i = nn((1+eps) x_i + sum_{j->i} msg_j), source = edge_index[0], target = edge_index[1],
aggregation along dim -2 (so [K,N,d] inputs aggregate over N)."""
import torch
from . import inits


class MessagePassing(torch.nn.Module):
    def __init__(self, *a, **k):
        super().__init__()


class _GINBase(torch.nn.Module):
    def __init__(self, nn, eps=0.0, train_eps=False, **kw):
        super().__init__()
        self.nn = nn
        self.initial_eps = eps
        if train_eps:
            self.eps = torch.nn.Parameter(torch.Tensor([eps]))
        else:
            self.register_buffer("eps", torch.Tensor([eps]))

    def reset_parameters(self):
        inits.reset(self.nn)
        self.eps.data.fill_(self.initial_eps)


class GINConv(_GINBase):
    def forward(self, x, edge_index):
        src, dst = edge_index[0], edge_index[1]
        out = torch.zeros_like(x).index_add_(-2, dst, x.index_select(-2, src))
        out = out + (1 + self.eps) * x
        return self.nn(out)


class GINEConv(_GINBase):
    def forward(self, x, edge_index, edge_attr):
        src, dst = edge_index[0], edge_index[1]
        msg = (x.index_select(-2, src) + edge_attr).relu()
        out = torch.zeros_like(x).index_add_(-2, dst, msg)
        out = out + (1 + self.eps) * x
        return self.nn(out)


class _Unused(torch.nn.Module):
    def __init__(self, *a, **k):
        super().__init__()


GATConv = GCNConv = _Unused


def global_add_pool(x, batch, size=None):
    size = int(batch.max()) + 1 if size is None else size
    return torch.zeros(size, x.size(1), dtype=x.dtype).index_add_(0, batch, x)

ARMAConv = ChebConv = _Unused
from . import conv  # noqa: E402
