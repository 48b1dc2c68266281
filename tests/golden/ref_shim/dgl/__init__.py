"""Stand-in for dgl (absent, unpinned by the reference): GINConv + a minimal batched graph."""
import torch
from . import nn
from . import function  # noqa: F401


class Graph:
    """Minimal batched graph: edge list + per-graph node counts."""
    def __init__(self, src, dst, batch_num_nodes):
        self.src, self.dst = src, dst
        self._bnn = torch.as_tensor(batch_num_nodes)

    def batch_num_nodes(self):
        return self._bnn

    def edges(self):
        return self.src, self.dst

    def number_of_nodes(self):
        return int(self._bnn.sum())

    @property
    def ndata(self):
        if not hasattr(self, "_ndata"):
            self._ndata = {}
        return self._ndata

    @property
    def edata(self):
        if not hasattr(self, "_edata"):
            self._edata = {}
        return self._edata


def _apply_edges(self, func):
    kind, f, out = func
    self.edata[out] = f(self)


def _update_all(self, message_func, reduce_func):
    _, f, mout = message_func
    _, msg, out = reduce_func
    assert msg == mout
    m = f(self)
    self.ndata[out] = torch.zeros(self.number_of_nodes(), *m.shape[1:], dtype=m.dtype).index_add_(0, self.dst, m)


Graph.apply_edges = _apply_edges
Graph.update_all = _update_all


def _segments(g):
    return torch.repeat_interleave(torch.arange(len(g.batch_num_nodes())), g.batch_num_nodes())


def sum_nodes(g, key):
    x = g.ndata[key]
    return torch.zeros(len(g.batch_num_nodes()), *x.shape[1:], dtype=x.dtype).index_add_(0, _segments(g), x)


def mean_nodes(g, key):
    n = g.batch_num_nodes().to(g.ndata[key].dtype).clamp(min=1)
    return sum_nodes(g, key) / n.view(-1, *([1] * (g.ndata[key].dim() - 1)))
