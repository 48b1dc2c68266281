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

    def edges(self, form=None):
        if form == "eid":
            return torch.arange(self.src.numel())
        return self.src, self.dst

    def number_of_edges(self):
        return int(self.src.numel())

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


class _Rows:
    """`edges.src[...]` / `edges.dst[...]` / `edges.data[...]` / `nodes.data[...]` of a user-defined function: a frame indexed by
    the batch's node or edge ids (DGL's EdgeBatch / NodeBatch views, restated from the documented semantics)."""
    def __init__(self, frame, ids):
        self.frame, self.ids = frame, ids

    def __getitem__(self, key):
        return self.frame[key][self.ids]


class _EdgeBatch:
    def __init__(self, g, eids):
        self.src, self.dst, self.data = _Rows(g.ndata, g.src[eids]), _Rows(g.ndata, g.dst[eids]), _Rows(g.edata, eids)


class _NodeBatch:
    def __init__(self, g, nids, mailbox):
        self.data, self.mailbox = _Rows(g.ndata, nids), mailbox


def _apply_edges(self, func, edges=None):
    """Built-in (tuple from dgl.function) or user-defined edge function; `edges`: edge ids the update is restricted to."""
    if isinstance(func, tuple):
        kind, f, out = func
        self.edata[out] = f(self)
        return
    eids = torch.arange(self.number_of_edges()) if edges is None else torch.as_tensor(edges).reshape(-1)
    res = func(_EdgeBatch(self, eids))
    for k, v in res.items():
        if edges is None or k not in self.edata or self.edata[k].shape[1:] != v.shape[1:]:
            full = torch.zeros(self.number_of_edges(), *v.shape[1:], dtype=v.dtype)
            full[eids] = v
            self.edata[k] = full
        else:
            self.edata[k] = self.edata[k].clone()
            self.edata[k][eids] = v


def _messages(self, message_func):
    if isinstance(message_func, tuple):
        _, f, mout = message_func
        return {mout: f(self)}
    return message_func(_EdgeBatch(self, torch.arange(self.number_of_edges())))


def _update_all(self, message_func, reduce_func):
    """Messages along every edge, reduced over each node's in-edges.  Built-in `sum`: index_add in edge order.  A user-defined
    reduce function sees `nodes.mailbox[field]` of shape [nodes, in_degree, ...] — DGL buckets the nodes by in-degree and keeps a
    node's messages in edge-id order; nodes without in-edges keep zeros in the fields the function returns."""
    msgs = _messages(self, message_func)
    if isinstance(reduce_func, tuple):
        _, msg, out = reduce_func
        m = msgs[msg]
        self.ndata[out] = torch.zeros(self.number_of_nodes(), *m.shape[1:], dtype=m.dtype).index_add_(0, self.dst, m)
        return
    N = self.number_of_nodes()
    deg = torch.bincount(self.dst, minlength=N)
    order = torch.argsort(self.dst, stable=True)                 # in-edges of node i, in edge-id order
    start = torch.cumsum(deg, 0) - deg
    outs = {}
    for D in deg.unique().tolist():
        if D == 0:
            continue
        nids = (deg == D).nonzero().reshape(-1)
        eids = order[(start[nids].unsqueeze(1) + torch.arange(D).unsqueeze(0)).reshape(-1)].reshape(nids.numel(), D)
        mailbox = {k: v[eids] for k, v in msgs.items()}
        res = reduce_func(_NodeBatch(self, nids, mailbox))
        for k, v in res.items():
            if k not in outs:
                outs[k] = torch.zeros(N, *v.shape[1:], dtype=v.dtype)
            outs[k][nids] = v
    for k, v in outs.items():
        self.ndata[k] = v


def _send_and_recv(self, edges, message_func, reduce_func):
    """The reference always passes every edge (`g.edges()`), i.e. update_all."""
    if isinstance(edges, tuple):
        assert edges[0].numel() == self.number_of_edges()
    else:
        assert torch.as_tensor(edges).numel() == self.number_of_edges()
    _update_all(self, message_func, reduce_func)


Graph.apply_edges = _apply_edges
Graph.update_all = _update_all
Graph.send_and_recv = _send_and_recv


def _segments(g):
    return torch.repeat_interleave(torch.arange(len(g.batch_num_nodes())), g.batch_num_nodes())


def sum_nodes(g, key):
    x = g.ndata[key]
    return torch.zeros(len(g.batch_num_nodes()), *x.shape[1:], dtype=x.dtype).index_add_(0, _segments(g), x)


def mean_nodes(g, key):
    n = g.batch_num_nodes().to(g.ndata[key].dtype).clamp(min=1)
    return sum_nodes(g, key) / n.view(-1, *([1] * (g.ndata[key].dim() - 1)))


def max_nodes(g, key):
    x = g.ndata[key]
    out = torch.full((len(g.batch_num_nodes()), *x.shape[1:]), float("-inf"), dtype=x.dtype)
    seg = _segments(g)
    return out.scatter_reduce(0, seg.view(-1, *([1] * (x.dim() - 1))).expand_as(x), x, reduce="amax")
