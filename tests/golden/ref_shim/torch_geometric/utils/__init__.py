"""degree / to_undirected / get_laplacian restated from PyG docs."""
import torch


def degree(index, num_nodes=None, dtype=None):
    n = int(index.max()) + 1 if num_nodes is None else num_nodes
    out = torch.zeros(n, dtype=dtype or torch.float)
    return out.scatter_add_(0, index, torch.ones(index.numel(), dtype=out.dtype))


def to_undirected(edge_index, num_nodes=None):
    row, col = edge_index
    r = torch.cat([row, col]); c = torch.cat([col, row])
    n = int(max(r.max(), c.max())) + 1 if num_nodes is None else num_nodes
    key = torch.unique(r * n + c)
    return torch.stack([key // n, key % n])


def get_laplacian(edge_index, edge_weight=None, normalization=None, dtype=None, num_nodes=None):
    row, col = edge_index
    keep = row != col
    row, col = row[keep], col[keep]
    n = num_nodes
    w = torch.ones(row.numel(), dtype=dtype or torch.float)
    deg = torch.zeros(n, dtype=w.dtype).scatter_add_(0, row, w)
    loop = torch.arange(n)
    if normalization is None:
        ei = torch.cat([torch.stack([row, col]), torch.stack([loop, loop])], 1)
        return ei, torch.cat([-w, deg])
    if normalization == "sym":
        dis = deg.pow(-0.5)
        dis.masked_fill_(dis == float("inf"), 0)
        w = dis[row] * w * dis[col]
        ei = torch.cat([torch.stack([row, col]), torch.stack([loop, loop])], 1)
        return ei, torch.cat([-w, torch.ones(n, dtype=w.dtype)])
    raise ValueError(normalization)


def add_self_loops(*a, **k):
    raise NotImplementedError("stand-in: not on the SignNet/BasisNet path")
