"""Eigendecomposition pre-transform with the reference's wire format (host side, per sample).

Mirrors Alchemy/sign_net/transform.py:7-23 (EVDTransform / EVD_Laplacian) without torch_geometric/torch_sparse:
undirected closure of edge_index, dense Laplacian (norm=None: D - A; 'sym': I - D^-1/2 A D^-1/2),
torch.linalg.eigh (ascending) -> data.eigen_values [n], data.eigen_vectors [n*n] row-major V[node, eig].
This is the step *before* the hot path; it runs on the CPU in DataLoader workers exactly as
the reference's does.
"""
from __future__ import annotations

import torch


def evd_laplacian(edge_index, num_nodes, norm=None):
    n = int(num_nodes)
    A = torch.zeros(n, n, dtype=torch.float32)
    if edge_index.numel():
        s, d = edge_index[0].long(), edge_index[1].long()
        keep = s != d
        A[s[keep], d[keep]] = 1.0
        A[d[keep], s[keep]] = 1.0                      # to_undirected
    deg = A.sum(1)
    if norm is None:
        L = torch.diag(deg) - A
    elif norm == "sym":
        dis = deg.pow(-0.5)
        dis[torch.isinf(dis)] = 0.0
        L = torch.eye(n) - dis[:, None] * A * dis[None, :]
    else:
        raise ValueError(f"unsupported normalization {norm!r}")
    return torch.linalg.eigh(L)


class EVDTransform:
    def __init__(self, norm=None):
        self.norm = norm

    def __call__(self, data):
        n = data.num_nodes if getattr(data, "num_nodes", None) is not None else int(data.x.shape[0])
        D, V = evd_laplacian(data.edge_index, n, self.norm)
        data.eigen_values = D
        data.eigen_vectors = V.reshape(-1)
        return data


# ---------------------------------------------------------------------------------------------------------------
# Device-side, whole-batch variant: one call per collated batch instead of one eigh per sample in
# the DataLoader workers.  All arithmetic is sn_laplacian_evd_f32 (csrc/evd.hip); torch only allocates.

def _graph_ptr(batch, num_graphs, ptr=None):
    import torch
    if ptr is not None:
        return ptr.to(torch.int32)
    counts = torch.bincount(batch, minlength=int(num_graphs))
    gp = torch.zeros(int(num_graphs) + 1, dtype=torch.int64, device=batch.device)
    gp[1:] = counts.cumsum(0)
    return gp.to(torch.int32)


def _dense_eigh_on_device(edge_index, n0, n, norm):
    """Graphs beyond the register kernel's 64 nodes: dense Laplacian + torch.linalg.eigh on the device (library call)."""
    sel = (edge_index[0] >= n0) & (edge_index[0] < n0 + n) & (edge_index[0] != edge_index[1])
    s, d = edge_index[0][sel] - n0, edge_index[1][sel] - n0
    A = torch.zeros(n, n, dtype=torch.float32, device=edge_index.device)
    A[s, d] = 1.0
    A[d, s] = 1.0
    deg = A.sum(1)
    if norm is None:
        L = torch.diag(deg) - A
    else:
        dis = torch.where(deg > 0, deg.rsqrt(), torch.zeros_like(deg))
        L = torch.eye(n, device=A.device) - dis[:, None] * A * dis[None, :]
    return torch.linalg.eigh(L)


def evd_laplacian_batch(edge_index, batch=None, num_graphs=None, norm=None, ptr=None, sizes=None, pos_enc_dim=0, skip=1,
                        check=True):
    """Eigendecomposition of every graph Laplacian of a collated batch, on the device.

    edge_index [2,E] int64 (device), and either `ptr` [B+1] (PyG Batch.ptr) or `batch` [N] ascending + num_graphs.
    `sizes` (host list of node counts) avoids the one host read of the graph sizes.
    Returns (eigen_values [N], eigen_vectors [sum n_b^2], pos_enc [N,k] | None) in the reference's wire format
    (transform.py:12-15: ascending eigenvalues, V.reshape(-1) row-major per graph, blocks concatenated)."""
    from . import ops
    gp = _graph_ptr(batch, num_graphs, ptr)
    if sizes is None:
        host_ptr = gp.tolist()
        sizes = [b - a for a, b in zip(host_ptr[:-1], host_ptr[1:])]
    sizes = [int(v) for v in sizes]
    N, total = sum(sizes), sum(v * v for v in sizes)
    val, vec, evoff, pe, status = ops.laplacian_evd(edge_index, gp, N, total, norm, pos_enc_dim, skip)
    if check:
        st = int(status[0].item())
        if st & ~2:
            raise RuntimeError("sn_laplacian_evd_f32: " + "; ".join(m for b, m in ops.EVD_STATUS.items() if st & b & ~2))
        if st & 2:
            n0 = 0
            off = 0
            for n in sizes:
                if n > 64:
                    D, V = _dense_eigh_on_device(edge_index, n0, n, norm)
                    val[n0:n0 + n] = D
                    vec[off:off + n * n] = V.reshape(-1)
                    if pe is not None:
                        kk = max(0, min(pos_enc_dim, n - skip))
                        pe[n0:n0 + n, :kk] = V[:, skip:skip + kk]
                n0 += n
                off += n * n
    return val, vec, pe


class BatchEVDTransform:
    """`EVDTransform` for a collated batch on the device: fills data.eigen_values / data.eigen_vectors exactly as a
    DataLoader over per-sample `EVDTransform(norm)` outputs would have collated them (transform.py:11-15)."""

    def __init__(self, norm=None):
        if norm not in (None, "sym"):
            raise ValueError(f"unsupported normalization {norm!r}")
        self.norm = norm

    def __call__(self, data):
        B = getattr(data, "num_graphs", None)
        if B is None:
            B = int(data.batch.max()) + 1
        D, V, _ = evd_laplacian_batch(data.edge_index, data.batch, B, self.norm, ptr=getattr(data, "ptr", None),
                                      sizes=getattr(data, "sizes", None))
        data.eigen_values = D
        data.eigen_vectors = V
        return data


def lap_positional_encoding_batch(edge_index, batch=None, num_graphs=None, pos_enc_dim=8, ptr=None, sizes=None):
    """DGL-tree positional encoding for a batch (data/molecules.py:148-181, tau = 0): eigenvectors 1..k of
    I - D^-1/2 A D^-1/2 (ascending), zero padded to k columns when n <= k.  Returns pos_enc [N, k]."""
    _, _, pe = evd_laplacian_batch(edge_index, batch, num_graphs, "sym", ptr=ptr, sizes=sizes, pos_enc_dim=pos_enc_dim, skip=1)
    return pe
