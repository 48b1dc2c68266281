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
