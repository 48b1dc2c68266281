"""Oracle (test infrastructure): the eigendecomposition transforms that feed the path, restated with numpy.

Reference: /root/reference/Alchemy/sign_net/transform.py:7-23 and GINESignNetPyG/core/transform.py:7-26
(EVDTransform / EVD_Laplacian: to_undirected -> get_laplacian(normalization) -> dense -> torch.linalg.eigh);
/root/reference/GraphPrediction/data/molecules.py:148-181 (lap_positional_encoding, tau = 0: numpy `eig` of
I - D^-1/2 A D^-1/2 with clipped in-degrees, ascending sort, real part, columns 1..k, zero padding).

Third-party pieces restated from their documented behaviour (torch_geometric==2.0.1, absent from the image):
to_undirected = union with the reversed edges, duplicates coalesced; get_laplacian = self loops removed, degree of the
row index, None -> D - A, 'sym' -> I - D^-1/2 A D^-1/2 with 1/sqrt(0) := 0 and a unit diagonal on every node.

Pinned: `evd_laplacian` against tests/golden/evd_transform.npz, which holds the outputs of the reference's own
EVDTransform (eigenvalues directly; eigenvectors through the projectors onto separated eigenvalue clusters, the only
quantity two eigensolvers agree on).  `lap_positional_encoding` is parity unpinned: molecules.py does not import
here (dgl, networkx absent) and the reference holds no test or fixture for it.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.
"""
from __future__ import annotations

import numpy as np


def dense_laplacian(edge_index, n, norm=None, dtype=np.float32):
    """transform.py:18-20.  edge_index [2,E] integer array with local node ids in [0, n)."""
    ei = np.asarray(edge_index).reshape(2, -1)
    A = np.zeros((n, n), dtype=dtype)
    s, d = ei[0], ei[1]
    keep = s != d                                   # get_laplacian removes self loops
    A[s[keep], d[keep]] = 1                         # to_undirected: both directions, coalesced
    A[d[keep], s[keep]] = 1
    deg = A.sum(1)
    if norm is None:
        return np.diag(deg) - A
    if norm == "sym":
        with np.errstate(divide="ignore"):
            dis = np.where(deg > 0, deg ** dtype(-0.5), 0).astype(dtype)
        return (np.eye(n, dtype=dtype) - dis[:, None] * A * dis[None, :]).astype(dtype)
    raise ValueError(f"unsupported normalization {norm!r}")


def evd_laplacian(edge_index, n, norm=None, dtype=np.float32):
    """transform.py:17-23: ascending eigenvalues D [n] and eigenvectors V [n,n] (columns) of the dense Laplacian."""
    return np.linalg.eigh(dense_laplacian(edge_index, n, norm, dtype))


def evd_batch(edge_index, sizes, norm=None):
    """EVDTransform applied to every graph of a batch + PyG collation: (eigen_values [N], eigen_vectors [sum n^2])."""
    ei = np.asarray(edge_index)
    vals, vecs, off = [], [], 0
    for n in sizes:
        sel = (ei[0] >= off) & (ei[0] < off + n)
        D, V = evd_laplacian(ei[:, sel] - off, n, norm)
        vals.append(D)
        vecs.append(V.reshape(-1))                  # transform.py:14
        off += n
    return np.concatenate(vals), np.concatenate(vecs)


def lap_positional_encoding(edge_index, n, pos_enc_dim):
    """molecules.py:161-178 (tau = 0) for one graph: [n, pos_enc_dim] float32."""
    ei = np.asarray(edge_index).reshape(2, -1)
    A = np.zeros((n, n))
    A[ei[0], ei[1]] = 1.0                           # adjacency_matrix_scipy (directed edges as given)
    indeg = A.sum(0)
    Nm = np.diag(np.clip(indeg, 1, None) ** -0.5)
    L = np.eye(n) - Nm @ A @ Nm
    w, v = np.linalg.eig(L)
    idx = w.argsort()
    v = np.real(v[:, idx])
    pe = v[:, 1:pos_enc_dim + 1].astype(np.float32)
    if n <= pos_enc_dim:
        pe = np.pad(pe, ((0, 0), (0, pos_enc_dim - n + 1)))
    return pe


def clusters(vals, sep):
    """Index ranges [a, b) of ascending eigenvalues split wherever consecutive values differ by more than `sep`."""
    cuts = [0] + [i + 1 for i in range(len(vals) - 1) if vals[i + 1] - vals[i] > sep] + [len(vals)]
    return list(zip(cuts[:-1], cuts[1:]))


def compare_decompositions(D, V, D_ref, V_ref, L, atol):
    """What two correct eigensolvers agree on: eigenvalues, residual, orthogonality, and the projector onto every
    cluster of eigenvalues separated from the rest.  Returns a dict of worst-case errors (floats)."""
    n = len(D_ref)
    D, V, D_ref, V_ref, L = (np.asarray(x, dtype=np.float64) for x in (D, V, D_ref, V_ref, L))
    scale = max(1.0, float(np.abs(D_ref).max()))
    out = {"eigenvalues": float(np.abs(D - D_ref).max()) / scale,
           "residual": float(np.abs(L @ V - V * D[None, :]).max()) / scale,
           "orthogonality": float(np.abs(V.T @ V - np.eye(n)).max()),
           "ascending": float(max(0.0, (D[:-1] - D[1:]).max())) if n > 1 else 0.0,
           "projector": 0.0}
    for a, b in clusters(D_ref, 1e-3):
        P = V[:, a:b] @ V[:, a:b].T
        P_ref = V_ref[:, a:b] @ V_ref[:, a:b].T
        out["projector"] = max(out["projector"], float(np.abs(P - P_ref).max()))
    out["ok"] = (out["eigenvalues"] <= atol and out["residual"] <= atol and out["orthogonality"] <= atol
                 and out["ascending"] <= atol and out["projector"] <= 1e3 * atol)
    return out
