"""Synthetic graph batches shaped like the reference's datasets.

Host-side only (numpy/torch CPU): molecule-like graphs = random recursive tree plus
round(n/8) ring-closing chords, symmetrised; eigendecomposition of the sym-normalised
Laplacian I - D^-1/2 A D^-1/2 per graph, stored in the reference's wire format
(`eigen_values [N]`, `eigen_vectors [sum n_b^2]` row-major V[node, eig]; reference:
Alchemy/sign_net/transform.py:11-15) or the DGL format (`pos_enc [N,k]` = eigvecs 1..k,
zero padded; reference: GraphPrediction/data/molecules.py:159-177).
"""
from __future__ import annotations

import types

import numpy as np
import torch


def _random_molecule(rng: np.random.Generator, n: int):
    """Undirected edge set of a random tree on n nodes + round(n/8) chords."""
    und = set()
    for i in range(1, n):
        p = int(rng.integers(0, i))
        und.add((p, i))
    want = int(round(n / 8))
    tries = 0
    while want > 0 and tries < 64 and n > 2:
        a, b = (int(v) for v in rng.integers(0, n, size=2))
        tries += 1
        if a == b:
            continue
        e = (min(a, b), max(a, b))
        if e in und:
            continue
        und.add(e)
        want -= 1
    src = [a for a, b in und] + [b for a, b in und]
    dst = [b for a, b in und] + [a for a, b in und]
    ei = np.array([src, dst], dtype=np.int64)
    order = np.lexsort((ei[1], ei[0]))  # PyG COO order: sorted by source, then target
    return ei[:, order]


def sym_laplacian_eigh(ei: np.ndarray, n: int):
    """fp32 dense sym-normalised Laplacian -> torch.linalg.eigh (ascending), as the
    reference's EVD_Laplacian does (Alchemy/sign_net/transform.py:17-23)."""
    A = np.zeros((n, n), dtype=np.float32)
    A[ei[0], ei[1]] = 1.0
    deg = A.sum(1)
    dis = np.where(deg > 0, 1.0 / np.sqrt(np.maximum(deg, 1e-30)), 0.0).astype(np.float32)
    L = np.eye(n, dtype=np.float32) - dis[:, None] * A * dis[None, :]
    D, V = torch.linalg.eigh(torch.from_numpy(L))
    return D.contiguous(), V.contiguous()


def make_batch(num_graphs: int, seed: int = 1234, n_lo: int = 9, n_hi: int = 37,
               features: str = "zinc", sizes=None):
    """Build one collated batch (CPU tensors) in the PyG duck-typed layout of SURVEY §8(b).

    features: "zinc"  -> x [N,1] int64 in [0,28), edge_attr [E] int64 in [1,4)
              "alchemy" -> x [N,6] f32 U[0,1), edge_attr [E,4] f32 U[0,1)
    """
    rng = np.random.default_rng(seed)
    if sizes is None:
        sizes = rng.integers(n_lo, n_hi + 1, size=num_graphs)
    sizes = [int(s) for s in sizes]
    eis, evals, evecs, batch = [], [], [], []
    off = 0
    for b, n in enumerate(sizes):
        ei = _random_molecule(rng, n)
        D, V = sym_laplacian_eigh(ei, n)
        eis.append(ei + off)
        evals.append(D)
        evecs.append(V.reshape(-1))
        batch.append(np.full(n, b, dtype=np.int64))
        off += n
    # C-contiguous [2, E], as PyG's collate produces it (a strided edge_index costs every forward a device copy)
    edge_index = torch.from_numpy(np.ascontiguousarray(np.concatenate(eis, axis=1))) if eis else torch.zeros(2, 0, dtype=torch.long)
    N, E = off, edge_index.shape[1]
    g = torch.Generator().manual_seed(seed)
    if features == "zinc":
        x = torch.randint(0, 28, (N, 1), generator=g, dtype=torch.long)
        edge_attr = torch.randint(1, 4, (E,), generator=g, dtype=torch.long)
    elif features == "alchemy":
        x = torch.rand(N, 6, generator=g)
        edge_attr = torch.rand(E, 4, generator=g)
    else:
        raise ValueError(features)
    data = types.SimpleNamespace(
        x=x, edge_index=edge_index, edge_attr=edge_attr,
        batch=torch.from_numpy(np.concatenate(batch)) if batch else torch.zeros(0, dtype=torch.long),
        eigen_values=torch.cat(evals) if evals else torch.zeros(0),
        eigen_vectors=torch.cat(evecs) if evecs else torch.zeros(0),
        num_graphs=len(sizes), num_nodes=N,
    )
    data.sizes = sizes
    return data


def batch_to(data, device):
    """`.to(device)` for the SimpleNamespace batches built here."""
    out = types.SimpleNamespace(**vars(data))
    for k, v in vars(data).items():
        if torch.is_tensor(v):
            setattr(out, k, v.to(device))
    return out


def dgl_pos_enc(data, k: int):
    """DGL-layout positional encoding [N,k]: eigenvectors 1..k, zero padded when n <= k
    (GraphPrediction/data/molecules.py:167,176-177)."""
    outs = []
    voff = 0
    for n in data.sizes:
        V = data.eigen_vectors[voff:voff + n * n].view(n, n)
        voff += n * n
        pe = V[:, 1:k + 1]
        if n <= k:
            pe = torch.nn.functional.pad(pe, (0, k - n + 1), value=0.0)
        outs.append(pe)
    return torch.cat(outs, 0).contiguous()


def grid_graph(side: int = 32):
    """4-neighbour side x side grid (the structure of LearningFilters/data/2Dgrid)."""
    idx = np.arange(side * side).reshape(side, side)
    src, dst = [], []
    for a, b in ((idx[:, :-1], idx[:, 1:]), (idx[:-1, :], idx[1:, :])):
        src += [a.ravel(), b.ravel()]
        dst += [b.ravel(), a.ravel()]
    ei = np.stack([np.concatenate(src), np.concatenate(dst)]).astype(np.int64)
    order = np.lexsort((ei[1], ei[0]))
    return ei[:, order], side * side
