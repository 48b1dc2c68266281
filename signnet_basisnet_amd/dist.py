"""Multi-GPU driver pieces: one process per GPU, the batch sharded by graph, no data-path collective.

The forward of SignNet+GINE never mixes graphs (every aggregation is within a graph, pooling is per
graph, eval-mode BatchNorm uses fixed statistics — SURVEY.md §8(e)), so N ranks simply own disjoint
contiguous ranges of graphs.  torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" on CPU) is used
only for the start/stop barrier and the max-over-ranks step time.  The training variant (BASELINE config 4) adds the
gradient all-reduce of `optim.FlatAdam` (flat fp32 buckets, SUM over RCCL, overlapped with the rest of the backward).
"""
from __future__ import annotations

import os
import types

import torch


def shard_range(num_graphs: int, rank: int, world: int, work=None):
    """Contiguous [lo, hi) range of graph ids owned by `rank`.  Without `work`: balanced graph COUNT.  With `work` (one non-negative
    number per graph, e.g. n_b * min(n_b, k) — the phi / rho rows a graph contributes, n_b^2 in the reference's all-eigenvector
    mode): contiguous ranges whose work sums are as equal as a prefix split allows (boundary r = the first
    graph at which the running sum reaches r/world of the total; ranges stay contiguous so that no collation is needed)."""
    if work is None:
        base, rem = divmod(num_graphs, world)
        lo = rank * base + min(rank, rem)
        return lo, lo + base + (1 if rank < rem else 0)
    w = [float(v) for v in work]
    if len(w) != num_graphs:
        raise ValueError("shard_range: one work value per graph expected")
    total = sum(w)
    if total <= 0:
        return shard_range(num_graphs, rank, world)
    bounds, run, r = [0], 0.0, 1
    for i, v in enumerate(w):
        run += v
        while r < world and run >= total * r / world - 1e-9:
            # graph i closes range r-1 unless leaving it to the next range is closer to the target
            cut = i + 1 if (run - total * r / world) <= (total * r / world - (run - v)) else i
            bounds.append(max(cut, bounds[-1]))
            r += 1
    while len(bounds) < world:
        bounds.append(num_graphs)
    bounds.append(num_graphs)
    if num_graphs >= world:       # never an empty rank when there is a graph for everyone (one heavy graph must not starve a rank)
        for r in range(1, world):
            bounds[r] = min(max(bounds[r], bounds[r - 1] + 1, r), num_graphs - (world - r))
    return bounds[rank], bounds[rank + 1]


def shard_batch(data, rank: int, world: int, balance: str = "count", max_k=None):
    """Slice a collated batch (the duck-typed layout of SURVEY.md §8(b)) down to this rank's graphs.
    Pure indexing on the host; node / edge ids are re-based to the shard.  balance="rows": ranges of equal phi / rho work
    (sum of n_b * min(n_b, max_k), i.e. n_b^2 with all eigenvectors) instead of equal graph count."""
    B = int(data.num_graphs)
    sizes = list(data.sizes) if hasattr(data, "sizes") else torch.bincount(data.batch, minlength=B).tolist()
    if balance == "rows":
        lo, hi = shard_range(B, rank, world, [n * (min(n, int(max_k)) if max_k else n) for n in sizes])
    elif balance == "count":
        lo, hi = shard_range(B, rank, world)
    else:
        raise ValueError("balance must be 'count' or 'rows'")
    return slice_graphs(data, lo, hi, sizes)


def slice_graphs(data, lo: int, hi: int, sizes=None):
    """The graphs [lo, hi) of a collated batch as a batch of their own (node / edge ids re-based): pure indexing."""
    B = int(data.num_graphs)
    if sizes is None:
        sizes = list(data.sizes) if hasattr(data, "sizes") else torch.bincount(data.batch, minlength=B).tolist()
    nstart = sum(sizes[:lo])
    nend = nstart + sum(sizes[lo:hi])
    vstart = sum(s * s for s in sizes[:lo])
    vend = vstart + sum(s * s for s in sizes[lo:hi])
    src = data.edge_index[0]
    emask = (src >= nstart) & (src < nend)
    out = types.SimpleNamespace(
        x=data.x[nstart:nend], edge_index=data.edge_index[:, emask] - nstart, edge_attr=data.edge_attr[emask],
        batch=data.batch[nstart:nend] - lo, eigen_values=data.eigen_values[nstart:nend],
        eigen_vectors=data.eigen_vectors[vstart:vend], num_graphs=hi - lo, num_nodes=nend - nstart)
    out.sizes = sizes[lo:hi]
    return out


def init_process_group(backend: str):
    """Env-driven init (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT as set by torch.distributed.run)."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this driver
    kw = {}
    if backend == "nccl":
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        kw["device_id"] = torch.device("cuda", local)
    dist.init_process_group(backend, **kw)
    return dist


def max_over_ranks(value: float, dist, device) -> float:
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_ranks(value: float, dist, device):
    """Every rank's value, in rank order (a one-element list without a process group)."""
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if dist is None:
        return [float(value)]
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def gather_outputs(y: torch.Tensor, dist):
    """All ranks' [B_r, n_out] outputs concatenated in rank order (graph order)."""
    if dist is None:
        return y
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, y.cpu())
    return torch.cat(parts, 0)
