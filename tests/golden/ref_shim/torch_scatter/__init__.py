"""Stand-in for torch_scatter (absent from this image), restating its published semantics
(SURVEY.md A.9). Used ONLY by tests/golden/make_golden.py to import the reference modules
in the build container. Not product code, not shipped to the GPU path."""
import torch


def _expand(index, src, dim):
    if dim < 0:
        dim += src.dim()
    if index.dim() == 1:
        shape = [1] * src.dim()
        shape[dim] = -1
        index = index.view(shape)
    return index.expand_as(src), dim


def scatter(src, index, dim=-1, out=None, dim_size=None, reduce="sum"):
    idx, dim = _expand(index, src, dim)
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() else 0
    shape = list(src.shape)
    shape[dim] = dim_size
    tot = torch.zeros(shape, dtype=src.dtype, device=src.device).scatter_add_(dim, idx, src)
    if reduce in ("add", "sum"):
        return tot
    if reduce == "mean":
        cnt = torch.zeros(shape, dtype=src.dtype, device=src.device).scatter_add_(
            dim, idx, torch.ones_like(src)).clamp_(min=1)
        if src.is_floating_point():
            return tot / cnt
        return torch.div(tot, cnt, rounding_mode="floor")
    raise ValueError(reduce)


def scatter_add(src, index, dim=-1, out=None, dim_size=None):
    return scatter(src, index, dim, out, dim_size, "sum")
