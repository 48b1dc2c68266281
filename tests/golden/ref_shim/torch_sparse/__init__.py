"""Stand-in for torch_sparse.SparseTensor(...).to_dense()."""
import torch


class SparseTensor:
    def __init__(self, row, col, value=None, sparse_sizes=None):
        self.row, self.col, self.value, self.sizes = row, col, value, sparse_sizes

    def to_dense(self):
        v = self.value if self.value is not None else torch.ones(self.row.numel())
        out = torch.zeros(self.sizes, dtype=v.dtype)
        out.index_put_((self.row, self.col), v, accumulate=True)
        return out
