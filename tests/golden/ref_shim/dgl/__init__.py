"""Stand-in for dgl (absent, unpinned by the reference): GINConv + a minimal batched graph."""
import torch
from . import nn


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
