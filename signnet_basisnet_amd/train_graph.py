"""A training step of fixed shape captured ONCE as a HIP graph and replayed (MI355X: HIP graphs instead of a tracing compiler).

The differentiable forward of `SignNetGNN` + loss + backward is ~350 kernel launches whose order and shapes depend only on the batch
SHAPE (nodes, edges, graphs, eigenvector entries): the kernels read graph sizes, CSR offsets and validity from device memory.  For a
fixed shape — the bench's synthetic batch, a padded / bucketed loader — the step is recorded once (`torch.cuda.graph` over the same
ctypes launches: the C ABI takes the capture stream) and replayed with ONE graph launch + the single Adam launch of
`optim.FlatAdam` (its bias correction depends on the step count, so it stays outside, as in learning_filters.GraphedEpoch).  Same
kernels, same order, same arithmetic as the eager step: bit-identical losses (tests/test_training_gpu.py).

Before constructing a GraphedStep drop every reference to the results of earlier EAGER steps of the same model (`del loss`): a live
loss keeps that step's autograd graph, and with it AccumulateGrad nodes bound to the default stream; the captured backward would then
have to synchronise with the default stream, which a capturing stream must not do.

What the reference does per step (Alchemy/main_alchemy.py:99-110, GINESignNetPyG/core/train.py:55-66): optimizer.zero_grad(),
model(data), L1 loss, loss.backward(), optimizer.step().
"""
from __future__ import annotations

import contextlib

import torch

_FIELDS = ("x", "edge_index", "edge_attr", "batch", "eigen_values", "eigen_vectors")


def l1_loss(y, target):
    return (y - target).abs().mean()


class GraphedStep:
    def __init__(self, model, optimizer, data, target, loss_fn=l1_loss, warmup=2):
        from .optim import FlatAdam
        if not isinstance(optimizer, FlatAdam):
            raise TypeError("GraphedStep needs optim.FlatAdam (static flat parameter / gradient buffers)")
        if optimizer.dist is not None:
            # data parallel: the captured forward + backward is replayed per rank, then optimizer.step() all-reduces the flat gradient
            # (every bucket, after the replay — the hooks that issue buckets from inside an eager backward cannot run in a capture) and
            # launches Adam.  3.4 ms + one 27.6 MB all-reduce per step against the eager step's ~4.4 ms of host time.
            optimizer.disable_overlap()
        if not getattr(model, "max_k", None):
            raise ValueError("GraphedStep: the number of eigenvector slots must be fixed (max_k): the all-eigenvector mode sizes tensors from the batch")
        from . import train_stage
        train_stage.flush_deferred()          # (nothing of an earlier eager backward may be left for the captured one to reduce)
        self.model, self.optimizer, self.loss_fn = model, optimizer, loss_fn
        self.data, self.target = data, target
        self.shapes = {f: tuple(getattr(data, f).shape) for f in _FIELDS}
        model.train()
        # the attention dropout (transformer_module.py:46,55; the one dropout the reference leaves active) is random per step: its masks
        # are drawn OUTSIDE the graph with torch's device generator — the same draws in the same order as the eager step — into static
        # buffers the captured kernels read
        self._masks = None
        if getattr(model, "attn_dropout", 0.0):
            from . import ops
            from .pyg import N_HEAD
            N, K = int(data.batch.numel()), int(model.max_k)
            self._draw = lambda: [ops.attention_dropout_mask(N, K, N_HEAD, model.attn_dropout, data.batch.device)
                                  for _ in model.sign_net.rho.transformer_layers]
            st0 = torch.cuda.get_rng_state(data.batch.device)
            self._masks = [torch.empty_like(m) for m in self._draw()]
            torch.cuda.set_rng_state(st0, data.batch.device)       # (the sizing draw does not count)
        self._status = None

        def fwd_bwd():
            optimizer.flat_g.zero_()
            y = model(self.data)
            loss = loss_fn(y, self.target)
            loss.backward()
            return loss, y

        # warm-up on a side stream (lazy one-time setup: LDS limits, allocator pools) without touching the model's state
        saved = [b.detach().clone() for b in model.buffers()]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        rng = torch.cuda.get_rng_state(data.batch.device) if self._masks is not None else None
        with self._scoped():
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    self._refresh_masks()
                    fwd_bwd()
            torch.cuda.current_stream().wait_stream(side)
            model.check_train()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                loss, y = fwd_bwd()
            # the captured forward's status words (static memory of the graph): every replay rewrites them, check() reads them
            self._status, model._train_status = getattr(model, "_train_status", None), None
        with torch.no_grad():
            for b, sv in zip(model.buffers(), saved):
                b.copy_(sv)
        if rng is not None:
            torch.cuda.set_rng_state(rng, data.batch.device)      # the warm-up's draws do not count: the first step draws what an eager one would
        self.loss, self.y = loss.detach(), y.detach()

    @contextlib.contextmanager
    def _scoped(self):
        """The two switches the captured step needs on the MODEL — no host read of the status words inside the forward, attention
        dropout masks read from this object's static buffers — are set only while this object runs the model (warm-up and capture;
        a replay does not call the model at all) and restored afterwards: an eager train-mode forward of the same model (a ragged last
        batch, a second loop) draws fresh masks and checks its embedding indices as if no GraphedStep existed."""
        m = self.model
        saved = (getattr(m, "_defer_status", False), getattr(m, "_attn_masks", None))
        m._defer_status, m._attn_masks = True, self._masks
        try:
            yield
        finally:
            m._defer_status, m._attn_masks = saved

    def check(self):
        """Reads the status words of the LAST replayed step (one host wait): raises IndexError for a discrete feature outside its
        embedding table, as nn.Embedding does in the reference's eager step.  step() calls it every `check_every` steps."""
        if self._status is not None and int(self._status[5]):
            from . import ops
            raise IndexError(ops.EMBEDDING_INDEX_ERROR)

    def _refresh_masks(self):
        if self._masks is not None:
            for dst, src in zip(self._masks, self._draw()):
                dst.copy_(src)

    def load(self, data=None, target=None):
        """Copy a new batch of the SAME shape into the static input buffers."""
        if data is not None:
            for f in _FIELDS:
                src, dst = getattr(data, f), getattr(self.data, f)
                if tuple(src.shape) != self.shapes[f]:
                    raise ValueError(f"GraphedStep: {f} has shape {tuple(src.shape)}, the captured step {self.shapes[f]}")
                dst.copy_(src, non_blocking=True)
            if int(data.num_graphs) != int(self.data.num_graphs):
                raise ValueError("GraphedStep: number of graphs differs from the captured step")
        if target is not None:
            self.target.copy_(target, non_blocking=True)

    check_every = 64     # replays between two reads of the status words (0 = only when the caller calls check())

    def step(self, data=None, target=None):
        self.load(data, target)
        self._refresh_masks()
        self.graph.replay()
        self.optimizer.step()
        self._nsteps = getattr(self, "_nsteps", 0) + 1
        if self.check_every and self._nsteps % self.check_every == 0:
            self.check()
        return self.loss


class GraphedForward:
    """A no-grad forward of FIXED shape captured once as a HIP graph: `fn` is a closure over static device tensors (BasisNet on the one
    grid graph of LearningFilters, a serving loop with a padded batch); `replay()` re-runs its launches with one graph launch and
    returns the same output tensors, refreshed in place.  Copy new inputs into the tensors `fn` closes over before replaying."""

    def __init__(self, fn, warmup=2):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.out = fn()

    def replay(self):
        self.graph.replay()
        return self.out
