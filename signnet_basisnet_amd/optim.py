"""Adam on the device (sn_adam_step_f32), the optimiser the reference's training scripts use
(Alchemy/main_alchemy.py:52 `torch.optim.Adam(model.parameters(), lr=...)`, GINESignNetPyG/core/train.py:32).
Same update as torch.optim.Adam (no amsgrad): weight decay added to the gradient, bias-corrected moments."""
from __future__ import annotations

import torch

from ._lib import check, lib, ptr, stream


class Adam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.params = [p for p in params]
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        self.state = {}
        self.t = 0
        # torch.optim-style view for LR schedulers (ReduceLROnPlateau reads and writes param_groups[i]['lr'])
        self.param_groups = [{"params": self.params, "lr": self.lr}]

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if p.grad is not None:
                if set_to_none:
                    p.grad = None
                else:
                    p.grad.zero_()

    @torch.no_grad()
    def step(self):
        self.t += 1
        lr = float(self.param_groups[0]["lr"])
        for p in self.params:
            if p.grad is None:
                continue
            if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                raise ValueError("Adam: parameters must be contiguous float32 device tensors")
            st = self.state.get(p)
            if st is None:
                st = self.state[p] = (torch.zeros_like(p), torch.zeros_like(p))
            g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
            check(lib().sn_adam_step_f32(ptr(p), ptr(g), ptr(st[0]), ptr(st[1]), p.numel(), lr, self.betas[0], self.betas[1],
                                         self.eps, self.weight_decay, self.t, 1.0, stream()), "sn_adam_step_f32")


_HOOK_PROBE = None


def _hooks_fire_without_a_returned_gradient() -> bool:
    """Do a leaf's post-accumulate-grad hooks run when the adjoint that feeds it returned None (an undefined gradient)?  The stage
    kernels `+=` straight into FlatAdam's flat gradient and hand autograd None (train_stage.direct_grad); the bucketed all-reduce
    learns from those hooks that a bucket is complete.  True on the torch this was written against (2.10); probed once per process
    instead of pinning the version — where it is False, FlatAdam keeps the overlap and drops the direct accumulation."""
    global _HOOK_PROBE
    if _HOOK_PROBE is None:
        fired = []
        if not hasattr(torch.Tensor, "register_post_accumulate_grad_hook"):
            _HOOK_PROBE = False
            return False

        class _NoGrad(torch.autograd.Function):
            @staticmethod
            def forward(ctx, x, w):
                return x * 1.0

            @staticmethod
            def backward(ctx, g):
                return g, None

        with torch.enable_grad():
            w = torch.zeros(1, requires_grad=True)
            x = torch.ones(1, requires_grad=True)
            h = w.register_post_accumulate_grad_hook(lambda p: fired.append(1))
            _NoGrad.apply(x, w).sum().backward()
            h.remove()
        _HOOK_PROBE = bool(fired)
    return _HOOK_PROBE


class FlatAdam:
    """Adam over ONE flat parameter buffer: every parameter's storage and gradient become views into two contiguous fp32
    buffers, so a step is a single kernel launch over the whole model and — with `dist` (torch.distributed, RCCL on ROCm) —
    the data-parallel gradient exchange of BASELINE config 4 is a handful of SUM all-reduces over contiguous BUCKETS of the flat
    gradient (`bucket_mb` each, parameter order), the 1/world averaging folded into the Adam kernel's gradient read.

    Overlap with the backward: autograd finishes the gradients roughly in reverse parameter order — the
    GINE stack and its embedding tables (most of the bytes) long before phi — so a bucket's all-reduce is issued, asynchronously
    on RCCL's stream, by the post-accumulate hook of its LAST gradient while the rest of the backward is still running; `step()`
    waits for all of them.  Which parameters receive a gradient is learned from the first step (the reference registers modules
    its forward never uses: `GNN3d.edge_encoders`, `SetTransformer.pos_encoder`, core/sign_net.py:22,54); a bucket whose set is
    not complete when `step()` is called is reduced there, so the result never depends on the overlap.  xGMI is point-to-point
    (7 links x ~153 GB/s): a ring all-reduce of this model's 27.6 MB is ~0.3 ms per-link bound, hence few, large buckets.
    Build it AFTER `model.to(device)`; use its own `zero_grad()` (the .grad views must stay attached)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, dist=None, bucket_mb=8.0, overlap=True):
        seen, self.params = set(), []
        for p in params:
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                self.params.append(p)
        if not self.params:
            raise ValueError("FlatAdam: no parameters")
        dev = self.params[0].device
        # every parameter starts on a 16-byte boundary of the flat buffers (4-float padding: zero parameters with zero gradients,
        # which Adam leaves at zero): the kernels that read raw parameters can then use vector loads
        total = sum((p.numel() + 3) // 4 * 4 for p in self.params)
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        self.m, self.v = torch.zeros_like(self.flat_p), torch.zeros_like(self.flat_p)
        off = 0
        cap = max(1, int(float(bucket_mb) * (1 << 20) / 4))
        self.buckets, self._bucket_of = [[0, 0]], []          # [lo, hi) float ranges of the flat buffers, parameter order
        with torch.no_grad():
            for p in self.params:
                n = p.numel()
                self.flat_p[off:off + n].copy_(p.detach().reshape(-1))
                p.data = self.flat_p[off:off + n].view(p.shape)
                p.grad = self.flat_g[off:off + n].view(p.shape)
                if self.buckets[-1][1] - self.buckets[-1][0] >= cap:
                    self.buckets.append([off, off])
                self.buckets[-1][1] = off + (n + 3) // 4 * 4
                self._bucket_of.append(len(self.buckets) - 1)
                off += (n + 3) // 4 * 4
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        self.dist, self.t = dist, 0
        self.param_groups = [{"params": self.params, "lr": self.lr}]
        nb = len(self.buckets)
        self._fired = [set() for _ in range(nb)]       # parameters of the bucket whose gradient has arrived in this backward
        self._expect = [None] * nb                     # the set of the previous step (None: not learned yet)
        self._work = [None] * nb                       # in-flight all-reduce of the bucket
        self.early_launches = 0                        # buckets whose all-reduce was issued from inside the backward (last step)
        self._hooks = []
        if dist is not None and overlap and hasattr(self.params[0], "register_post_accumulate_grad_hook"):
            for i, p in enumerate(self.params):
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))
        # .grad is a view of the zeroed flat buffer: `+=` inside an adjoint kernel IS AccumulateGrad.  With the bucket hooks armed that
        # is only safe where autograd still runs a parameter's hooks after an adjoint that returned None for it (probed, see above):
        # otherwise a bucket could be all-reduced before the kernels that add into it have run.
        direct = not self._hooks or _hooks_fire_without_a_returned_gradient()
        for p in self.params:
            p._sn_direct_grad = direct
        self._sync = True

    def disable_overlap(self):
        """Drop the post-accumulate hooks: every bucket's all-reduce is then issued by step() (one after the other, after the backward).
        train_graph.GraphedStep does this — a captured backward cannot launch collectives from hooks."""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        self._expect = [None] * len(self.buckets)
        self._fired = [set() for _ in self.buckets]

    def _make_hook(self, i):
        b = self._bucket_of[i]

        def hook(_param):
            if self._work[b] is not None:
                raise RuntimeError("FlatAdam: a gradient arrived in a bucket whose all-reduce is already in flight — a second backward() "
                                   "before step() (gradient accumulation: run every backward but the last under `with optimizer.no_sync():`), "
                                   "or the set of parameters the forward uses changed between steps (construct with overlap=False). "
                                   "The bucket's gradient has already been summed across ranks in place: discard this step")
            self._fired[b].add(i)
            if self._sync and self._expect[b] is not None and self._fired[b] == self._expect[b]:
                self._launch(b)
                self.early_launches += 1
        return hook

    def no_sync(self):
        """Gradient accumulation (as DistributedDataParallel.no_sync): backward passes inside the context only accumulate into the flat
        gradient; the buckets go out from the hooks of the first backward OUTSIDE it (or from step())."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            prev, self._sync = self._sync, False
            try:
                yield
            finally:
                self._sync = prev
                for f in self._fired:       # the last backward's hooks start from an empty set again
                    f.clear()
        return ctx()

    def _launch(self, b):
        from . import train_stage
        train_stage.flush_deferred()          # weight gradients still waiting as per-workgroup partials are added first
        lo, hi = self.buckets[b]
        self._work[b] = self.dist.all_reduce(self.flat_g[lo:hi], async_op=True)

    def zero_grad(self, set_to_none=False):
        from . import train_stage
        train_stage.flush_deferred()
        self._finish()
        self.flat_g.zero_()

    def _finish(self):
        for b, w in enumerate(self._work):
            if w is not None:
                w.wait()
                self._work[b] = None

    def all_reduce_gradients(self):
        """SUM over the data-parallel ranks (buckets not yet issued by the backward hooks are issued here, then all are waited
        for); returns the scale that turns the sum into the mean.  With a process group the collective always runs, also at
        world size 1 (RCCL init and ncclAllReduce are then exercised on a one-GPU box)."""
        from . import train_stage
        train_stage.flush_deferred()          # (a no-op after a complete loss.backward(): its end-of-backward callback has run)
        if self.dist is None:
            return 1.0
        early = sum(w is not None for w in self._work)
        for b in range(len(self.buckets)):
            if self._work[b] is None:
                self._launch(b)
        self._finish()
        for b in range(len(self.buckets)):
            if self._hooks:
                self._expect[b] = self._fired[b] if self._fired[b] else None
            self._fired[b] = set()
        self.early_launches = early
        return 1.0 / self.dist.get_world_size()

    @torch.no_grad()
    def step(self):
        for p in self.params:
            if p.grad is None or p.grad.data_ptr() < self.flat_g.data_ptr() or \
                    p.grad.data_ptr() >= self.flat_g.data_ptr() + 4 * self.flat_g.numel():
                raise RuntimeError("FlatAdam: a .grad was detached from the flat buffer (use this optimiser's zero_grad())")
        scale = self.all_reduce_gradients()
        self.t += 1
        check(lib().sn_adam_step_f32(ptr(self.flat_p), ptr(self.flat_g), ptr(self.m), ptr(self.v), self.flat_p.numel(),
                                     float(self.param_groups[0]["lr"]), self.betas[0], self.betas[1], self.eps,
                                     self.weight_decay, self.t, scale, stream()), "sn_adam_step_f32")
