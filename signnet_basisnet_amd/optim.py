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


class FlatAdam:
    """Adam over ONE flat parameter buffer: every parameter's storage and gradient become views into two contiguous fp32
    buffers, so a step is a single kernel launch over the whole model and — with `dist` (torch.distributed, RCCL on ROCm) —
    the data-parallel gradient exchange is a single SUM all-reduce of the flat gradient (the few MB of this model fit one
    xGMI-friendly message; the 1/world averaging is folded into the Adam kernel's gradient read).
    Build it AFTER `model.to(device)`; use its own `zero_grad()` (the .grad views must stay attached)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, dist=None):
        seen, self.params = set(), []
        for p in params:
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                self.params.append(p)
        if not self.params:
            raise ValueError("FlatAdam: no parameters")
        dev = self.params[0].device
        total = sum(p.numel() for p in self.params)
        self.flat_p = torch.empty(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        self.m, self.v = torch.zeros_like(self.flat_p), torch.zeros_like(self.flat_p)
        off = 0
        with torch.no_grad():
            for p in self.params:
                n = p.numel()
                self.flat_p[off:off + n].copy_(p.detach().reshape(-1))
                p.data = self.flat_p[off:off + n].view(p.shape)
                p.grad = self.flat_g[off:off + n].view(p.shape)
                off += n
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        self.dist, self.t = dist, 0
        self.param_groups = [{"params": self.params, "lr": self.lr}]

    def zero_grad(self, set_to_none=False):
        self.flat_g.zero_()

    def all_reduce_gradients(self):
        """SUM over the data-parallel ranks; returns the scale that turns the sum into the mean."""
        if self.dist is None or self.dist.get_world_size() == 1:
            return 1.0
        self.dist.all_reduce(self.flat_g)
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
