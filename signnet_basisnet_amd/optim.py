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
                                         self.eps, self.weight_decay, self.t, stream()), "sn_adam_step_f32")
