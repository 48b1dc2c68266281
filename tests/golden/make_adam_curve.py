#!/usr/bin/env python
"""Loss curves of 300 Adam steps of the CPU ORACLE (oracle/pyg_signnet.py under torch.autograd + torch.optim.Adam) in float64 and in
float32, from the `gine_d16` fixture's weights and batch: what tests/test_training_gpu.py compares the HIP training loop's curve with.
Derived from the oracle (not from the reference: its backward does not run under torch >= 2); stored so that
the GPU test does not spend two minutes of CPU time on it.

    python tests/golden/make_adam_curve.py          # rewrites tests/golden/adam_curve_gine_d16.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import golden_util as G  # noqa: E402
from oracle import pyg_signnet as O  # noqa: E402

STEPS, LR, SEED = 300, 1e-3, 3


def oracle_setup(fx, dt):
    sd = {k: (v.clone().to(dt).requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone().to(dt)
              if v.is_floating_point() else v.clone()) for k, v in fx.sd.items()}
    data = G.as_data(fx.inp)
    for a in ("eigen_values", "eigen_vectors"):
        setattr(data, a, getattr(data, a).to(dt))
    return sd, data


def main():
    torch.set_num_threads(1)
    fx = G.load("gine_d16")
    cfg = G.pyg_cfg(fx)
    n_out = int(fx.meta["ctor"][3])
    target = torch.randn(len(fx.inp["sizes"]), n_out, generator=torch.Generator().manual_seed(SEED), dtype=torch.float64)
    curves = {}
    for name, dt in (("f64", torch.float64), ("f32", torch.float32)):
        sd, data = oracle_setup(fx, dt)
        opt = torch.optim.Adam([v for v in sd.values() if torch.is_tensor(v) and v.requires_grad], lr=LR)
        ls = []
        for _ in range(STEPS):
            opt.zero_grad()
            loss = (O.signnet_gnn(sd, cfg, data, training=True) - target.to(dt)).abs().mean()
            loss.backward()
            opt.step()
            ls.append(loss.item())
        curves[name] = np.array(ls, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "adam_curve_gine_d16.npz"), f64=curves["f64"], f32=curves["f32"], target=target.numpy(),
                        meta=np.array([STEPS, SEED], dtype=np.int64), lr=np.array(LR))
    print("adam_curve_gine_d16:", curves["f64"][0], "->", curves["f64"][-1], "| f32", curves["f32"][-1])


if __name__ == "__main__":
    main()
