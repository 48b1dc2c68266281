"""Which ATen launches does one training step still make, and from where?  (round 6: ~30 launches, 0.18 ms of the 3.03 ms step)
    gpurun -- python profiles/scripts/train_glue.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import ProfilerActivity, profile

import bench
from signnet_basisnet_amd import optim, synth

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = bench.build_model(dev).train()
opt = optim.FlatAdam(model.parameters(), lr=1e-3)
data = synth.batch_to(synth.make_batch(128, seed=1236), dev)
target = torch.randn(128, 1, device=dev)


def step():
    opt.zero_grad()
    loss = (model(data) - target).abs().mean()
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
SKIP = ("aten::empty", "aten::as_strided", "aten::view", "aten::select", "aten::slice", "aten::reshape", "aten::detach", "aten::alias", "aten::expand",
        "aten::t", "aten::transpose", "aten::unsqueeze", "aten::squeeze", "aten::_unsafe_view", "aten::result_type", "aten::to", "aten::item",
        "aten::_local_scalar_dense", "aten::is_nonzero", "aten::resize_", "aten::set_", "aten::narrow", "aten::permute", "aten::contiguous", "aten::stride",
        "aten::size", "aten::lift_fresh", "aten::unbind", "aten::split", "aten::chunk", "aten::empty_like", "aten::empty_strided", "aten::view_as")
from collections import Counter
c = Counter()
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CPU and ev.name.startswith("aten::") and ev.name not in SKIP and not any(
            ch.name.startswith("aten::") and ch.name not in SKIP for ch in ev.cpu_children):
        stack = [s for s in (ev.stack or []) if "signnet_basisnet_amd" in s or "train_glue" in s]
        c[(ev.name, str([tuple(s) for s in (ev.input_shapes or [])][:2]), " <- ".join(stack[:2]))] += 1
for (name, shapes, where), n in sorted(c.items(), key=lambda kv: -kv[1]):
    print(n, name, shapes, "|", where)
