"""Fused stage kernels against the layer-at-a-time path (itself parity-tested against the CPU oracle) over random batches:
sizes, slot counts, seeds — a wider net than the fixed test batches, for edits of the kernels' decode / start-up code."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from signnet_basisnet_amd import synth

W = bench.WORKLOAD
dev = torch.device("cuda:0")
worst = 0.0
cases = 0
configs = [dict(W, k=k) for k in (16, None, 8, 37)] + [dict(bench.WORKLOADS[0]), dict(bench.WORKLOADS[2]), dict(bench.WORKLOADS[2], k=6)]
for WK in configs:
    k = (WK["name"][:24], WK["k"])
    bench.WORKLOAD = WK
    model = bench.build_model(dev)
    model.strict = False
    for B, lo, hi in ((1, 1, 64), (2, 1, 3), (7, 9, 37), (33, 1, 64), (128, 9, 37), (300, 2, 20), (64, 40, 64)):
        for seed in range(3):
            host = synth.make_batch(B, seed=100 * B + seed, n_lo=lo, n_hi=hi, features=WK["features"])
            data = synth.batch_to(host, dev)
            with torch.no_grad():
                model.use_fused, model._prep = True, None
                y1 = model(data).clone()
                model.check_last()
                model.use_fused, model._prep = False, None
                y0 = model(data).clone()
            assert not torch.equal(y1, y0) or B <= 2, "the two paths returned identical bits: the switch did not take"
            err = float((y1 - y0).abs().max() / y0.abs().max().clamp_min(1e-30))
            worst = max(worst, err)
            cases += 1
            if not (err < 2e-5) or not torch.isfinite(y1).all():
                print(f"MISMATCH k={k} B={B} n in [{lo},{hi}] seed={seed}: rel err {err:.3e}")
print(f"{cases} cases, worst relative error {worst:.3e}")
