"""Round-6 verdict item 7 (mixed batches).  The layer-at-a-time path is a chain of ~100 launches: 0.6 ms for ONE graph, 1.2 ms for 128.
Measured here: the layer path on 1 / 8 / 128 graphs, the fused path on the same batches, and the default (strict) module on 127 molecules +
one 70-node graph — 3.67 ms until round 6 (the whole batch layer by layer, every weight re-packed on the way), 1.63 ms since (only the
oversize graph layer by layer, `SignNetGNN._serve_beyond_limits`): bounded below by fused(127) + layer path(1) = 0.86 ms.
    gpurun -- python profiles/scripts/layer_path_vs_batch.py"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from signnet_basisnet_amd import synth

dev = torch.device("cuda:0")
model = bench.build_model(dev)


def timed(fn, n=50, w=10):
    with torch.no_grad():
        for _ in range(w):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


out = {}
for B in (1, 8, 128):
    d = synth.batch_to(synth.make_batch(B, seed=1236), dev)
    model.use_fused, model._prep, model.strict = False, None, False
    out[f"layer_path_B{B}_ms"] = timed(lambda: model(d))
    model.use_fused, model._prep = True, None
    out[f"fused_B{B}_ms"] = timed(lambda: model(d))
    model.check_last()
sizes = list(synth.make_batch(128, seed=1236).sizes)
sizes[64] = 70
mixed = synth.batch_to(synth.make_batch(128, seed=1236, sizes=sizes), dev)
model.strict, model.use_fused, model._prep = True, True, None
out["strict_mixed_127_plus_one_70_node_graph_ms"] = timed(lambda: model(mixed), n=20, w=5)
out["note"] = ("the mixed batch = the refused whole-batch attempt + two fused runs of the graphs around the oversize one + that graph layer by layer "
               "+ the host-side slicing (per-graph node / in-edge counts read back, boolean edge masks)")
print(json.dumps(out))
