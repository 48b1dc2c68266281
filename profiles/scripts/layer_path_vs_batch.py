"""Round-6 verdict item 7 (mixed batches): what would splitting a batch buy?  The layer-at-a-time path is a chain of ~100 launches whose
cost hardly depends on the number of graphs; measured here: the layer path on 1 / 8 / 128 graphs, the fused path on 127 / 128 graphs, and
the default (strict) module on 127 good graphs + one 70-node graph (today: the whole batch on the layer path).
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
out["note"] = ("splitting the mixed batch into (good graphs: fused) + (oversize graph: layer path) would cost fused_B128 + layer_path_B1; "
               "the layer path is launch-bound, so that sum is not below what the whole batch costs on the layer path today")
print(json.dumps(out))
