"""A data-loader-like loop: a different batch (different N, E) every step."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from signnet_basisnet_amd import synth
W = bench.WORKLOAD
dev = torch.device("cuda:0")
model = bench.build_model(dev); model.strict = False
batches = [synth.batch_to(synth.make_batch(W["B"], seed=s, n_lo=W["n_lo"], n_hi=W["n_hi"], features=W["features"]), dev) for s in range(1, 17)]
with torch.no_grad():
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ts = []
        for i in range(160):
            t1 = time.perf_counter()
            model(batches[i % 16])
            ts.append(time.perf_counter() - t1)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 160
        ts.sort()
        print(f"rep {rep}: {dt*1e3:.4f} ms per forward over 16 alternating batches; host time per call median {ts[80]*1e6:.0f} us, max {ts[-1]*1e6:.0f} us, p99 {ts[-2]*1e6:.0f} us")
    # one batch repeated, for comparison
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(160):
        model(batches[0])
    torch.cuda.synchronize()
    print(f"same batch: {(time.perf_counter()-t0)/160*1e3:.4f} ms")
    print("allocator:", {k: v for k, v in torch.cuda.memory_stats().items() if k in ("num_alloc_retries", "num_device_alloc", "num_device_free", "reserved_bytes.all.current")})
