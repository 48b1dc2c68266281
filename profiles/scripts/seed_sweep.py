import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, time
import bench
from signnet_basisnet_amd import synth, ops
W = bench.WORKLOAD
dev = torch.device("cuda:0")
model = bench.build_model(dev); model.strict = False
for seed in (1236, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11):
    host = synth.make_batch(W["B"], seed=seed, n_lo=W["n_lo"], n_hi=W["n_hi"], features=W["features"])
    data = synth.batch_to(host, dev)
    with torch.no_grad():
        for _ in range(30):
            model(data)
        torch.cuda.synchronize()
        rec = ops.KernelTimer()
        with rec:
            for _ in range(20):
                model(data)
        s = rec.summary()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            model(data)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 200
    N = host.num_nodes
    rows = sum(n * min(n, 16) for n in host.sizes)
    print(f"seed {seed}: N={N} phi_rows={rows} bins>={-(-rows//64)}  step {dt*1e3:.4f} ms  " + "  ".join(f"{k.replace('sn_','').replace('_f32','')} {v[1]*1e3:.1f}" for k, v in s.items()))
