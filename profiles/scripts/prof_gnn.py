"""In-kernel stamps of a -DSN_PROFILE build of fused_gnn.hip for one workgroup (= one graph)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from signnet_basisnet_amd import synth, _lib
W = bench.WORKLOAD
host = synth.make_batch(W["B"], seed=1236, n_lo=W["n_lo"], n_hi=W["n_hi"], features=W["features"])
sizes = list(host.sizes)
dev = torch.device("cuda:0")
data = synth.batch_to(host, dev)
model = bench.build_model(dev); model.strict = False
L = C.CDLL(_lib.LIB_PATH)
order = sorted(range(len(sizes)), key=lambda i: sizes[i])
for blk in (order[-1], order[len(order) // 2], order[0]):
    L.sn_prof_set_block(C.c_int(blk))
    with torch.no_grad():
        for _ in range(60):
            model(data)
    torch.cuda.synchronize()
    buf = (C.c_longlong * 64)()
    L.sn_prof_read_gnn(buf)
    g = list(buf)
    t0 = g[0]
    print(f"block {blk}: n={g[6]} ne={g[7]}  total {g[5]-t0}  inputs {g[1]-t0}  lin_a {g[20]-g[1]}  lin_b+bar {g[2]-g[20]}  "
          f"layers {[g[24+l]-(g[2] if l==0 else g[24+l-1]) for l in range(6)]}  pool {g[4]-g[3]}  head {g[5]-g[4]}")
    print(f"    prologue: loads issued {g[30]-t0}  clear {g[31]-g[30]}  csr+efeat+bar {g[32]-g[31]}  classes+EE table {g[33]-g[32]}  slot sum {g[34]-g[33]}  encoder {g[35]-g[34]}  bar {g[1]-g[35]}")
    print(f"    sums over layers: aggregation+bar {g[9]}  gemm1+bar {g[11]}  gemm2+bar {g[12]}")
