"""In-kernel stamps of a -DSN_PROFILE build of plan.hip (k_plan_small): cycles from each workgroup's start."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from signnet_basisnet_amd import synth, _lib, ops
W = bench.WORKLOAD
host = synth.make_batch(W["B"], seed=1236, n_lo=W["n_lo"], n_hi=W["n_hi"], features=W["features"])
dev = torch.device("cuda:0")
d = synth.batch_to(host, dev)
L = C.CDLL(_lib.LIB_PATH)
for kmax in (16, 0):
    for _ in range(30):
        plan = ops.build_plan(d.batch, d.edge_index, d.num_graphs, kmax, bins=True)
    torch.cuda.synchronize()
    buf = (C.c_longlong * 64)()
    L.sn_prof_read_plan(buf)
    g = list(buf)
    b0, b1 = g[0:16], g[16:32]
    print(f"kmax={kmax}: meta {plan.bins.meta.cpu().tolist()}")
    print("  CSR block   : " + "  ".join(f"{n} {b0[i] - b0[0]}" for i, n in ((1, "loads+graph_ptr"), (2, "degrees"), (3, "scan"), (4, "fill"), (5, "rank"), (6, "write-out"))))
    print("  bins block  : " + "  ".join(f"{n} {b1[i] - b1[0]}" for i, n in ((1, "graph_ptr"), (3, "enter"), (4, "hist+prefix"), (9, "column chain"), (8, "slab chain"),
                                                                             (5, "chains+ranks barrier"), (6, "scans"), (7, "records"), (2, "write-out"))))
