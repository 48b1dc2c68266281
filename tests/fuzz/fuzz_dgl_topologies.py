"""Not collected by pytest — a fuzzing tool (python tests/fuzz/fuzz_dgl_topologies.py [iterations], needs an MI355X): random batches of
2-13 random graphs (2-37 nodes; directed multigraphs, symmetric graphs, hubs; a self loop on every node for GAT) through the eight
shipped GraphPrediction configurations (sign-invariant net + base net, eval) against the fp32 + float64 oracle.  Last run (end of
round 3): 8 x 5 batches, 0 failures.  (Test infrastructure: imports oracle/.)"""
import sys, numpy as np, torch, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')      # run from the repo root
import test_topology_gpu as T
from test_full_size_parity_gpu import run_shipped_dgl_config
t0 = time.time(); nfail = 0
names = ["gin", "gatedgcn", "gat", "pna", "transformer", "gatedgcn_mask", "pna_mask", "transformer_mask"]
for ni, name in enumerate(names):
    rng = np.random.default_rng(500 + ni)
    for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
        B = int(rng.integers(2, 14))
        group = []
        for b in range(B):
            n = int(rng.integers(2, 38))
            E = int(rng.integers(0, min(170, n * n) + 1)) if rng.random() < 0.6 else int(rng.integers(0, 3 * n + 1))
            mode = rng.integers(0, 3)
            if mode == 0:
                e = rng.integers(0, n, size=(E, 2))
            elif mode == 1:
                e = T._sym(rng.integers(0, n, size=(E // 2, 2)))
            else:
                e = np.stack([rng.integers(0, n, size=E), rng.integers(0, min(n, 3), size=E)], 1)
            e = e.astype(np.int64).reshape(-1, 2)
            if name == "gat":      # DGL's GATConv needs an in-edge on every node
                e = np.concatenate([e, np.array([(i, i) for i in range(n)], dtype=np.int64)], 0)
            group.append((f"g{b}", n, e))
        host = T._batch(group, "zinc", seed=10 * ni + it)
        try:
            run_shipped_dgl_config(name, host, elementwise=False)
        except Exception as ex:
            nfail += 1
            print("FAIL", name, it, f"B={B} sizes={host.sizes} E={host.edge_index.shape[1]}", type(ex).__name__, str(ex)[:300], flush=True)
print(f"dgl fuzz done: {nfail} failures, {time.time() - t0:.0f} s", flush=True)
