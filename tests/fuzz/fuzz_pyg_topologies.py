"""Not collected by pytest — a fuzzing tool (python tests/fuzz/fuzz_pyg_topologies.py [iterations], needs an MI355X): random batches of 1-9
random graphs (1-64 nodes, 0-192 edges: arbitrary directed multigraphs with self loops, symmetric graphs, all edges into / out of a
few hubs) through the eval forward of the three SignNetGNN cases of tests/test_topology_gpu.py, fused stages and layer path, against
the fp32 + float64 oracle.  Last run (end of round 3): 3 x 40 batches, 0 failures.  (Test infrastructure: imports oracle/.)"""
import sys, numpy as np, torch, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')      # run from the repo root
import parity_util as PU
import test_topology_gpu as T
from oracle import pyg_signnet as O
from signnet_basisnet_amd import synth
DEV = "cuda:0"
t0 = time.time()
nfail = 0
for ci, case in enumerate(T.CASES):
    variant, feats, ctor, max_k = case
    model = T._model(variant, ctor, max_k, seed=ci)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    cfg = O.make_cfg(variant, *ctor)
    m = model.to(DEV).eval()
    rng = np.random.default_rng(100 + ci)
    for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
        B = int(rng.integers(1, 10))
        group = []
        for b in range(B):
            n = int(rng.choice([1, 2, 3, 5, 8, 13, 21, 34, 50, 63, 64])) if rng.random() < 0.5 else int(rng.integers(1, 65))
            emax = min(192, max(0, n * n))
            E = int(rng.integers(0, emax + 1)) if rng.random() < 0.7 else int(min(emax, rng.integers(0, 3 * n + 1)))
            mode = rng.integers(0, 4)
            if mode == 0:      # arbitrary directed multigraph with self loops
                e = rng.integers(0, n, size=(E, 2))
            elif mode == 1:    # symmetric
                e = T._sym(rng.integers(0, n, size=(E // 2, 2)))
            elif mode == 2:    # everything into a few hubs
                e = np.stack([rng.integers(0, n, size=E), rng.integers(0, min(n, 3), size=E)], 1)
            else:              # everything out of a few hubs
                e = np.stack([rng.integers(0, min(n, 3), size=E), rng.integers(0, n, size=E)], 1)
            group.append((f"g{b}", n, e.astype(np.int64).reshape(-1, 2)))
        if sum(t[1] for t in group) < 2:
            continue      # (the reference's data.x.squeeze() turns a one-node batch into a 0-d tensor and fails)
        host = T._batch(group, feats, seed=1000 * ci + it)
        ref = O.signnet_gnn(sd, cfg, host, training=False, max_k=max_k)
        ref64 = O.signnet_gnn(PU.to_f64(sd), cfg, PU.data_f64(host), training=False, max_k=max_k)
        dd = synth.batch_to(host, DEV)
        with torch.no_grad():
            y = m(dd)
            yl, _ = m(dd, return_stages=True)
        desc = f"case {ci} it {it}: B={B} sizes={host.sizes} E={host.edge_index.shape[1]}"
        for what, out in (("fused", y), ("layer", yl)):
            try:
                assert torch.isfinite(out).all(), "non-finite"
                PU.close_conditioned(out, ref, ref64, what)
            except AssertionError as ex:
                nfail += 1
                print("FAIL", desc, what, str(ex)[:300], flush=True)
print(f"fuzz done: {nfail} failures, {time.time() - t0:.0f} s", flush=True)
