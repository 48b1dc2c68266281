import os, sys, json, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from oracle import evd as OE
from signnet_basisnet_amd import synth, transform as T
dev="cuda:0"
worst={}
for B, kw in ((128, {}), (24, {"sizes": [40 + (7 * i) % 25 for i in range(24)]}), (300, {"n_lo":2,"n_hi":20})):
    for seed in (77, 5):
        data = synth.make_batch(B, seed=seed, **kw)
        dd = synth.batch_to(data, torch.device(dev)); dd.eigen_values = dd.eigen_vectors = None
        out = T.BatchEVDTransform("sym")(dd)
        D, V = out.eigen_values.cpu().numpy(), out.eigen_vectors.cpu().numpy()
        ei = data.edge_index.numpy(); off=o2=0
        for n in data.sizes:
            sel = (ei[0] >= off) & (ei[0] < off + n)
            L = OE.dense_laplacian(ei[:, sel] - off, n, "sym")
            dr, vr = OE.evd_laplacian(ei[:, sel] - off, n, "sym")
            r = OE.compare_decompositions(D[off:off+n], V[o2:o2+n*n].reshape(n,n), dr, vr, L, 4e-6)
            for k,v in r.items(): worst[k]=max(worst.get(k,0), float(v) if k!="ok" else float(not v))
            off+=n; o2+=n*n
print("tol (build-time constant since the sweep)", {k: (f"{v:.2e}") for k,v in worst.items()})
