"""Per-workgroup wall-clock stamps of a -DSN_TIMELINE build of fused_phi.hip (s_memrealtime, 10 ns ticks)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
import bench
from signnet_basisnet_amd import synth, _lib
W = bench.WORKLOAD
host = synth.make_batch(W["B"], seed=1236, n_lo=W["n_lo"], n_hi=W["n_hi"], features=W["features"])
dev = torch.device("cuda:0")
data = synth.batch_to(host, dev)
model = bench.build_model(dev); model.strict = False
with torch.no_grad():
    for _ in range(400):
        model(data)
torch.cuda.synchronize()
buf = (C.c_longlong * (1024 * 8))()
C.CDLL(_lib.LIB_PATH).sn_timeline_read_phi(buf)
t = np.array(list(buf), dtype=np.int64).reshape(1024, 8)[:256]
t0 = t[:, 0].min()
rel = (t - t0) / 100.0          # us
names = ["entry", "after prologue", "after decode", "bin1 end", "bin2 end", "bin3 end", "bin4 end", "exit"]
for i, n in enumerate(names):
    c = rel[:, i]; c = c[t[:, i] > 0]
    if len(c): print(f"{n:16s} n={len(c):3d} min {c.min():7.2f} med {np.median(c):7.2f} max {c.max():7.2f} us")
d1 = rel[:, 3] - rel[:, 2]; d2 = rel[:, 4] - rel[:, 3]
print("bin1 duration med %.2f  bin2 duration med %.2f us" % (np.median(d1), np.median(d2[t[:, 4] > 0])))
three = t[:, 5] > 0
print("workgroups with 3 bins:", int(three.sum()), " bin3 duration med %.2f" % np.median((rel[:, 5] - rel[:, 4])[three]))
# per column (16 consecutive bins = the 16 eigenvector slots of one column of graphs): median bin duration
if os.environ.get("SN_TL_COLUMNS"):
    starts = rel[:, 2:6]
    for rnd in range(3):
        dur = rel[:, 3 + rnd] - rel[:, 2 + rnd]
        ok = t[:, 3 + rnd] > 0
        cols = {}
        for b in range(256):
            if ok[b]: cols.setdefault((rnd * 256 + b) // 16, []).append(dur[b])
        print("round", rnd, " ".join(f"{c}:{np.median(v):.1f}" for c, v in sorted(cols.items())))
    tot = rel[:, 5].copy(); tot[~(t[:, 5] > 0)] = rel[:, 4][~(t[:, 5] > 0)]
    print("per-WG end (us) by WG index /16 :", " ".join(f"{np.median(tot[16*i:16*i+16]):.1f}" for i in range(16)))
