"""Phase profile of the training link kernels (csrc/train.hip built with -DSN_PROFILE by prof_train.sh): a few training steps, then the
cycle sums of workgroup 0 / wave 0 per phase."""
import ctypes as C, sys, os, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from signnet_basisnet_amd import _lib
sys.argv = ["bench.py", "--workload", "train", "--steps", "5", "--warmup", "3", "--no-cpu-baseline"]
try:
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
except SystemExit:
    pass
torch.cuda.synchronize()
buf = (C.c_longlong * 64)()
C.CDLL(_lib.LIB_PATH).sn_prof_read_train(buf)
g = list(buf)
names = ["prologue", "phase1 (stage dz, x)", "barrier1", "phase2 (dX)", "phase3 (dW)", "barrier2", "epilogue"]
for o, nm in ((0, "bwd"), (10, "bwd + dot_x")):
    print(nm, "rounds", g[o + 8], "R", g[o + 9], {n: g[o + i] for i, n in enumerate(names)}, "total", sum(g[o:o + 7]))
names = ["prologue", "bn/relu (+ load wait)", "split", "epilogues", "final statistics", "MFMA blocks"]
for o, nm in ((20, "fwd + statistics"), (30, "fwd")):
    print(nm, "tiles of the workgroup", g[o + 8], "of wave 0", g[o + 7], "R", g[o + 9], {n: g[o + i] for i, n in enumerate(names)}, "total", sum(g[o:o + 6]))
print("fwd + statistics: cycles from the loop start to each wave's exit (workgroup 0):", g[40:48])
