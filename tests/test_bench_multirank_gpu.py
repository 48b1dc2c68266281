"""-m gpu: bench.py's N > 1 path (one process per rank under torch.distributed.run: barriers, max-over-ranks step time, the gradient
all-reduce probe, rank 0's single JSON line) executed end to end on the one-GPU test box — both ranks on cuda:0 over gloo (RCCL refuses
two ranks on one device; the driver's real launch uses RCCL, one rank per GPU).  The kernels, the sharding by rank and the timing
protocol are the ones the driver's scaling run uses."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, port, world=2):
    env = dict(os.environ, SN_BENCH_BACKEND="gloo", SN_BENCH_SHARE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "4", "--warmup", "2"] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                     # rank 0 prints ONE line
    return json.loads(lines[0])


def test_forward_bench_two_ranks():
    d = _run(["--no-cpu-baseline", "--no-scatter", "--streams", "1"], 29731)
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 2 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 256 and d["config"]["graphs_per_gpu"] == 128
    assert d["value"] > 0 and abs(d["value"] - 256 * 4 / (d["ms_per_step"] * 4e-3)) <= 1e-6 * d["value"]
    assert d["distributed"]["world_size"] == 2 and d["distributed"]["data_path_collectives"] == 0
    assert d["roofline"]["bound"] == "mfma" and 0 < d["roofline"]["frac"] < 1
    # (round 4: `value` is the plain one-stream loop again — where the roofline's HIP events are taken; the module's opt-in overlap
    #  mode and the literal `--warmup W --steps K` reading are labelled extra blocks; `warmup` is the driver's W, `warmup_effective`
    #  the untimed forwards actually issued in front of the timed region)
    assert "overlap_front=False" in d["config"]["module_mode"] and d["sequential"]["value"] == d["value"]
    assert d["overlap_mode"]["value"] > 0 and d["cold"]["value"] > 0 and d["cold"]["untimed_forwards_before"] == 2
    assert d["warmup_effective"] >= d["warmup"] + d["steps"]
    assert d["roofline"]["timed_pass"] == "sequential"
    # an N-rank line is self-evidently N ranks: what the backend reports, every rank's own clock and shard, the all-reduce probe
    g = d["distributed"]
    assert g["ranks_seen_by_backend"] == 2 and len(g["per_rank_ms_per_step"]) == 2 and g["graphs_per_rank"] == [128, 128]
    assert max(g["per_rank_ms_per_step"]) <= d["ms_per_step"] * (1 + 1e-9) and len(g["devices"]) == 2 and g["allreduce_us"] > 0


def test_forward_bench_eight_ranks():
    """The driver's largest scaling point (N = 8, BASELINE configs[3]: 1 024 graphs as 8 x 128), launched exactly as the driver launches
    it, on the one-GPU box: eight ranks share cuda:0 over gloo.  What an 8-GPU node cannot break silently afterwards: eight ranks seen
    by the backend, eight clocks, eight shards of 128 graphs, a whole-job value over all of them, ONE line."""
    d = _run(["--no-cpu-baseline", "--no-scatter", "--streams", "1", "--no-overlap", "--no-extras"], 29741, world=8)
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["config"]["global_batch"] == 1024 and d["config"]["graphs_per_gpu"] == 128
    g = d["distributed"]
    assert g["world_size"] == 8 and g["ranks_seen_by_backend"] == 8 and g["data_path_collectives"] == 0
    assert len(g["per_rank_ms_per_step"]) == 8 and g["graphs_per_rank"] == [128] * 8 and len(g["devices"]) == 8
    assert max(g["per_rank_ms_per_step"]) <= d["ms_per_step"] * (1 + 1e-9)
    assert abs(d["value"] - 1024 * 4 / (d["ms_per_step"] * 4e-3)) <= 1e-6 * d["value"]


def test_train_bench_eight_ranks_allreduces_the_flat_gradient():
    d = _run(["--workload", "train", "--no-cpu-baseline", "--no-graph"], 29743, world=8)
    assert d["n_gpus"] == 8 and d["value"] > 0 and d["distributed"]["world_size"] == 8
    assert d["distributed"]["gradient_allreduce"] is not None


def test_train_bench_two_ranks_allreduces_the_flat_gradient():
    d = _run(["--workload", "train", "--no-cpu-baseline"], 29733)
    assert d["n_gpus"] == 2 and d["value"] > 0
    g = d["distributed"]
    assert g["world_size"] == 2
    # the captured step with a process group: replay per rank, then the flat gradient's all-reduce + Adam
    assert d["graphed"]["value"] > 0 and "all-reduce" in d["graphed"]["note"] and d["eager"]["value"] > 0


def test_plain_python_launch_reexecs_under_torchrun():
    """`python bench.py --gpus 2` (no launcher, no WORLD_SIZE) re-executes itself under torch.distributed.run — the driver's
    scaling run may start it either way.  Same one-GPU arrangement as above (gloo, both ranks on cuda:0)."""
    env = dict(os.environ, SN_BENCH_BACKEND="gloo", SN_BENCH_SHARE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--no-cpu-baseline",
           "--no-scatter", "--streams", "1"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 256 and d["distributed"]["world_size"] == 2
