"""CPU, 2 processes, gloo: the N>1 data-parallel path (graph sharding, no data-path collective, max-over-ranks
timing).  The per-rank compute is the CPU oracle here (there is no GPU in this container); on the GPU box the
same driver code runs the HIP modules over RCCL."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    import golden_util as G
    from oracle import pyg_signnet as O
    from signnet_basisnet_amd import dist as D
    from signnet_basisnet_amd import synth
    dist = D.init_process_group("gloo")
    fx = G.load("gine_d16")
    cfg = G.pyg_cfg(fx)
    data = synth.make_batch(7, seed=21, sizes=[5, 9, 4, 12, 6, 7, 10])
    shard = D.shard_batch(data, rank, world)
    y = O.signnet_gnn(fx.sd, cfg, shard, training=False)
    ally = D.gather_outputs(y, dist)
    tmax = D.max_over_ranks(1.0 + rank, dist, "cpu")
    dist.barrier()
    if rank == 0:
        full = O.signnet_gnn(fx.sd, cfg, data, training=False)
        q.put((torch.allclose(ally, full, rtol=1e-6, atol=1e-6), float((ally - full).abs().max()), tmax))
    dist.destroy_process_group()


def test_two_rank_sharded_forward_equals_single_rank():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    ok, err, tmax = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok, f"sharded forward differs from the unsharded one by {err}"
    assert tmax == 2.0          # max over ranks of (1 + rank)


def _worker8(rank, world, port, q, balance):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    import golden_util as G
    from oracle import pyg_signnet as O
    from signnet_basisnet_amd import dist as D
    from signnet_basisnet_amd import synth
    dist = D.init_process_group("gloo")
    fx = G.load("gine_d16")
    cfg = G.pyg_cfg(fx)
    # 11 graphs over 8 ranks: uneven shards, ranks with a single graph; one graph (24 nodes) heavier than several ranks' whole share
    data = synth.make_batch(11, seed=23, sizes=[5, 9, 4, 24, 6, 7, 10, 3, 8, 12, 5])
    shard = D.shard_batch(data, rank, world, balance=balance, max_k=None)
    y = O.signnet_gnn(fx.sd, cfg, shard, training=False) if shard.num_graphs else torch.zeros(0, fx.sd[[k for k in fx.sd][-1]].shape[0])
    ally = D.gather_outputs(y, dist)
    counts = D.all_ranks(float(shard.num_graphs), dist, "cpu")
    tmax = D.max_over_ranks(1.0 + rank, dist, "cpu")
    dist.barrier()
    if rank == 0:
        full = O.signnet_gnn(fx.sd, cfg, data, training=False)
        q.put((tuple(ally.shape) == tuple(full.shape) and torch.allclose(ally, full, rtol=1e-6, atol=1e-6),
               float((ally - full).abs().max()) if tuple(ally.shape) == tuple(full.shape) else -1.0, tmax, [int(c) for c in counts]))
    dist.destroy_process_group()


def _run8(balance):
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q, balance)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return out


def test_eight_rank_sharded_forward_equals_single_rank():
    """BASELINE configs[3]'s world size on CPU / gloo: 11 graphs over 8 ranks — uneven shards, ranks holding one graph — by graph count
    and by (node, slot) rows; the concatenation of the ranks' outputs is the unsharded forward, the time is the maximum over the ranks."""
    for balance in ("count", "rows"):
        ok, err, tmax, counts = _run8(balance)
        assert ok, f"balance={balance}: sharded forward differs from the unsharded one by {err} (graphs per rank {counts})"
        assert tmax == 8.0 and sum(counts) == 11 and min(counts) >= 1 and len(counts) == 8, counts
        assert 1 in counts and max(counts) >= 2, counts


def test_shard_batch_is_a_partition():
    sys.path.insert(0, ROOT)
    from signnet_basisnet_amd import dist as D
    from signnet_basisnet_amd import synth
    data = synth.make_batch(9, seed=3)
    parts = [D.shard_batch(data, r, 4) for r in range(4)]
    assert sum(p.num_graphs for p in parts) == 9
    assert sum(p.batch.numel() for p in parts) == data.batch.numel()
    assert sum(p.edge_index.shape[1] for p in parts) == data.edge_index.shape[1]
    assert torch.equal(torch.cat([p.eigen_vectors for p in parts]), data.eigen_vectors)
    for p in parts:
        if p.batch.numel():
            assert int(p.batch.min()) == 0 and int(p.edge_index.max()) < p.batch.numel()


def _grad_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    import golden_util as G
    from oracle import pyg_signnet as O
    from signnet_basisnet_amd import dist as D
    from signnet_basisnet_amd import optim, synth
    dist = D.init_process_group("gloo")
    fx = G.load("gine_d16")
    cfg = G.pyg_cfg(fx)
    data = synth.make_batch(6, seed=22, sizes=[5, 9, 4, 12, 6, 7])

    def names_of(fx_):
        return [k for k, v in fx_.sd.items() if v.is_floating_point() and "running" not in k]

    def grads_of(batch, dist_=dist):
        sd = {k: (torch.nn.Parameter(v.clone()) if v.is_floating_point() and "running" not in k else v.clone())
              for k, v in fx.sd.items()}
        names = [k for k, v in sd.items() if isinstance(v, torch.nn.Parameter)]
        # CPU tensors: only the gradient plumbing is used here; small buckets so that this tiny model has several
        opt = optim.FlatAdam([sd[k] for k in names], dist=dist_, bucket_mb=0.004)
        O.signnet_gnn(sd, cfg, batch, training=False).sum().backward()   # eval-mode BN: graphs do not interact
        return opt

    opt = grads_of(D.shard_batch(data, rank, world))
    scale = opt.all_reduce_gradients()                                   # first step: every bucket reduced here (sets learned)
    summed = opt.flat_g.clone()
    # second backward on the SAME optimiser: the hooks issue each bucket's all-reduce from inside the backward
    sd2 = {k: v for k, v in zip(names_of(fx), opt.params)}
    opt.zero_grad()
    O.signnet_gnn({**fx.sd, **sd2}, cfg, D.shard_batch(data, rank, world), training=False).sum().backward()
    early_before_step = sum(w is not None for w in opt._work)
    opt.all_reduce_gradients()
    summed2 = opt.flat_g.clone()
    dist.barrier()
    if rank == 0:
        full = grads_of(data, None)                                      # single-rank gradient of the whole batch, no collective
        err = max(float((summed - full.flat_g).abs().max()), float((summed2 - full.flat_g).abs().max()))
        q.put((err, float(full.flat_g.abs().max()), scale, summed.numel(), len(opt.buckets), early_before_step))
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_equals_full_batch_gradient():
    """Data-parallel training exchange (SURVEY.md §8 f1 / BASELINE config 4): per-rank gradients of the rank's graphs,
    summed by one all-reduce of FlatAdam's flat gradient buffer, equal the gradient of the unsharded batch."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    err, gmax, scale, n, nbuckets, early = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert scale == 0.5 and n > 1000
    assert err <= 1e-4 * gmax, (err, gmax)
    assert nbuckets >= 3 and early >= nbuckets - 1, (nbuckets, early)    # second step: all-reduces issued from inside the backward
