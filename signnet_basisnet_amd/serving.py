"""Throughput mode for independent batches: issue consecutive forwards round-robin on a few HIP streams.

One forward of a 128-graph batch is four dependent kernels; φ and ρ end with a partially filled round of their persistent grids
and the GINE stage is a latency chain on half the CUs, so a single stream leaves ≈ 25 % of the device idle.  Independent batches
(a serving loop, an evaluation pass over a dataset) can overlap: `StreamPipeline` keeps `streams` forwards in flight, which is
what bench.py reports as `pipelined` (5.3-5.7·10⁵ graphs/s against 4.1·10⁵ on one stream).  Outputs come back in submission order.
"""
from __future__ import annotations

import torch


class GraphedDGLForward:
    """sign_inv_net + base network of the DGL tree (GraphPrediction/train/train_ZINC_graph_regression.py:20-25,77-80) in eval mode as
    ONE HIP-graph launch, for batches of a FIXED SHAPE (node, edge and graph counts: a padded or bucketed loader, a serving loop).

    Every net of `dgl_nets` but GatedGCN runs its eval forward layer by layer — 50-130 launches of ~12 us of host time each for a few
    us of device work: GIN 0.8, GAT 0.6, PNA 1.0-1.3, Transformer 1.6 ms per 128 graphs, all host-bound.  The launches depend on the
    batch's SHAPE only (graph sizes, CSR offsets, validity are read from device memory by the kernels; the batch plan — sn_batch_plan —
    is part of the recorded launches), so for a fixed shape they are recorded once (`torch.cuda.graph` over the same ctypes launches)
    and replayed: HIP graphs instead of a tracing compiler.  `__call__` copies a new batch of the same shape into the static input
    buffers, replays, and returns the (static) output tensor — consume or clone it before the next call.

    Host-side decisions are frozen at capture time: the largest graph (<= 64 nodes for the stage kernels) and, for GatedGCN, the
    largest in-edge count are those of the example batch; a later batch beyond the kernels' limits is flagged on the device (NaN rows,
    `net.check_last()` raises) instead of being re-routed to the layer path.

    THE WEIGHTS ARE FROZEN AT CAPTURE TOO.  The recorded launches carry device pointers into the eval-time packed copies of the
    parameters (the nets' `_cache`, the sign-invariant net's `_prep` / `_fused`), which `train()`, `eval()`, `.to()`,
    `load_state_dict()` — anything that drops those caches — would free under the recorded graph, while the embedding tables are read
    in place.  This object therefore keeps the caches of the capture alive and remembers the version counter of every parameter and
    buffer: a replay after the net was switched to another mode, moved, re-loaded or trained in between raises instead of returning
    scores of freed or half-updated weights — build a new GraphedDGLForward then."""

    def __init__(self, net, g, h, pos_enc, e=None, snorm_n=None, warmup=2):
        from .dgl_deepsigns import Graph
        from .train_graph import GraphedForward
        if net.training:
            raise ValueError("GraphedDGLForward records the eval forward: call net.eval() first")
        dev = h.device
        src, dst = g.edges()
        bnn = torch.as_tensor(g.batch_num_nodes())
        bne = g.batch_num_edges() if callable(getattr(g, "batch_num_edges", None)) else None
        self.net = net
        self.src, self.dst = src.to(dev).clone(), dst.to(dev).clone()
        self.bnn = bnn.to(dev).clone()                     # on the device: the recorded repeat_interleave reads it at every replay
        self.g = Graph(self.src, self.dst, self.bnn, bne)
        # the two host-side facts the modules read from a graph object, taken from the example batch once (no device read later)
        self.g._sn_node_counts = (int(bnn.max()) if bnn.numel() else 0, int(bnn.sum()))
        self.h, self.pe = h.clone(), pos_enc.clone()
        self.e = None if e is None else e.clone()
        self.snorm = None if snorm_n is None else snorm_n.clone()
        self.shapes = {"src": tuple(self.src.shape), "h": tuple(self.h.shape), "pos_enc": tuple(self.pe.shape), "graphs": int(self.bnn.numel()),
                       "e": None if e is None else tuple(e.shape)}

        def fwd():
            self.g._sn_plans = {}                          # the batch plan is rebuilt inside the recorded step
            p = net.sign_inv_net(self.g, self.pe).squeeze(-1)
            return net(self.g, self.h, p, self.e, self.snorm)[0]

        from . import ops
        # (ops that would read their own status word back — the embeddings' index check — hand it over instead: check() reads them)
        with ops.defer_status() as words:
            self._graphed = GraphedForward(fwd, warmup=warmup)
        self._status_words = list(words)[-max(1, len(words) // (warmup + 1)):] if words else []      # those of the recorded run
        # (a net whose one-launch kernel reports through check_last() remembers the plan of its last forward and forgets it once
        #  checked: the recorded plan's status block is rewritten by every replay, so check() re-arms it)
        self._last_plan = getattr(net, "_last_plan", None)
        self.out = self._graphed.out
        # what the recorded pointers refer to: held here so that the allocator cannot hand the blocks out, and fingerprinted
        self._held = self._cache_objects()
        self._stamp = self._fingerprint()

    def _cache_objects(self):
        held = []
        for m in self.net.modules():
            for name in ("_cache", "_prep", "_fused", "_keep", "_pk", "_bn"):
                v = m.__dict__.get(name)
                if v is not None:
                    held.append((m, name, v))
        return held

    def _fingerprint(self):
        ver = sum(int(t._version) for t in list(self.net.parameters()) + list(self.net.buffers()))
        return (ver, bool(self.net.training), tuple(id(v) for _, _, v in self._held))

    def _assert_fresh(self):
        now = (sum(int(t._version) for t in list(self.net.parameters()) + list(self.net.buffers())), bool(self.net.training),
               tuple(id(m.__dict__.get(name)) for m, name, _ in self._held))
        if now != self._stamp:
            raise RuntimeError("GraphedDGLForward: the network changed since the forward was recorded (train()/eval(), .to(), load_state_dict "
                               "or an update of its parameters): the recorded launches point into the packed weights of the capture — "
                               "record a new GraphedDGLForward")

    def check(self):
        """One host wait: raises what the LAST replay flagged on the device (an atom / bond type outside its embedding table:
        IndexError as nn.Embedding; a batch the stage kernels could not hold: the net's own check_last())."""
        from . import ops
        ops.raise_deferred(self._status_words)
        if hasattr(self.net, "check_last"):
            if self._last_plan is not None:
                self.net._last_plan = self._last_plan
            self.net.check_last()

    def __call__(self, g=None, h=None, pos_enc=None, e=None, snorm_n=None):
        self._assert_fresh()
        todo = []

        def put(dst, src, what):
            if src is None:
                return
            if tuple(src.shape) != tuple(dst.shape):
                raise ValueError(f"GraphedDGLForward: {what} has shape {tuple(src.shape)}, the recorded batch {tuple(dst.shape)}")
            todo.append((dst, src))
        # every argument is validated BEFORE anything is copied: a rejected call leaves the recorded inputs as they were
        if g is not None:
            s, d = g.edges()
            put(self.src, s, "src"); put(self.dst, d, "dst")
            bnn = torch.as_tensor(g.batch_num_nodes())
            put(self.bnn, bnn, "batch_num_nodes")
            if not bnn.is_cuda:            # (host-side counts, as DGL keeps them: the frozen totals are checked here, not left to check())
                nmax, ntot = (int(bnn.max()), int(bnn.sum())) if bnn.numel() else (0, 0)
                if ntot != self.g._sn_node_counts[1]:
                    raise ValueError(f"GraphedDGLForward: batch_num_nodes sums to {ntot}, the recorded batch has {self.g._sn_node_counts[1]} nodes")
                if nmax > max(self.g._sn_node_counts[0], 64):
                    raise ValueError(f"GraphedDGLForward: a graph of {nmax} nodes exceeds the recorded batch's largest graph and the stage kernels' 64")
        put(self.h, h, "h"); put(self.pe, pos_enc, "pos_enc")
        if self.e is not None:
            put(self.e, e, "e")
        if self.snorm is not None:
            put(self.snorm, snorm_n, "snorm_n")
        for dst, src in todo:
            dst.copy_(src, non_blocking=True)
        return self._graphed.replay()


class StreamPipeline:
    def __init__(self, model, streams: int = 3, device=None):
        if streams < 1:
            raise ValueError("streams must be >= 1")
        self.model = model
        dev = device if device is not None else next(model.parameters()).device
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(streams)]
        self._i = 0
        self._inflight = []          # (event, output) in submission order

    def submit(self, data):
        """Queue model(data) on the next stream; `data` must already be resident on the device and must not be modified until
        the result has been collected.  Returns the output tensor (valid after `collect()` / the returned event)."""
        s = self.streams[self._i % len(self.streams)]
        self._i += 1
        s.wait_stream(torch.cuda.current_stream())          # inputs produced on the caller's stream
        with torch.cuda.stream(s), torch.no_grad():
            y = self.model(data)
            ev = torch.cuda.Event()
            ev.record(s)
        self._inflight.append((ev, y))
        return y, ev

    def collect(self):
        """Wait for every submitted forward; returns the outputs in submission order and makes them safe to use on the
        caller's current stream."""
        cur = torch.cuda.current_stream()
        outs = []
        for ev, y in self._inflight:
            cur.wait_event(ev)
            y.record_stream(cur)
            outs.append(y)
        self._inflight = []
        if hasattr(self.model, "check_last"):
            self.model.check_last()
        return outs

    def map(self, batches):
        """Outputs of model(b) for every b in `batches` (an iterable of device-resident batches), in order."""
        for b in batches:
            self.submit(b)
        return self.collect()
