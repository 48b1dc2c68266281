"""-m gpu: the differentiable train-mode forward of SignNetGNN: parameter gradients and a few Adam steps
against torch.autograd / torch.optim.Adam running the fp32 and the float64 CPU oracle on the reference's fixtures (same
state_dict, same batch) and at the BASELINE sizes.  Rule: within 1e-5 of the fp32 oracle's gradient, else attributed to float64
(`_grad_rule`): the HIP gradient is no further from the exact one than the reference's own fp32 arithmetic."""
import pytest
import torch

import golden_util as G
import parity_util as PU
from test_pyg_parity_gpu import build

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def oracle_setup(fx, dt=torch.float64):
    sd = {k: (v.clone().to(dt).requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone().to(dt)
              if v.is_floating_point() else v.clone()) for k, v in fx.sd.items()}
    data = G.as_data(fx.inp)
    for a in ("eigen_values", "eigen_vectors"):
        setattr(data, a, getattr(data, a).to(dt))
    if data.x.is_floating_point():
        data.x, data.edge_attr = data.x.to(dt), data.edge_attr.to(dt)
    return sd, data


def ref_grad(sd, name):
    """Gradient the oracle assigns to a model parameter; PyG registers GINEConv's MLP twice (`.nn.` and `.layer.nn.`)."""
    g = None
    for k in (name, name.replace(".nn.", ".layer.nn.", 1)):
        if k in sd and sd[k].requires_grad and sd[k].grad is not None:
            g = sd[k].grad if g is None else g + sd[k].grad
        if k == name.replace(".nn.", ".layer.nn.", 1) and k == name:
            break
    return g


@pytest.mark.parametrize("name", ["gine_d16", "gine_d44_ragged", "alchemy_d12", "alchemy_d36"])
def test_parameter_gradients_match_oracle_autograd(name):
    from oracle import pyg_signnet as O
    from signnet_basisnet_amd import synth
    fx = G.load(name)
    model = build(fx).train()
    data = synth.batch_to(G.as_data(fx.inp), DEV)
    y = model(data)
    assert y.requires_grad
    cot = torch.randn(y.shape, generator=torch.Generator().manual_seed(0), dtype=torch.float64)
    (y * cot.float().to(DEV)).sum().backward()
    sds, ys = {}, {}
    tp = None
    for dt in (torch.float64, torch.float32):
        sd, odata = oracle_setup(fx, dt)
        if dt == torch.float64:
            with term_probe() as tp:
                yo = O.signnet_gnn(sd, G.pyg_cfg(fx), odata, training=True)
                (yo * cot.to(dt)).sum().backward()
        else:
            yo = O.signnet_gnn(sd, G.pyg_cfg(fx), odata, training=True)
            (yo * cot.to(dt)).sum().backward()
        sds[dt], ys[dt] = sd, yo.detach()
    PU.close(y, ys[torch.float32], "train-mode forward", ref64=ys[torch.float64])
    _assert_gradient_population(model.named_parameters(), sds[torch.float32], sds[torch.float64], name, 40, term_norms=tp.norms)


def test_adam_training_steps_follow_the_oracle():
    """Three optimiser steps (forward, L1 loss as main_alchemy.py:106 / train.py:60, backward, Adam) on the device against
    the same three steps of torch.optim.Adam on the float64 oracle: losses and the final eval output agree."""
    from oracle import pyg_signnet as O
    from signnet_basisnet_amd import optim, synth
    fx = G.load("gine_d16")
    model = build(fx).train()
    data = synth.batch_to(G.as_data(fx.inp), DEV)
    sd, odata = oracle_setup(fx)
    cfg = G.pyg_cfg(fx)
    target = torch.randn(len(odata.sizes), int(fx.meta["ctor"][3]), generator=torch.Generator().manual_seed(3), dtype=torch.float64)
    opt = optim.Adam(model.parameters(), lr=1e-3)
    names = {k for k, _ in model.named_parameters()}
    # the oracle holds GINEConv's MLP under two names; optimise the one it reads
    oparams = [v for k, v in sd.items() if torch.is_tensor(v) and v.requires_grad]
    oopt = torch.optim.Adam(oparams, lr=1e-3)
    for step in range(3):
        opt.zero_grad()
        loss = (model(data) - target.float().to(DEV)).abs().mean()
        loss.backward()
        opt.step()
        oopt.zero_grad()
        # running statistics are buffers of the module; the functional oracle does not carry them between calls
        oloss = (O.signnet_gnn(sd, cfg, odata, training=True) - target).abs().mean()
        oloss.backward()
        oopt.step()
        assert abs(loss.item() - oloss.item()) <= 2e-3 * max(1.0, abs(oloss.item())), (step, loss.item(), oloss.item())
    with torch.no_grad():
        y = model(data)          # still train mode: batch statistics, no running-stat dependence
        yo = O.signnet_gnn(sd, cfg, odata, training=True)
    err = (y.cpu().double() - yo).abs().max().item()
    assert err <= 5e-3 * max(1.0, yo.abs().max().item()), err
    assert names


def test_three_hundred_adam_steps_track_the_float64_oracle_loss_curve():
    """No dataset ships with this image, so accuracy on real data (GraphPrediction/README.md:14-30) cannot be reproduced; what can be
    bounded is drift: 300 optimiser steps (forward, L1 loss, backward, Adam, lr 1e-3) on the device against the same 300 steps of
    torch.optim.Adam on the float64 oracle AND on the float32 oracle (what the reference computes), from the same weights on the same
    batch and targets (curves stored by tests/golden/make_adam_curve.py).  L1's sign gradients make a trajectory sensitive to rounding
    (one residual changing sign moves the loss curve by a percent for a few steps), so the bound is relative to what fp32 itself
    does: the HIP curve stays as close to the float64 one as the fp32 CPU curve does (x 4), its first ten steps within 1e-5, and all
    three learn."""
    import os
    import numpy as np
    from signnet_basisnet_amd import optim, synth
    z = np.load(os.path.join(G.GOLDEN, "adam_curve_gine_d16.npz"))
    o64, o32 = z["f64"].tolist(), z["f32"].tolist()
    steps = int(z["meta"][0])
    fx = G.load("gine_d16")
    model = build(fx).train()
    data = synth.batch_to(G.as_data(fx.inp), DEV)
    tdev = torch.from_numpy(z["target"]).float().to(DEV)
    opt = optim.FlatAdam(model.parameters(), lr=float(z["lr"]))
    losses = []
    for step in range(steps):
        opt.zero_grad()
        loss = (model(data) - tdev).abs().mean()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    dev = [abs(a - b) / max(1.0, abs(b)) for a, b in zip(losses, o64)]
    dev32 = [abs(a - b) / max(1.0, abs(b)) for a, b in zip(o32, o64)]
    mean = lambda v: sum(v) / len(v)
    print(f"\n{steps} Adam steps: loss {o64[0]:.4f} -> {o64[-1]:.4f} (float64 oracle), {losses[0]:.4f} -> {losses[-1]:.4f} (HIP fp32), "
          f"{o32[-1]:.4f} (fp32 oracle); deviation from float64 per step: HIP max {max(dev):.2e} mean {mean(dev):.2e} first ten {max(dev[:10]):.2e}; "
          f"fp32 CPU oracle max {max(dev32):.2e} mean {mean(dev32):.2e}")
    for ls in (o64, o32, losses):
        assert ls[-1] < 0.9 * ls[0], "every trajectory learns"
    assert max(dev[:10]) <= 1e-5, max(dev[:10])
    assert max(dev) <= 4 * max(dev32) + 1e-3 and mean(dev) <= 4 * mean(dev32) + 1e-4, (max(dev), max(dev32), mean(dev), mean(dev32))


def test_flat_adam_equals_per_tensor_adam_and_trains():
    """FlatAdam (one kernel launch over the flat parameter buffer, gradients accumulated straight into the flat gradient
    buffer by autograd) gives the same parameters as the per-tensor Adam after three real training steps."""
    from signnet_basisnet_amd import optim, synth
    fx = G.load("gine_d32_deep")
    data = synth.batch_to(G.as_data(fx.inp), DEV)
    target = torch.randn(len(G.as_data(fx.inp).sizes), 1, generator=torch.Generator().manual_seed(4)).to(DEV)
    finals, losses = [], []
    for cls in (optim.Adam, optim.FlatAdam):
        model = build(fx).train()
        opt = cls(model.parameters(), lr=2e-3)
        ls = []
        for _ in range(3):
            opt.zero_grad()
            loss = (model(data) - target).abs().mean()
            loss.backward()
            opt.step()
            ls.append(loss.item())
        losses.append(ls)
        finals.append({k: v.detach().clone() for k, v in model.named_parameters()})
    assert losses[0][-1] < losses[0][0]                     # it learns
    for a, b in zip(*losses):
        assert abs(a - b) <= 1e-4 * max(1.0, abs(a))
    for k in finals[0]:
        # (with weight_decay the two differ on parameters the forward never uses: a None gradient is skipped, FlatAdam's
        # zero gradient is not — the same difference as torch's zero_grad(set_to_none=True / False))
        torch.testing.assert_close(finals[1][k], finals[0][k], rtol=2e-3, atol=2e-5)


@pytest.mark.parametrize("name", ["dgl_gin_k8", "dgl_masked_k10"])
def test_deepsigns_gradients_match_oracle_autograd(name):
    """GraphPrediction tree: the gradient that the DGL base network sends back into sign_inv_net's output reaches every
    parameter of GINDeepSigns / MaskedGINDeepSigns as torch.autograd computes it on the float64 oracle."""
    from oracle import dgl_deepsigns as OD
    from signnet_basisnet_amd import dgl_deepsigns as DS
    fx = G.load(name)
    hidden, c, layers, k = (int(v) for v in fx.meta["params"])
    kind = str(fx.meta["kind"])
    net = DS.get_sign_inv_net(dict(sign_inv_net=kind, hidden_dim=hidden, phi_out_dim=c, sign_inv_layers=layers, pos_enc_dim=k,
                                   dropout=0.0, sign_inv_activation="relu", device=DEV))
    net.load_state_dict(fx.sd)
    net = net.to(DEV).train()
    ei, sizes = fx.inp["edge_index"], fx.inp["sizes"]
    x = fx.inp["pos_enc"].unsqueeze(-1)
    y = net(DS.Graph(ei[0].to(DEV), ei[1].to(DEV), sizes), x.to(DEV))
    assert y.requires_grad
    cot = torch.randn(y.shape, generator=torch.Generator().manual_seed(0), dtype=torch.float64)
    (y * cot.float().to(DEV)).sum().backward()
    sd = {kk: (v.clone().double().requires_grad_(True) if v.is_floating_point() and "running" not in kk else
               (v.clone().double() if v.is_floating_point() else v.clone())) for kk, v in fx.sd.items()}
    if kind == "gin":
        yo = OD.gin_deepsigns(sd, ei[0], ei[1], x.double(), layers, k, training=True)
    else:
        yo = OD.masked_gin_deepsigns(sd, ei[0], ei[1], torch.as_tensor(sizes), x.double(), layers, k, training=True)
    assert (y.detach().cpu().double() - yo.detach()).abs().max().item() <= 5e-4 * max(1.0, yo.abs().max().item())
    (yo * cot).sum().backward()
    gmax = max(v.grad.abs().max().item() for v in sd.values() if v.requires_grad and v.grad is not None)
    checked = 0
    for pname, p in net.named_parameters():
        gr = sd[pname].grad
        if gr is None:
            assert p.grad is None or p.grad.abs().max().item() <= 1e-5 * gmax + 1e-6, pname
            continue
        e = (p.grad.detach().cpu().double() - gr).abs().max().item()
        assert e <= 2e-3 * gr.abs().max().item() + 1e-5 * gmax + 1e-6, f"{pname}: {e:.3e} vs {gr.abs().max().item():.3e}"
        checked += 1
    assert checked >= 10


def test_dgl_gin_base_net_training_step_gradients():
    """loss.backward() through GINNet AND the sign_inv_net that produced its p (one autograd graph, as in
    train_ZINC_graph_regression.py:70-85): every parameter gradient against torch.autograd on the float64 oracle."""
    from oracle import dgl_nets as ON
    from signnet_basisnet_amd import dgl_deepsigns as DS
    from test_dgl_basisnet_gpu import _ginnet
    fx = G.load("dgl_ginnet_k6")
    hidden, L, k = (int(v) for v in fx.meta["params"])
    net = _ginnet(fx).train()
    ei = fx.inp["edge_index"]
    g = DS.Graph(ei[0].to(DEV), ei[1].to(DEV), fx.inp["sizes"])
    target = torch.randn(len(fx.inp["sizes"]), 1, generator=torch.Generator().manual_seed(1), dtype=torch.float64)
    p = net.sign_inv_net(g, fx.inp["pos_enc"].unsqueeze(-1).to(DEV)).squeeze(-1)
    y, _ = net(g, fx.inp["x"].squeeze(-1).to(DEV), p, fx.inp["edge_attr"].to(DEV), None)
    net.loss(y, target.float().to(DEV)).backward()
    sd = {kk: (v.clone().double().requires_grad_(True) if v.is_floating_point() and "running" not in kk and not kk.endswith("eps") else
               (v.clone().double() if v.is_floating_point() else v.clone())) for kk, v in fx.sd.items()}
    yo, _ = ON.gin_net_with_sign_inv(sd, ei[0], ei[1], fx.inp["sizes"], fx.inp["x"].squeeze(-1), fx.inp["pos_enc"].double(), L, 3, k,
                                     training=True)
    (yo - target).abs().mean().backward()
    gmax = max(v.grad.abs().max().item() for v in sd.values() if v.requires_grad and v.grad is not None)
    checked = 0
    for pname, prm in net.named_parameters():
        gr = sd[pname].grad
        if gr is None or gr.abs().max() == 0:
            assert prm.grad is None or prm.grad.abs().max().item() <= 1e-5 * gmax + 1e-7, pname
            continue
        e = (prm.grad.detach().cpu().double() - gr).abs().max().item()
        assert e <= 2e-3 * gr.abs().max().item() + 1e-5 * gmax + 1e-7, f"{pname}: {e:.3e} vs {gr.abs().max().item():.3e}"
        checked += 1
    assert checked >= 25, checked


@pytest.mark.parametrize("name", ["dgl_gatedgcn_concat_k6", "dgl_gatedgcn_add_k8"])
def test_dgl_gatedgcn_training_step_gradients(name):
    """loss.backward() through GatedGCNNet (edge-gated aggregation adjoint) and its MaskedGINDeepSigns: parameter gradients
    against torch.autograd on the float64 oracle."""
    from oracle import dgl_deepsigns as OD
    from oracle import dgl_nets as ON
    from signnet_basisnet_amd import dgl_deepsigns as DS
    from test_dgl_basisnet_gpu import _gatedgcn
    fx = G.load(name)
    hidden, L, k = (int(v) for v in fx.meta["params"])
    net = _gatedgcn(fx).train()
    ei, sizes = fx.inp["edge_index"], fx.inp["sizes"]
    g = DS.Graph(ei[0].to(DEV), ei[1].to(DEV), sizes)
    target = torch.randn(len(sizes), 1, generator=torch.Generator().manual_seed(1), dtype=torch.float64)
    p = net.sign_inv_net(g, fx.inp["pos_enc"].unsqueeze(-1).to(DEV)).squeeze(-1)
    y, _ = net(g, fx.inp["x"].squeeze(-1).to(DEV), p, fx.inp["edge_attr"].to(DEV), None)
    net.loss(y, target.float().to(DEV)).backward()
    sd = {kk: (v.clone().double().requires_grad_(True) if v.is_floating_point() and "running" not in kk and not kk.endswith("eps") else
               (v.clone().double() if v.is_floating_point() else v.clone())) for kk, v in fx.sd.items()}
    ssd = {kk[len("sign_inv_net."):]: v for kk, v in sd.items() if kk.startswith("sign_inv_net.")}
    po = OD.masked_gin_deepsigns(ssd, ei[0], ei[1], torch.as_tensor(sizes), fx.inp["pos_enc"].double().unsqueeze(-1), 3, k,
                                 training=True).squeeze(-1)
    yo = ON.gatedgcn_net(sd, ei[0], ei[1], sizes, fx.inp["x"].squeeze(-1), po, fx.inp["edge_attr"], L,
                         pe_aggregate=str(fx.meta["pe_aggregate"]), training=True)
    assert (y.detach().cpu().double() - yo.detach()).abs().max().item() <= 1e-3 * max(1.0, yo.abs().max().item())
    (yo - target).abs().mean().backward()
    gmax = max(v.grad.abs().max().item() for v in sd.values() if v.requires_grad and v.grad is not None)
    checked = 0
    for pname, prm in net.named_parameters():
        gr = sd[pname].grad
        if gr is None or gr.abs().max() == 0:
            assert prm.grad is None or prm.grad.abs().max().item() <= 2e-5 * gmax + 1e-7, pname
            continue
        e = (prm.grad.detach().cpu().double() - gr).abs().max().item()
        assert e <= 3e-3 * gr.abs().max().item() + 2e-5 * gmax + 1e-7, f"{pname}: {e:.3e} vs {gr.abs().max().item():.3e} (max {gmax:.3e})"
        checked += 1
    assert checked >= 30, checked


_RCCL_SCRIPT = r'''
import json, os, sys
sys.path.insert(0, os.environ["SN_ROOT"]); sys.path.insert(0, os.path.join(os.environ["SN_ROOT"], "tests"))
import torch
import golden_util as G
import parity_util as PU
from test_pyg_parity_gpu import build
from signnet_basisnet_amd import dist as D, optim, synth
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ["SN_PORT"])
dist = D.init_process_group("nccl")                      # RCCL: communicator init + ncclAllReduce really execute, world size 1
fx = G.load("gine_d16")
data = synth.batch_to(G.as_data(fx.inp), "cuda:0")
target = torch.randn(len(G.as_data(fx.inp).sizes), 1, generator=torch.Generator().manual_seed(4)).to("cuda:0")
out = {"backend": dist.get_backend(), "world": dist.get_world_size()}
finals, early = [], []
for use_dist in (None, dist):
    model = build(fx).train()
    opt = optim.FlatAdam(model.parameters(), lr=2e-3, dist=use_dist, bucket_mb=0.05)
    for _ in range(3):
        opt.zero_grad()
        (model(data) - target).abs().mean().backward()
        n_early = sum(w is not None for w in opt._work)
        opt.step()
        early.append(n_early)
    torch.cuda.synchronize()
    finals.append(opt.flat_p.clone())
out["buckets"] = len(opt.buckets)
out["early"] = early[3:]
out["equal"] = bool(torch.equal(finals[0], finals[1]))       # SUM over one rank, scale 1: bit-identical parameters
# the captured step with a process group: replay, then step() all-reduces every bucket (RCCL) and launches Adam — the same trajectory
from signnet_basisnet_amd.train_graph import GraphedStep
model = build(fx).train(); model.max_k = 16
opt = optim.FlatAdam(model.parameters(), lr=2e-3, dist=None)
for _ in range(3):
    opt.zero_grad(); torch.nn.functional.l1_loss(model(data), target).backward(); opt.step()
torch.cuda.synchronize(); want = opt.flat_p.clone()
model = build(fx).train(); model.max_k = 16
opt = optim.FlatAdam(model.parameters(), lr=2e-3, dist=dist, bucket_mb=0.05)
gs = GraphedStep(model, opt, data, target)
for _ in range(3):
    gs.step()
torch.cuda.synchronize()
out["graphed_equal"] = bool(torch.equal(opt.flat_p, want)) and not opt._hooks
probe = torch.ones(1 << 20, device="cuda:0")
dist.all_reduce(probe); torch.cuda.synchronize()
out["probe"] = float(probe.sum())
dist.destroy_process_group()
print("RESULT " + json.dumps(out))
'''


def test_flat_adam_over_rccl_world_size_one(tmp_path):
    """`FlatAdam.step()` with `backend="nccl"` (RCCL) on device buffers: communicator init and ncclAllReduce execute on the
    one-GPU box (world size 1), the bucketed all-reduces are issued from inside the backward from the second step on, and the
    trained parameters are bit-identical to the run without a process group."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for attempt in range(2):            # (the rendezvous port is picked free, then released: one retry on a fresh port if it was taken)
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, SN_ROOT=root, SN_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        r = subprocess.run([sys.executable, "-c", _RCCL_SCRIPT], env=env, capture_output=True, text=True, timeout=600, cwd=root)
        if r.returncode == 0 or "address already in use" not in (r.stderr + r.stdout).lower():
            break
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert out["backend"] == "nccl" and out["world"] == 1
    assert out["equal"], "all-reduce over one rank must not change the gradients"
    assert out["probe"] == float(1 << 20)
    assert out["graphed_equal"], "GraphedStep with a process group (replay, then the all-reduce + Adam) must follow the eager trajectory"
    # (buckets that hold only parameters the forward never uses — GNN3d.edge_encoders, pos_encoder, ... — have no last gradient to
    #  trigger them and are reduced in step(); every other bucket goes out from inside the backward from the second step on)
    assert out["buckets"] >= 3 and out["early"][0] == 0 and min(out["early"][1:]) >= out["buckets"] // 2, out


class term_probe:
    """`with term_probe() as tp:` around the FLOAT64 oracle's forward + backward: tp.norms[name] = sqrt(sum_r t_r^2) of the per-row terms
    t_r whose sum is the gradient of the one-entry parameter `name` (oracle.pyg_signnet.TERM_PROBE)."""

    def __enter__(self):
        from oracle import pyg_signnet as O
        O.TERM_PROBE = {}
        self.norms = {}
        return self

    def __exit__(self, *a):
        from oracle import pyg_signnet as O
        probe, O.TERM_PROBE = O.TERM_PROBE, None
        for k, v in probe.items():
            if v.grad is not None:
                self.norms[k] = float(v.grad.double().pow(2).sum().sqrt())
        return False


def _assert_gradient_population(named_params, sd32, sd64, label, min_tensors, term_norms=None):
    """Every parameter gradient against the fp32 and the float64 oracle gradients.
    The fp32 oracle's OWN gradient sits 1e-3 ... 1e-2 (relative) from the float64 one as soon as there are tens of thousands of
    ReLU / BatchNorm decisions: a handful fall on the other side of zero in any fp32 evaluation, each moving a gradient entry by
    O(1).  Two fp32 evaluations (the reference's and this one) are then two draws from the same error distribution, so "never
    further from float64 than cpu32" cannot hold tensor by tensor (the reference's own luck on a tensor is not a bound).  Asserted:
      (a) every tensor: within 1e-5 of the fp32 oracle, or |hip - f64| <= 8 x max(|cpu32 - f64|, L) (+1e-6), L = the 90th percentile of
          the fp32 oracle's own relative distances over all tensors (its error LEVEL at this size: the reference's luck on one tensor
          is not a bound; measured worst: 6 x L on a [108,108] weight of the 28-layer Alchemy model, against O(1) for a wrong adjoint);
      (b) the median over the tensors above 1e-5 of |hip - f64| / |cpu32 - f64| is <= 2: as a population the HIP gradients are as
          exact as the reference's fp32 gradients (measured: 0.4 on the headline model, 1.2-1.4 on the 28-layer Alchemy model, for the
          layer-at-a-time kernels and the stage kernels alike)."""
    gmax = max(v.grad.abs().max().item() for v in sd64.values() if torch.is_tensor(v) and v.requires_grad and v.grad is not None)
    rows = []
    for pname, p in named_params:
        g64, g32 = ref_grad(sd64, pname), ref_grad(sd32, pname)
        if g64 is None:
            assert p.grad is None or p.grad.abs().max().item() <= 1e-6 * gmax, f"{pname}: gradient where the reference has none"
            continue
        assert p.grad is not None, f"{pname}: no gradient"
        hip, a32, a64 = p.grad.detach().cpu().double(), g32.detach().double(), g64.detach().double()
        s64 = a64.abs().max().item()
        if s64 <= 1e-7 * gmax:        # a bias in front of a batch-statistics BatchNorm: exact gradient 0
            assert hip.abs().max().item() <= 1e-6 * gmax + 10 * a32.abs().max().item(), f"{pname}: gradient where the exact one is zero"
            continue
        # A ONE-ENTRY gradient (GIN / GINE eps, the Linear(1,1) / BatchNorm1d(1) of GINESignNetPyG's first phi layer) is one sum over up to
        # ~10^5 rows, often a cancelling one: it has no largest entry to normalise by but itself, and |entry| says nothing about the size
        # of what was summed.  Its scale is the root-sum-square of its per-row terms (sqrt(rows) x rms |term|: what the sum would be without
        # cancellation), read off the float64 oracle (term_probe) — the absolute scale the round-4 / round-5 advice asked for.
        tn = None
        if term_norms is not None and hip.numel() <= 2:
            for k in (pname, pname.replace(".nn.", ".layer.nn.", 1)):
                if k in term_norms:
                    tn = term_norms[k]
        sc = s64 if tn is None else max(s64, tn)
        rows.append((pname, (hip - a64).abs().max().item() / sc, (a32 - a64).abs().max().item() / sc,
                     (hip - a32).abs().max().item() / max(a32.abs().max().item(), 1e-300), hip.numel(), None if tn is None else tn / s64))
    assert len(rows) >= min_tensors, len(rows)
    level = sorted(r[2] for r in rows)[int(0.9 * (len(rows) - 1))]
    ratios = sorted(r[1] / max(r[2], 1e-300) for r in rows if max(r[1], r[2]) > PU.REL)
    med = ratios[len(ratios) // 2] if ratios else 0.0
    worst = max(rows, key=lambda r: r[1] / max(r[2], 1e-300))
    print(f"{label}: {len(rows)} parameter tensors, {len(ratios)} above 1e-5; fp32 oracle's own error level (90th pct) {level:.2e}; "
          f"median |hip-f64|/|cpu32-f64| {med:.2f}; worst ratio {worst[1] / max(worst[2], 1e-300):.1f} ({worst[1]:.2e} vs {worst[2]:.2e}) at {worst[0]}")
    for pname, eh, ec, e32, numel, tn in rows:
        # Every tensor is held to the same factor: 8 x the larger of the reference's own distance on that tensor and its error level.
        # (Rounds 4-5 allowed tensors of one or two entries 16 x: relative to the ENTRY, the BatchNorm1d(1) bias of the headline model sat
        #  at 8.1 x the level — its terms are 9.6 x the entry; relative to the terms it sits below the level.  Roundoff alone — 2^-24 x
        #  sqrt(sum t_r^2) — is 3-5 orders of magnitude below every distance measured here, the float32 oracle's included: what moves
        #  these gradients are ReLU decisions that fall on the other side of zero in any fp32 evaluation, each worth one term.)
        if tn is not None:
            print(f"   one-entry {pname}: rss(terms) / |entry| = {tn:.3g}; against that scale |hip-f64| {eh:.2e}  |cpu32-f64| {ec:.2e}")
        assert e32 <= PU.REL or eh <= 8.0 * max(ec, level) + PU.ATTR, (
            f"{pname}: |hip - cpu32| {e32:.2e}; |hip - f64| {eh:.2e} vs |cpu32 - f64| {ec:.2e} (oracle error level {level:.2e})")
    assert len(ratios) < 10 or med <= 2.0, med


SIZE_CASES = {
    # BASELINE configs[1] / the per-rank shape of configs[3]: GINESignNetPyG SignNetGNN(None,None,128,1,4,6), 128 graphs, k = 16
    "configs1_zinc_k16_h128_b128": dict(variant="gine", ctor=(None, None, 128, 1, 4, 6), feat="zinc", lo=9, hi=37, B=128, k=16, seed=1235),
    # BASELINE configs[2]: main_alchemy.py:35 SignNetGNN(6,4,108,12,8,16), 256 graphs, all eigenvectors
    "configs2_alchemy_b256": dict(variant="alchemy", ctor=(6, 4, 108, 12, 8, 16), feat="alchemy", lo=6, hi=14, B=256, k=None, seed=1236),
}


@pytest.mark.parametrize("name", list(SIZE_CASES))
def test_parameter_gradients_at_baseline_size_vs_oracle_fp32_and_fp64(name):
    """d loss / d theta of a training step AT SIZE against torch.autograd over the fp32 oracle (what the reference's loss.backward()
    computes) and over the float64 oracle (the exact gradient), every parameter tensor, rule of `_grad_rule`."""
    import parity_util as PU_
    from oracle import pyg_signnet as O
    from signnet_basisnet_amd import synth
    from signnet_basisnet_amd.pyg import SignNetGNN
    c = SIZE_CASES[name]
    torch.manual_seed(0)
    m = SignNetGNN(*c["ctor"], variant=c["variant"], max_k=c["k"])
    m.attn_dropout = 0.0
    data = synth.make_batch(c["B"], seed=c["seed"], n_lo=c["lo"], n_hi=c["hi"], features=c["feat"])
    cfg = O.make_cfg(c["variant"], *c["ctor"])
    n_out = c["ctor"][3]
    cot = torch.randn(c["B"], n_out, generator=torch.Generator().manual_seed(1), dtype=torch.float64)

    def oracle(dt):
        sd = {}
        for k, v in m.state_dict().items():
            if v.is_floating_point():
                t = v.detach().clone().to(dt)
                sd[k] = t.requires_grad_(True) if "running" not in k else t
            else:
                sd[k] = v.detach().clone()
        dd = PU_.data_f64(data) if dt == torch.float64 else data
        y = O.signnet_gnn(sd, cfg, dd, training=True, max_k=c["k"])
        (y * cot.to(dt)).sum().backward()
        return y.detach(), sd

    with term_probe() as tp:
        y64, sd64 = oracle(torch.float64)
    y32, sd32 = oracle(torch.float32)
    m = m.cuda().train()
    y = m(synth.batch_to(data, DEV))
    assert y.requires_grad
    (y * cot.float().to(DEV)).sum().backward()
    PU_.close(y, y32, "train-mode forward at size", ref64=y64)
    _assert_gradient_population(m.named_parameters(), sd32, sd64, name, 60, term_norms=tp.norms)


def test_graphed_step_replays_the_eager_training_step_bit_for_bit():
    """train_graph.GraphedStep: forward + L1 loss + backward captured once as a HIP graph (fixed batch shape) and replayed, Adam outside
    — same kernels, same order: the loss trajectory and the parameters equal the eager loop's bit for bit; a batch of another shape is
    refused; a new batch of the same shape is taken through the static input buffers."""
    from signnet_basisnet_amd import optim, synth
    from signnet_basisnet_amd.pyg import SignNetGNN
    from signnet_basisnet_amd.train_graph import GraphedStep
    host = synth.make_batch(12, seed=5)
    host2 = synth.make_batch(12, seed=5)
    host2.x = (host2.x + 1) % 28                                   # same shape, other node types
    target = torch.randn(12, 1, generator=torch.Generator().manual_seed(2)).to(DEV)

    def model_and_opt():
        torch.manual_seed(7)
        m = SignNetGNN(None, None, 32, 1, 3, 2, variant="gine", max_k=8).to(DEV).train()      # attention dropout 0.1 stays on
        return m, optim.FlatAdam(m.parameters(), lr=2e-3)

    m1, o1 = model_and_opt()
    torch.manual_seed(11)
    eager = []
    for i in range(4):
        d = synth.batch_to(host if i < 2 else host2, DEV)
        o1.zero_grad()
        loss = (m1(d) - target).abs().mean()
        loss.backward()
        o1.step()
        eager.append(loss.item())
    m2, o2 = model_and_opt()
    torch.manual_seed(11)                                           # the same dropout draws as the eager loop
    gs = GraphedStep(m2, o2, synth.batch_to(host, DEV), target.clone())
    graphed = [gs.step().item(), gs.step().item(), gs.step(synth.batch_to(host2, DEV)).item(), gs.step().item()]
    m2.check_train()
    assert graphed == eager, (graphed, eager)
    assert torch.equal(o1.flat_p, o2.flat_p)
    for b1, b2 in zip(m1.buffers(), m2.buffers()):
        assert torch.equal(b1, b2)                                 # running statistics / counters advanced by the same steps
    with pytest.raises(ValueError, match="shape"):
        gs.step(synth.batch_to(synth.make_batch(12, seed=6), DEV))
    # the model is left as it was found (round-3 advice): the capture's switches are scoped to the capture, so an eager train-mode
    # step on a batch of ANOTHER shape (a ragged last batch) draws its own dropout masks and checks its embedding indices
    gs.check()
    assert not getattr(m2, "_defer_status", False) and getattr(m2, "_attn_masks", None) is None
    small = synth.make_batch(7, seed=9)
    o2.zero_grad()
    (m2(synth.batch_to(small, DEV)) - target[:7]).abs().mean().backward()
    small.x = small.x + 1000
    with pytest.raises(IndexError):
        m2(synth.batch_to(small, DEV))
