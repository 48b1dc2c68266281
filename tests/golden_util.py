"""Load tests/golden/*.npz fixtures (made by tests/golden/make_golden.py from the reference)."""
import os
import types

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    sd, inp, out, meta, eq = {}, {}, {}, {}, {}
    for k in z.files:
        grp, rest = k.split("/", 1)
        a = z[k]
        if grp == "sd":
            sd[rest] = torch.from_numpy(a)
        elif grp == "in":
            inp[rest] = torch.from_numpy(a) if a.dtype.kind in "fiu" else a
        elif grp == "out":
            out[rest] = torch.from_numpy(a)
        elif grp == "eq":
            eq[rest] = torch.from_numpy(a)
        else:
            meta[rest] = a
    return types.SimpleNamespace(sd=sd, inp=inp, out=out, meta=meta, eq=eq)


def as_data(inp):
    d = types.SimpleNamespace(**{k: v for k, v in inp.items() if k != "sizes"})
    d.sizes = [int(s) for s in inp["sizes"]]
    d.num_graphs = len(d.sizes)
    d.num_nodes = int(sum(d.sizes))
    return d


def full_state_dict(fx):
    """Reference-keyed state_dict with the skipped (forward-irrelevant) tensors zero-filled."""
    sd = dict(fx.sd)
    for k, shp in zip(fx.meta["sd_keys"], fx.meta["sd_shapes"]):
        k = str(k)
        if k not in sd and ".layer.nn." in k:      # second registration of the same GINE MLP tensors
            sd[k] = sd[k.replace(".layer.nn.", ".nn.")]
        if k not in sd:
            shape = tuple(int(s) for s in str(shp).split(",") if s)
            sd[k] = torch.zeros(shape, dtype=torch.long if k.endswith("num_batches_tracked") else torch.float32)
    return sd


PYG_CASES = ["gine_d16", "gine_d44_ragged", "gine_d32_deep", "alchemy_d12", "alchemy_d36"]


def pyg_cfg(fx):
    from oracle import pyg_signnet as O
    c = [None if v < 0 else int(v) for v in fx.meta["ctor"]]
    return O.make_cfg(str(fx.meta["variant"]), *c)


def load_filters(name="learning_filters_grid6"):
    """The LearningFilters training fixture: shared inputs + per case {args, sd, sd1, grad, feat, pre, losses}."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    inp = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in/")}
    cases = {}
    for k in z.files:
        if not k.startswith("c/"):
            continue
        _, cname, rest = k.split("/", 2)
        c = cases.setdefault(cname, {"args": {}, "sd": {}, "sd1": {}, "grad": {}})
        a = z[k]
        if "/" in rest and rest.split("/", 1)[0] in c:
            grp, key = rest.split("/", 1)
            c[grp][key] = a.item() if grp == "args" else torch.from_numpy(a)
        else:
            c[rest] = torch.from_numpy(a) if a.ndim else a.item()
    for c in cases.values():
        c["cfg"] = dict(c["args"], mults=[int(m) for m in c["mults"]] if "mults" in c else [])
    return types.SimpleNamespace(inp=inp, cases=cases, side=int(z["meta/side"]))


def attn_keep_masks(fx, p=0.1, device="cpu", dtype=torch.float32):
    """The attention dropout's keep-masks the reference drew for the `train_do` outputs of a PyG fixture (make_golden.py), scaled as
    nn.Dropout scales them: one [N, heads, K, K] tensor per encoder layer with entries 0 or 1 / (1 - p)."""
    masks, l = [], 0
    while f"train_do/keep{l}" in fx.out:
        shape = [int(v) for v in fx.out[f"train_do/keep{l}_shape"]]
        n = 1
        for v in shape:
            n *= v
        bits = np.unpackbits(fx.out[f"train_do/keep{l}"].numpy())[:n].reshape(shape)
        masks.append((torch.from_numpy(bits.astype(np.float32)) * (1.0 / (1.0 - p))).to(dtype).to(device).contiguous())
        l += 1
    return masks
