"""A/B of the training step's launch structure (round 6): the same seeded steps of the headline model (hidden 128, k = 16, 128 graphs)
and of a small ragged model under the environment switches SN_TRAIN_FUSE_FINISH / SN_TRAIN_DEFER_DW, one process per setting:
    python profiles/scripts/train_ab_bits.py out.pt        # run under the current environment, save gradients / parameters / losses
    python profiles/scripts/train_ab_bits.py --compare a.pt b.pt
The in-launch finishes and the deferred dW reduction are required to give the SAME BITS as the launches they replace."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def run(path):
    from signnet_basisnet_amd import optim, synth
    from signnet_basisnet_amd.pyg import SignNetGNN
    out = {}
    for tag, ctor, nb, k in (("headline", (None, None, 128, 1, 4, 6), 128, 16), ("small", (None, None, 32, 1, 3, 2), 12, 8)):
        torch.manual_seed(7)
        m = SignNetGNN(*ctor, variant="gine", max_k=k).cuda().train()
        m.attn_dropout = 0.0
        o = optim.FlatAdam(m.parameters(), lr=1e-3)
        d = synth.batch_to(synth.make_batch(nb, seed=5), "cuda")
        target = torch.randn(nb, 1, generator=torch.Generator().manual_seed(2)).cuda()
        losses = []
        for i in range(3):
            o.zero_grad()
            loss = (m(d) - target).abs().mean()
            loss.backward()
            if i == 0:
                out[tag + "/grad0"] = o.flat_g.clone().cpu()
            o.step()
            losses.append(loss.item())
        out[tag + "/losses"] = losses
        out[tag + "/params"] = o.flat_p.clone().cpu()
        out[tag + "/buffers"] = [b.clone().cpu() for b in m.buffers()]
    torch.save(out, path)
    print("saved", path, {k: v for k, v in out.items() if k.endswith("losses")})


def compare(a, b):
    A, B = torch.load(a), torch.load(b)
    ok = True
    for k in A:
        if k.endswith("losses"):
            same = A[k] == B[k]
        elif k.endswith("buffers"):
            same = all(torch.equal(x, y) for x, y in zip(A[k], B[k]))
        else:
            same = torch.equal(A[k], B[k])
            if not same:
                d = (A[k] - B[k]).abs()
                print(k, "max abs diff", d.max().item(), "of", A[k].abs().max().item(), "differing", int((d > 0).sum()), "/", d.numel())
        print(k, "identical" if same else "DIFFERENT")
        ok &= bool(same)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    if sys.argv[1] == "--compare":
        compare(sys.argv[2], sys.argv[3])
    else:
        run(sys.argv[1])
