"""Writes tests/golden/contrastive_*.npz by running the REFERENCE's contrastive_loss (frl/losses/contrastive.py:29-212) and
extract_at_locations / extract_temporal_at_locations (frl/utils/spatial.py:132-173) -- both importable in the build container (torch
only) -- on seeded inputs, in float64 and in float32, with autograd gradients; checks that the oracle restatement agrees.

    python oracle/make_contrastive_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference/frl")
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from losses.contrastive import contrastive_loss  # noqa: E402
from utils.spatial import extract_at_locations, extract_temporal_at_locations  # noqa: E402

import frl_oracle as O  # noqa: E402

CASES = [dict(name="a", n=120, d=64, p=300, m=900, sim="l2", t=0.07, weights=True, seed=0),
         dict(name="b", n=60, d=12, p=90, m=400, sim="cosine", t=0.1, weights=True, seed=1),
         dict(name="c", n=200, d=64, p=500, m=2500, sim="dot", t=2.0, weights=False, seed=2),
         # anchors with positives but no negatives, negatives of anchors without positives, a zero weight, a single anchor
         dict(name="d", n=40, d=8, p=25, m=60, sim="l2", t=0.05, weights=True, seed=3, edge=True)]


def main():
    for c in CASES:
        g = torch.Generator().manual_seed(c["seed"])
        emb = torch.randn(c["n"], c["d"], generator=g, dtype=torch.float64) * (0.3 if c["sim"] == "dot" else 1.0)
        pos = torch.randint(0, c["n"], (c["p"], 2), generator=g)
        neg = torch.randint(0, c["n"], (c["m"], 2), generator=g)
        if c.get("edge"):
            pos[:, 0] = pos[:, 0] % 10                       # positives only for anchors 0..9
            neg[:20, 0] = 20 + neg[:20, 0] % 10              # negatives of anchors that have no positive: ignored
            neg[20:, 0] = neg[20:, 0] % 8                    # anchors 8, 9: positives but no negatives -> loss 0 for them
        pw = torch.rand(c["p"], generator=g, dtype=torch.float64) + 0.1 if c["weights"] else None
        nw = torch.rand(c["m"], generator=g, dtype=torch.float64) + 0.1 if c["weights"] else None
        if c.get("edge"):
            nw[25] = 0.0                                     # log(0) = -inf: the pair drops out
        e64 = emb.clone().requires_grad_(True)
        l64 = contrastive_loss(e64, pos, neg, pw, nw, temperature=c["t"], similarity=c["sim"])
        l64.backward()
        e32 = emb.float().requires_grad_(True)
        l32 = contrastive_loss(e32, pos, neg, None if pw is None else pw.float(), None if nw is None else nw.float(), temperature=c["t"],
                               similarity=c["sim"])
        l32.backward()
        eo = emb.clone().requires_grad_(True)
        lo = O.contrastive_loss_oracle(eo, pos, neg, pw, nw, c["t"], c["sim"])
        lo.backward()
        assert abs(float(lo) - float(l64)) < 1e-12 * max(1.0, abs(float(l64))), (c["name"], float(lo), float(l64))
        assert (eo.grad - e64.grad).abs().max() < 1e-12 * max(1.0, float(e64.grad.abs().max())), c["name"]
        out = dict(emb=emb.numpy(), pos=pos.numpy(), neg=neg.numpy(), t=c["t"], sim=c["sim"], loss64=float(l64), grad64=e64.grad.numpy(),
                   loss32=float(l32), grad32=e32.grad.numpy())
        if pw is not None:
            out.update(pw=pw.numpy(), nw=nw.numpy())
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", f"contrastive_{c['name']}.npz"), **out)
        print(c["name"], c["sim"], "loss", float(l64), "f32 - f64", float(l32) - float(l64))
    # gathers
    g = torch.Generator().manual_seed(9)
    feat = torch.randn(12, 24, 20, generator=g)
    coords = torch.stack([torch.randint(0, 24, (50,), generator=g), torch.randint(0, 20, (50,), generator=g)], 1)
    coords[7] = coords[3]                                    # a repeated location (gradient rows must add up)
    ft = torch.randn(6, 5, 24, 20, generator=g)
    fa = feat.clone().requires_grad_(True)
    w = torch.randn(50, 12, generator=g)
    (extract_at_locations(fa, coords) * w).sum().backward()
    assert np.array_equal(extract_at_locations(feat, coords).numpy(), O.extract_at_locations_np(feat.numpy(), coords.numpy()))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "extract_locations.npz"), feat=feat.numpy(), coords=coords.numpy(),
                        out=extract_at_locations(feat, coords).numpy(), w=w.numpy(), grad=fa.grad.numpy(), feat_t=ft.numpy(),
                        out_t=extract_temporal_at_locations(ft, coords).numpy())
    print("gather fixtures written")


if __name__ == "__main__":
    main()
