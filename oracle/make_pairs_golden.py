"""Writes tests/golden/mutual_knn_*.npz by running the REFERENCE's pairs_mutual_knn_chunked (frl/losses/pairs.py:531-610; importable
in the build container, torch only) on seeded anchors, and checks that the oracle restatement returns the same pairs.

    python oracle/make_pairs_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference/frl")
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from losses.pairs import pairs_mutual_knn_chunked  # noqa: E402

import frl_oracle as O  # noqa: E402

CASES = [dict(name="a", patches=[120, 97, 150], D=64, k=8, min_sp=4.0, size=32, seed=0, chunk=128),
         dict(name="b", patches=[300], D=12, k=16, min_sp=6.0, size=24, seed=1, chunk=64),
         dict(name="c", patches=[5, 3], D=8, k=16, min_sp=2.0, size=8, seed=2, chunk=128),          # k > N - 1: -1 padding
         dict(name="d", patches=[200, 180, 220, 210], D=64, k=4, min_sp=4.0, size=32, seed=3, chunk=100)]


def main():
    for c in CASES:
        g = torch.Generator().manual_seed(c["seed"])
        n = sum(c["patches"])
        feats = torch.randn(n, c["D"], generator=g)
        coords = [torch.stack([torch.randint(0, c["size"], (m,), generator=g), torch.randint(0, c["size"], (m,), generator=g)], 1) for m in c["patches"]]
        offsets = [0] + list(np.cumsum(c["patches"]))
        ref = pairs_mutual_knn_chunked(feats, coords, [int(o) for o in offsets], c["k"], pos_min_spatial=c["min_sp"], chunk_size=c["chunk"])
        mine, _ = O.mutual_knn_pairs_np(feats.numpy(), [x.numpy() for x in coords], [int(o) for o in offsets], c["k"], c["min_sp"])
        key = lambda a: sorted(map(tuple, a.tolist()))                                       # noqa: E731
        assert key(ref.numpy()) == key(mine), c["name"]          # same set of pairs (row order inside an anchor: topk tie order)
        assert np.array_equal(ref.numpy(), mine), c["name"]      # and, on these inputs, the same order
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", f"mutual_knn_{c['name']}.npz"), features=feats.numpy(),
                            coords=np.concatenate([x.numpy() for x in coords]).astype(np.int64), offsets=np.asarray(offsets, dtype=np.int64),
                            k=c["k"], min_sp=c["min_sp"], pairs=ref.numpy())
        print(c["name"], n, "anchors ->", ref.shape[0], "pairs")


if __name__ == "__main__":
    main()
