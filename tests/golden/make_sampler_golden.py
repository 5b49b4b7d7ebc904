"""Writes tests/golden/sampler_batches.json by running the REFERENCE's ChunkBatchSampler (utils/samplers.py:42-108, importable in
the build container: it needs numpy and torch only).  The reference does not travel; only the batch lists do.

    python tests/golden/make_sampler_golden.py        (in the build container, /root/reference present)
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, "/root/reference")
from utils.samplers import ChunkBatchSampler  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def chunks_for(case):
    rng = np.random.default_rng(case["layout_seed"])
    out, nxt = [], 0
    for n in case["chunk_sizes"]:
        out.append(np.arange(nxt, nxt + n, dtype=np.int64))
        nxt += n
    if case.get("scramble"):
        out = [rng.permutation(c) for c in out]
    return out


CASES = [
    {"name": "plain", "chunk_sizes": [10, 0, 7, 16, 3], "batch_size": 4, "drop_last": False, "replacement": False, "seed": 11, "np_seed": 5, "layout_seed": 0, "epochs": 2},
    {"name": "drop_last", "chunk_sizes": [10, 0, 7, 16, 3], "batch_size": 4, "drop_last": True, "replacement": False, "seed": 3, "np_seed": 9, "layout_seed": 0, "epochs": 2},
    {"name": "replacement", "chunk_sizes": [5, 9, 0, 2], "batch_size": 4, "drop_last": False, "replacement": True, "seed": 7, "np_seed": 1, "layout_seed": 0, "epochs": 2},
    {"name": "scrambled_big", "chunk_sizes": [64, 33, 1, 128], "batch_size": 32, "drop_last": False, "replacement": False, "seed": 2024, "np_seed": 77, "layout_seed": 4, "scramble": True, "epochs": 1},
]


def main():
    out = []
    for case in CASES:
        s = ChunkBatchSampler(chunks_for(case), case["batch_size"], drop_last=case["drop_last"],
                              replacement_within_chunk=case["replacement"], seed=case["seed"])
        np.random.seed(case["np_seed"])
        epochs = [[list(map(int, b)) for b in s] for _ in range(case["epochs"])]
        out.append(dict(case, length=len(s), batches=epochs))
    with open(os.path.join(HERE, "sampler_batches.json"), "w") as fh:
        json.dump(out, fh)
    print("wrote", len(out), "cases")


if __name__ == "__main__":
    main()
