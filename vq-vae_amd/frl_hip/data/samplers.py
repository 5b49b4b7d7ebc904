"""Chunk-locked batch sampler: every batch is drawn from ONE (y, x) chunk of the tile store, so a batch touches one chunk
file / one contiguous region of the cube.  Same constructor, iteration order and random streams as the reference's
`ChunkBatchSampler` (utils/samplers.py:42-108): the chunk order is shuffled with `random.Random(seed)`, the order inside a chunk
with numpy's global generator (`np.random.shuffle`), full batches first, then the remainder unless `drop_last`; with
`replacement_within_chunk` each chunk yields ceil(n / batch) batches of `rng.choices`.  tests/golden/sampler_*.json holds
batches produced by the reference class itself."""
from __future__ import annotations

import random
from typing import Iterator, List, Optional, Sequence

import numpy as np
from torch.utils.data import Sampler


class ChunkBatchSampler(Sampler[List[int]]):
    def __init__(self, xy_by_chunk: Sequence[np.ndarray], batch_size: int, drop_last: bool = False,
                 replacement_within_chunk: bool = False, seed: Optional[int] = None) -> None:
        if int(batch_size) <= 0:
            raise ValueError("batch_size must be positive")
        self.xy_by_chunk = [np.asarray(c, dtype=np.int64).reshape(-1) for c in xy_by_chunk]
        self.batch_size, self.drop_last, self.replacement = int(batch_size), bool(drop_last), bool(replacement_within_chunk)
        self.rng = random.Random(seed)
        self.chunk_sizes = [int(c.size) for c in self.xy_by_chunk]
        self.non_empty = [i for i, n in enumerate(self.chunk_sizes) if n]
        self.total = sum(self.chunk_sizes)

    def _chunk_batches(self, members: np.ndarray) -> Iterator[List[int]]:
        bs = self.batch_size
        if self.replacement:
            pool = members.tolist()
            for _ in range(-(-max(1, members.size) // bs)):
                yield self.rng.choices(pool, k=bs)
            return
        stop = members.size - members.size % bs
        for lo in range(0, stop, bs):
            yield members[lo:lo + bs].tolist()
        if stop < members.size and not self.drop_last:
            yield members[stop:].tolist()

    def __iter__(self) -> Iterator[List[int]]:
        order = list(self.non_empty)
        self.rng.shuffle(order)
        members = {}
        for c in order:                       # all within-chunk permutations are drawn before the first batch is emitted
            m = self.xy_by_chunk[c]
            if not self.replacement:
                m = m.copy()
                np.random.shuffle(m)
            members[c] = m
        for c in order:
            yield from self._chunk_batches(members[c])

    def __len__(self) -> int:
        bs = self.batch_size
        return sum(n // bs if self.drop_last else -(-n // bs) for n in self.chunk_sizes)


def shard_batches(batches: Sequence[List[int]], rank: int, world_size: int) -> List[List[int]]:
    """Data-parallel split of an epoch's batch list: rank r takes batches r, r + W, ... of the first floor(n / W) * W batches,
    so every rank runs the same number of steps (the gradient all-reduce needs all ranks in every step)."""
    n = len(batches) // world_size * world_size
    return [batches[i] for i in range(rank, n, world_size)]
