"""Pair mining for the contrastive objective on the GPU.

`pairs_mutual_knn_chunked` keeps the reference's signature and result layout (frl/losses/pairs.py:531-610; called from
frl/training/representation/step.py:721-812 on N ~ 100-300 anchors per patch): a [P, 2] int64 tensor of (anchor, target) rows, both
(i, j) and (j, i) for each mutual pair, ordered by anchor and then by neighbour rank.  The distance matrix, the top-k and the mutual
test run in two HIP kernels (`frl_mutual_knn`); `chunk_size` is accepted for call compatibility and ignored (nothing N x N is ever
materialised).  Neighbour order on exactly tied distances is (distance, index) here; torch.topk leaves it unspecified.
"""
from __future__ import annotations

from typing import List, Sequence

import torch

from .. import ops


def pairs_mutual_knn_chunked(features: torch.Tensor, coord_list: Sequence[torch.Tensor], offsets: List[int], k: int,
                             pos_min_spatial: float = 4.0, chunk_size: int = 128) -> torch.Tensor:
    n = features.shape[0]
    dev = features.device
    if n == 0 or k <= 0:
        return torch.empty((0, 2), dtype=torch.long, device=dev)
    if len(offsets) != len(coord_list) + 1 or offsets[-1] != n:
        raise ValueError("offsets must hold the cumulative anchor counts of coord_list and end at N")
    counts = torch.tensor([offsets[p + 1] - offsets[p] for p in range(len(coord_list))], device=dev)
    patch_id = torch.repeat_interleave(torch.arange(len(coord_list), device=dev, dtype=torch.int32), counts)
    coords = torch.cat([c.to(dev).float().reshape(-1, 2) for c in coord_list], dim=0).contiguous()
    knn, mutual = ops.mutual_knn(features.float().contiguous(), patch_id.contiguous(), coords, int(k), float(pos_min_spatial))
    sel = mutual.reshape(-1).nonzero(as_tuple=True)[0]                       # the only data-dependent size, as in the reference
    if sel.numel() == 0:
        return torch.empty((0, 2), dtype=torch.long, device=dev)
    return torch.stack([sel // k, knn.reshape(-1)[sel].long()], dim=1)
