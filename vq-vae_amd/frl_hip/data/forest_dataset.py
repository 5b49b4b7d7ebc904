"""ForestDataset: (time, y, x, feature) tiles with the reference's Dataset / collate contract.

Reference contract (frl/data/loaders/dataset/forest_dataset_v2.py:328-477,745-796): `__getitem__` returns a dict of numpy
group arrays plus `metadata{spatial_window, channel_names, patch_idx}`; `collate_fn` stacks groups into [B, ...] tensors and
keeps `metadata` as a list.  The Zarr cube layout is `attrs_raw(time, y, x, feature)` with chunks (5, 32, 32, 64)
(utils/data_stack.py:271-309, README.md:27-30) -- i.e. one chunk IS one tile, already NHWC, so the tile is handed to the
HIP kernels without any transpose.  zarr is not installed in this image and the benchmark is synthetic,
so tiles are generated: `randn` (already z-scored features), all-valid mask.  With `channels_first=True` the dataset also
emits the reference's `[C,H,W]` / `[C,T,H,W]` group views for code written against ForestDatasetV2.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch
from torch.utils.data import Dataset


class LegacySchemaMixin:
    """Attributes the legacy VQ-VAE trainer reads off its dataset (scripts/train_vqvae.py:153-180,217): `cont_names` (the continuous
    features -> `cont_dim`), `cat_names` / `schema_cat` (categorical inputs and their vocabularies), `naip` (image patch array whose
    last axis is the band count) and `class_weights_by_cat_name(name)`.  A (time, y, x, feature) tile store holds continuous features
    only: no categorical inputs, no NAIP bands -- so the script's own loops build an empty `cat_vocab_sizes` and `naip_bands = 0`,
    which is what `VQVAE(cont_dim=..., cat_vocab_sizes={}, naip_bands=0, ...)` accepts.  Needs `self.channel_names`."""

    @property
    def cont_names(self) -> List[str]:
        return list(self.channel_names)

    @property
    def cat_names(self) -> List[str]:
        return []

    @property
    def schema_cat(self) -> Dict[str, dict]:
        return {}

    @property
    def naip(self) -> np.ndarray:
        return np.zeros((0, 0, 0), dtype=np.float32)                # (krow, kcol, band): no bands

    def class_weights_by_cat_name(self, name: str):
        raise KeyError(f"no categorical input {name!r}: the tile dataset has none (cat_names is empty)")


class ForestDataset(LegacySchemaMixin, Dataset):
    def __init__(self, num_tiles: int = 1024, time: int = 5, size: int = 32, features: int = 64, seed: int = 1234,
                 channels_first: bool = False, partial_edge: Optional[int] = None, tiles_per_chunk: int = 64):
        self.num_tiles, self.time, self.size, self.features = num_tiles, time, size, features
        self.seed, self.channels_first, self.partial_edge = seed, channels_first, partial_edge
        self.channel_names = [f"f{i:03d}" for i in range(features)]
        # chunk membership for the chunk-locked batch sampler (train_vqvae.py:153-159): consecutive tiles share a synthetic chunk
        self.xy_by_chunk = [np.arange(lo, min(lo + tiles_per_chunk, num_tiles), dtype=np.int64)
                            for lo in range(0, num_tiles, max(int(tiles_per_chunk), 1))]

    def __len__(self) -> int:
        return self.num_tiles

    def on_epoch_start(self) -> None:  # parity with ForestDatasetV2.on_epoch_start (train_representation.py:529)
        pass

    def __getitem__(self, idx: int) -> Dict:
        if idx < 0 or idx >= self.num_tiles:
            raise IndexError(idx)
        rng = np.random.default_rng(self.seed + idx)
        tile = rng.standard_normal((self.time, self.size, self.size, self.features), dtype=np.float32)
        mask = np.ones((self.size, self.size), dtype=bool)
        if self.partial_edge is not None and idx % 7 == 0:
            # partial patches at the raster edge are zero padded and masked out (forest_dataset_v2.py:357-369)
            tile[:, self.partial_edge:, :, :] = 0.0
            mask[self.partial_edge:, :] = False
        sample = {"tile": tile, "mask": mask,
                  "metadata": {"spatial_window": (idx * self.size, 0, self.size, self.size),
                               "channel_names": self.channel_names, "patch_idx": idx}}
        if self.channels_first:
            sample["static"] = np.ascontiguousarray(tile.mean(0).transpose(2, 0, 1))        # [C,H,W]
            sample["annual"] = np.ascontiguousarray(tile.transpose(3, 0, 1, 2))             # [C,T,H,W]
        return sample


def collate_fn(batch: List[Dict]) -> Dict:
    """Stacks array groups to [B, ...] tensors, keeps 'metadata' as a ragged list (forest_dataset_v2.py:745-796)."""
    out: Dict = {"metadata": [b["metadata"] for b in batch]}
    for k in batch[0]:
        if k == "metadata":
            continue
        out[k] = torch.from_numpy(np.stack([b[k] for b in batch]))
    return out


class SyntheticTileStream:
    """On-device tile generator for benchmarks: randn(B,T,H,W,F) from a per-rank seed (seed 1234 + rank)."""

    def __init__(self, batch: int, time: int = 5, size: int = 32, features: int = 64, device="cuda", dtype=torch.bfloat16,
                 seed: int = 1234, pool: int = 4):
        g = torch.Generator(device=device).manual_seed(seed)
        self.tiles = [torch.randn(batch, time, size, size, features, generator=g, device=device, dtype=torch.float32).to(dtype)
                      for _ in range(pool)]
        self.i = 0

    def next(self) -> torch.Tensor:
        t = self.tiles[self.i % len(self.tiles)]
        self.i += 1
        return t
