"""On-disk tile store: the reference's `attrs_raw(time, y, x, feature)` cube (utils/data_stack.py:271-309: dims and rechunking
to {time, y, x, feature} chunks; README.md:27-30) kept as one file per (y, x) chunk.

zarr is not installed in this image, so the container is a directory in Zarr-v2 style -- `meta.json` (shape, chunks, dtype,
feature names, per-feature normalisation presets and statistics) and `attrs_raw/0.<iy>.<ix>.0.npy`, one C-ordered array
`[T, cy, cx, F]` per chunk (edge chunks are stored at full chunk size, padded with NaN = no data, as Zarr pads with fill_value).
Chunks are memory-mapped; nothing is decompressed or transposed between the file and the HIP kernels: the (time, y, x, feature)
order IS the NHWC row order the kernels consume.  Values are float16 or float32; NaN marks missing observations."""
from __future__ import annotations

import json
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

FORMAT = "frl-tile-store-1"


def write_tile_store(path: str, cube: np.ndarray, chunks: Tuple[int, int], feature_names: Optional[Sequence[str]] = None,
                     normalization: Optional[Dict] = None, stats: Optional[Dict] = None, dtype: str = "float16") -> Dict:
    """cube [T, Y, X, F] -> store at `path` with (y, x) chunks `chunks`; returns the metadata dict."""
    if cube.ndim != 4:
        raise ValueError("cube must be (time, y, x, feature)")
    if dtype not in ("float16", "float32"):
        raise ValueError("dtype must be float16 or float32")
    t, ny, nx, f = cube.shape
    cy, cx = int(chunks[0]), int(chunks[1])
    names = list(feature_names) if feature_names is not None else [f"f{i:03d}" for i in range(f)]
    if len(names) != f:
        raise ValueError("one name per feature")
    os.makedirs(os.path.join(path, "attrs_raw"), exist_ok=True)
    for iy in range(-(-ny // cy)):
        for ix in range(-(-nx // cx)):
            blk = np.full((t, cy, cx, f), np.nan, dtype=dtype)
            src = cube[:, iy * cy:(iy + 1) * cy, ix * cx:(ix + 1) * cx, :]
            blk[:, :src.shape[1], :src.shape[2], :] = src
            np.save(os.path.join(path, "attrs_raw", f"0.{iy}.{ix}.0.npy"), blk)
    meta = {"format": FORMAT, "dims": ["time", "y", "x", "feature"], "shape": [t, ny, nx, f], "chunks": [t, cy, cx, f],
            "dtype": dtype, "features": names, "normalization": normalization or {}, "stats": stats or {}}
    with open(os.path.join(path, "meta.json"), "w") as fh:
        json.dump(meta, fh, indent=1)
    return meta


class TileStore:
    def __init__(self, path: str):
        self.path = path
        with open(os.path.join(path, "meta.json")) as fh:
            self.meta = json.load(fh)
        if self.meta.get("format") != FORMAT:
            raise ValueError(f"{path}: not a {FORMAT} store")
        self.shape = tuple(self.meta["shape"])
        self.chunks = tuple(self.meta["chunks"])
        self.dtype = np.dtype(self.meta["dtype"])
        self.grid = (-(-self.shape[1] // self.chunks[1]), -(-self.shape[2] // self.chunks[2]))
        self._maps: Dict[Tuple[int, int], np.ndarray] = {}

    @property
    def num_chunks(self) -> int:
        return self.grid[0] * self.grid[1]

    def chunk(self, iy: int, ix: int) -> np.ndarray:
        """Memory-mapped `[T, cy, cx, F]` block (allow_pickle stays off: the files hold plain arrays)."""
        key = (iy, ix)
        m = self._maps.get(key)
        if m is None:
            if not (0 <= iy < self.grid[0] and 0 <= ix < self.grid[1]):
                raise IndexError(key)
            m = np.load(os.path.join(self.path, "attrs_raw", f"0.{iy}.{ix}.0.npy"), mmap_mode="r")
            if m.shape != self.chunks or m.dtype != self.dtype:
                raise ValueError(f"chunk {key}: shape/dtype {m.shape}/{m.dtype} does not match the metadata")
            self._maps[key] = m
        return m
