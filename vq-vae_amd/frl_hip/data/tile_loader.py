"""Real-tile input pipeline: chunk store -> pinned host batches -> device -> normalised bf16 tiles.

  ChunkTileDataset   Dataset over the `tile x tile` patches of a TileStore with the reference's sample contract
                     (dict of arrays + metadata{spatial_window, channel_names, patch_idx}; partial patches at the raster edge are
                     zero padded and masked, forest_dataset_v2.py:328-369) and `xy_by_chunk` for the chunk-locked sampler
                     (utils/samplers.py:42-108; scripts/train_vqvae.py:153-180 builds the sampler from that attribute).
  TilePrefetcher     background thread: fills pinned staging buffers (a small thread pool does the chunk -> batch copies, numpy
                     releases the GIL), uploads them on a copy stream and runs `frl_normalize_tiles` there, so that the raw
                     (float16) rows cross PCIe once and normalisation / masking / the bf16 cast cost one HBM pass instead of the
                     reference's per-channel numpy passes in DataLoader workers (feature_builder.py:402-462).  The training
                     stream only waits on an event.
"""
from __future__ import annotations

import queue
import threading
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, Iterable, Iterator, List, Optional

import numpy as np
import torch
from torch.utils.data import Dataset

from .. import _lib
from .forest_dataset import LegacySchemaMixin
from .normalization import norm_table, presets_from_meta
from .tile_store import TileStore


class ChunkTileDataset(LegacySchemaMixin, Dataset):
    def __init__(self, store: TileStore, tile: int = 32):
        t, cy, cx, f = store.chunks
        if cy % tile or cx % tile:
            raise ValueError(f"chunk size {cy}x{cx} must be a multiple of the tile size {tile}")
        self.store, self.tile = store, int(tile)
        self.time, self.features = t, f
        self.channel_names = list(store.meta["features"])
        ny, nx = store.shape[1], store.shape[2]
        self._tiles: List[tuple] = []                # (iy, ix, y0 in chunk, x0 in chunk, valid h, valid w)
        self.xy_by_chunk: List[np.ndarray] = []
        for iy in range(store.grid[0]):
            for ix in range(store.grid[1]):
                members = []
                for ty in range(cy // tile):
                    for tx in range(cx // tile):
                        gy, gx = iy * cy + ty * tile, ix * cx + tx * tile
                        if gy >= ny or gx >= nx:
                            continue                 # entirely outside the raster
                        members.append(len(self._tiles))
                        self._tiles.append((iy, ix, ty * tile, tx * tile, min(tile, ny - gy), min(tile, nx - gx)))
                self.xy_by_chunk.append(np.asarray(members, dtype=np.int64))

    def __len__(self) -> int:
        return len(self._tiles)

    def on_epoch_start(self) -> None:
        pass

    def spatial_window(self, idx: int) -> tuple:
        iy, ix, y0, x0, h, w = self._tiles[idx]
        return (iy * self.store.chunks[1] + y0, ix * self.store.chunks[2] + x0, h, w)      # (row, col, height, width)

    def tile_desc(self, idx: int) -> tuple:
        """(chunk iy, chunk ix, y0, x0, valid height, valid width) of tile `idx` inside its chunk."""
        return self._tiles[idx]

    def read_into(self, idx: int, raw_out: np.ndarray, valid_out: np.ndarray) -> None:
        """raw_out [T, tile, tile, F] (store dtype), valid_out [T, tile, tile] uint8: raw values, zero padded outside the raster."""
        iy, ix, y0, x0, h, w = self._tiles[idx]
        blk = self.store.chunk(iy, ix)
        n = self.tile
        if h == n and w == n:
            raw_out[...] = blk[:, y0:y0 + n, x0:x0 + n, :]
            valid_out[...] = 1
        else:
            raw_out[...] = 0
            raw_out[:, :h, :w, :] = blk[:, y0:y0 + h, x0:x0 + w, :]
            valid_out[...] = 0
            valid_out[:, :h, :w] = 1

    def __getitem__(self, idx: int) -> Dict:
        if idx < 0 or idx >= len(self._tiles):
            raise IndexError(idx)
        raw = np.empty((self.time, self.tile, self.tile, self.features), dtype=self.store.dtype)
        valid = np.empty((self.time, self.tile, self.tile), dtype=np.uint8)
        self.read_into(idx, raw, valid)
        return {"tile": raw, "mask": valid[0].astype(bool),
                "metadata": {"spatial_window": self.spatial_window(idx), "channel_names": self.channel_names, "patch_idx": idx}}


class _Slot:
    def __init__(self, ds: ChunkTileDataset, batch: int, device, out_dtype):
        shape = (batch, ds.time, ds.tile, ds.tile, ds.features)
        tdt = torch.float16 if ds.store.dtype == np.float16 else torch.float32
        self.raw_host = torch.empty(shape, dtype=tdt).pin_memory()
        self.valid_host = torch.empty(shape[:-1], dtype=torch.uint8).pin_memory()
        self.raw_np, self.valid_np = self.raw_host.numpy(), self.valid_host.numpy()
        self.raw_dev = torch.empty(shape, dtype=tdt, device=device)
        self.valid_dev = torch.empty(shape[:-1], dtype=torch.uint8, device=device)
        self.tile_dev = torch.empty(shape, dtype=out_dtype, device=device)
        self.mask_dev = torch.empty(shape[:-1], dtype=torch.uint8, device=device)
        self.chunk_host = self.chunk_np = self.chunk_dev = None     # whole-chunk staging, allocated on first use
        self.desc_host = torch.empty((batch, 4), dtype=torch.int32).pin_memory()
        self.desc_dev = torch.empty((batch, 4), dtype=torch.int32, device=device)
        self.ready = torch.cuda.Event()
        self.released = torch.cuda.Event()          # recorded on the consumer's stream when it moves on to the next batch
        self.used = False

    def chunk_buffers(self, ds: ChunkTileDataset, device):
        if self.chunk_host is None:
            self.chunk_host = torch.empty(ds.store.chunks, dtype=self.raw_host.dtype).pin_memory()
            self.chunk_np = self.chunk_host.numpy()
            self.chunk_dev = torch.empty(ds.store.chunks, dtype=self.raw_host.dtype, device=device)
        return self.chunk_np


class TilePrefetcher:
    """Iterates `batches` (lists of dataset indices, e.g. a ChunkBatchSampler) and yields
    {"tile": [b, T, H, W, F] out_dtype, "mask": [b, T, H, W] uint8 (1 = valid), "indices": [...]} resident on `device`.
    The tensors of a batch stay valid until the next batch is requested; device work already enqueued on the consumer's stream at
    that point is honoured (the buffers rotate behind an event)."""

    def __init__(self, dataset: ChunkTileDataset, batches: Iterable[List[int]], device="cuda", out_dtype=torch.bfloat16,
                 depth: int = 3, workers: int = 8, table: Optional[np.ndarray] = None, max_batch: Optional[int] = None,
                 whole_chunk_fraction: float = 0.5):
        _lib.load()                                  # no HIP library -> fail here, there is no host-side normalisation path
        self.ds, self.batches, self.device, self.out_dtype = dataset, batches, torch.device(device), out_dtype
        if self.device.type != "cuda":
            raise RuntimeError("TilePrefetcher normalises on the GPU; it needs a cuda device")
        self.depth, self.workers = max(2, int(depth)), max(1, int(workers))
        if table is None:
            table = norm_table(*presets_from_meta(dataset.store.meta))
        self.table = torch.from_numpy(np.ascontiguousarray(table)).to(self.device)
        self.copy_stream = torch.cuda.Stream(device=self.device)
        if max_batch is None:
            max_batch = getattr(batches, "batch_size", None)
        if max_batch is None:
            if not isinstance(batches, (list, tuple)):
                raise ValueError("max_batch is required unless `batches` is a list or has a batch_size attribute")
            max_batch = max((len(b) for b in batches), default=0)
        self.max_batch = int(max_batch)
        # a chunk-locked batch that covers at least this fraction of its chunk is uploaded as ONE contiguous chunk copy and cut
        # into tiles on the device (frl_normalize_chunk_tiles); smaller batches are gathered tile by tile on the host
        self.whole_chunk_fraction = float(whole_chunk_fraction)

    def _fill(self, slot: _Slot, indices: List[int], pool: ThreadPoolExecutor) -> None:
        ds = self.ds
        list(pool.map(lambda j: ds.read_into(indices[j], slot.raw_np[j], slot.valid_np[j]), range(len(indices))))

    def _single_chunk(self, indices: List[int]):
        """(iy, ix) when every tile of the batch lies in one chunk and the batch is large enough for the whole-chunk path."""
        ds = self.ds
        first = ds.tile_desc(indices[0])[:2]
        if any(ds.tile_desc(i)[:2] != first for i in indices):
            return None
        _, cy, cx, _ = ds.store.chunks
        return first if len(indices) * ds.tile * ds.tile >= self.whole_chunk_fraction * cy * cx else None

    def _fill_chunk(self, slot: _Slot, indices: List[int], chunk_id, pool: ThreadPoolExecutor) -> None:
        ds = self.ds
        slot.chunk_buffers(ds, self.device)
        src = ds.store.chunk(*chunk_id)
        # ONE native call (frl_host_parallel_copy: std::threads, no GIL held): the training thread needs the interpreter for every
        # kernel launch, and a Python thread pool copying slabs here slows both sides down
        if not src.flags["C_CONTIGUOUS"]:
            raise ValueError("stored chunk is not C-contiguous")
        _lib.check(_lib.load().frl_host_parallel_copy(slot.chunk_host.data_ptr(), src.ctypes.data, src.nbytes, self.workers),
                   "frl_host_parallel_copy")
        d = slot.desc_host.numpy()
        for j, i in enumerate(indices):
            d[j] = ds.tile_desc(i)[2:]

    def _producer(self, q: "queue.Queue", free: "queue.Queue", slots: List[_Slot], stop: threading.Event) -> None:
        from .. import ops
        try:
            torch.cuda.set_device(self.device)
            with ThreadPoolExecutor(self.workers) as pool:
                for indices in self.batches:
                    slot = free.get()
                    if stop.is_set() or slot is None:
                        return
                    n = len(indices)
                    if n > self.max_batch:
                        raise ValueError(f"batch of {n} tiles exceeds max_batch={self.max_batch}")
                    if slot.used:
                        slot.ready.synchronize()     # the previous upload from this slot's pinned staging buffers has finished
                    chunk_id = self._single_chunk(indices) if n else None
                    if chunk_id is not None:
                        self._fill_chunk(slot, indices, chunk_id, pool)
                    else:
                        self._fill(slot, indices, pool)
                    with torch.cuda.stream(self.copy_stream):
                        if slot.used:
                            self.copy_stream.wait_event(slot.released)     # the consumer is done with this slot's tensors
                        if chunk_id is not None:
                            slot.chunk_dev.copy_(slot.chunk_host, non_blocking=True)
                            slot.desc_dev[:n].copy_(slot.desc_host[:n], non_blocking=True)
                            ops.normalize_chunk_tiles(slot.chunk_dev, slot.desc_dev[:n], self.ds.tile, self.table, out_dtype=self.out_dtype,
                                                      out=slot.tile_dev[:n], mask_out=slot.mask_dev[:n])
                        else:
                            slot.raw_dev[:n].copy_(slot.raw_host[:n], non_blocking=True)
                            slot.valid_dev[:n].copy_(slot.valid_host[:n], non_blocking=True)
                            ops.normalize_tiles(slot.raw_dev[:n], self.table, valid=slot.valid_dev[:n], out_dtype=self.out_dtype,
                                                out=slot.tile_dev[:n], mask_out=slot.mask_dev[:n])
                        slot.ready.record(self.copy_stream)
                    slot.used = True
                    q.put((slot, indices))
            q.put(None)
        except BaseException as e:                   # surfaced in the consumer
            q.put(e)

    def __iter__(self) -> Iterator[Dict]:
        if self.max_batch <= 0:
            return
        slots = [_Slot(self.ds, self.max_batch, self.device, self.out_dtype) for _ in range(self.depth)]
        q: "queue.Queue" = queue.Queue(maxsize=self.depth)
        free: "queue.Queue" = queue.Queue()
        for s in slots[:-1]:                         # one slot is always held by the consumer
            free.put(s)
        spare = [slots[-1]]
        stop = threading.Event()
        th = threading.Thread(target=self._producer, args=(q, free, slots, stop), daemon=True)
        th.start()
        held: Optional[_Slot] = None
        try:
            while True:
                item = q.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                slot, indices = item
                cur = torch.cuda.current_stream(self.device)
                if held is not None:                 # the previous batch is released as of this point of the consumer stream
                    held.released.record(cur)
                    free.put(held)
                elif spare:
                    free.put(spare.pop())
                held = slot
                cur.wait_event(slot.ready)
                n = len(indices)
                yield {"tile": slot.tile_dev[:n], "mask": slot.mask_dev[:n], "indices": indices}
        finally:
            stop.set()
            free.put(None)
            while th.is_alive():
                try:
                    q.get(timeout=0.05)
                except queue.Empty:
                    pass
            th.join()
