#!/usr/bin/env python3
"""Measures the tile ingest on the GPU box:  python3 tools/loader_bench.py [--tiles 1024] [--batch 256]
  1. `frl_normalize_tiles` alone (f16 rows -> bf16 rows + mask bytes, resident in HBM): us / launch and GB/s of its algorithmic bytes
     (F*(2+2)+2 bytes per (t,y,x) row),
  2. store -> pinned staging -> PCIe -> normalise, no training (tiles/s of the input pipeline by itself),
  3. the full train step fed from the prefetcher vs fed from resident synthetic tiles (tiles/s).
Prints one JSON object.  The store is synthetic (randn cube, 2 % missing observations) written to a temporary directory."""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vq-vae_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--workers", type=int, default=16)       # the CPU share of one GPU on the box
    ap.add_argument("--epochs", type=int, default=12)
    a = ap.parse_args()
    from frl_hip import ops
    from frl_hip.data import ChunkBatchSampler, ChunkTileDataset, TilePrefetcher, TileStore, write_tile_store
    from frl_hip.data.normalization import norm_table, presets_from_meta
    from frl_hip.models import VQVAE
    from frl_hip.training.trainer import VQVAETrainer
    dev = torch.device("cuda:0")
    res = {"tiles_in_store": a.tiles, "batch": a.batch, "workers": a.workers, "host_cores": os.cpu_count()}

    # ---- store: chunks of 512 x 512 px = 256 tiles each (utils/data_stack.py:298-303 default y/x chunking)
    nchunk = max(1, a.tiles // 256)
    rng = np.random.default_rng(0)
    root = tempfile.mkdtemp(prefix="frl_store_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        names = [f"f{i:03d}" for i in range(64)]
        cube = rng.standard_normal((5, 512, 512 * nchunk, 64), dtype=np.float32)
        cube[rng.random(cube.shape[:3]) < 0.02] = np.nan
        write_tile_store(root, cube, (512, 512), names, normalization={n: {"type": "zscore"} for n in names},
                         stats={n: {"mean": 0.1, "sd": 1.3} for n in names}, dtype="float16")
        del cube
        st = TileStore(root)
        ds = ChunkTileDataset(st, 32)
        res["store_bytes"] = sum(os.path.getsize(os.path.join(root, "attrs_raw", f)) for f in os.listdir(os.path.join(root, "attrs_raw")))

        # ---- 1. the kernel by itself
        table = torch.from_numpy(norm_table(*presets_from_meta(st.meta))).to(dev)
        raw = torch.randn(a.batch, 5, 32, 32, 64, device=dev).half()
        valid = torch.ones(a.batch, 5, 32, 32, dtype=torch.uint8, device=dev)
        out = torch.empty(raw.shape, dtype=torch.bfloat16, device=dev)
        mk = torch.empty(valid.shape, dtype=torch.uint8, device=dev)
        for _ in range(5):
            ops.normalize_tiles(raw, table, valid=valid, out=out, mask_out=mk)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            ops.normalize_tiles(raw, table, valid=valid, out=out, mask_out=mk)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 50
        rows = raw.numel() // 64
        res["normalize_kernel"] = {"us_per_launch": round(us, 2), "rows": rows, "algorithmic_bytes": rows * (64 * 4 + 2),
                                   "GB_per_s": round(rows * (64 * 4 + 2) / us / 1e3, 1), "frac_of_8TBs": round(rows * (64 * 4 + 2) / us / 1e3 / 8000, 3)}

        # ---- 2. input pipeline alone
        def epoch_batches(seed):
            np.random.seed(seed)
            return list(ChunkBatchSampler(ds.xy_by_chunk, a.batch, drop_last=True, seed=seed))
        batches = [b for ep in range(a.epochs) for b in epoch_batches(ep)]
        n = 0
        it = iter(TilePrefetcher(ds, batches, device=dev, workers=a.workers, max_batch=a.batch))
        next(it)                                   # thread start-up, first page-ins
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for item in it:
            n += item["tile"].shape[0]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res["pipeline_only"] = {"tiles_per_s": round(n / dt, 1), "GB_per_s_raw_f16": round(n * 5 * 32 * 32 * 64 * 2 / dt / 1e9, 2), "batches": n // a.batch}

        # ---- 3. training fed by the prefetcher vs resident synthetic tiles
        torch.manual_seed(0)
        model = VQVAE(in_features=64, codebook_size=512, emb_dim=64, type_encoder_dropout=0.0, phase_tcn_dropout=0.0,
                      compute_dtype=torch.bfloat16).to(dev)     # bench.py configuration (cfg2)
        with torch.no_grad():                       # bench.py's well-separated codebook
            model.quant.codebook.copy_(torch.randn(512, 64, generator=torch.Generator().manual_seed(7)))
        tr = VQVAETrainer(model, lr=1e-4, total_steps=10_000)
        syn = torch.randn(a.batch, 5, 32, 32, 64, device=dev).bfloat16()
        for _ in range(15):
            tr.step(syn)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(40):
            tr.step(syn)
        torch.cuda.synchronize()
        res["train_resident_synthetic"] = {"tiles_per_s": round(40 * a.batch / (time.perf_counter() - t0), 1)}
        it = iter(TilePrefetcher(ds, batches, device=dev, workers=a.workers, max_batch=a.batch))
        item = next(it)
        tr.step(item["tile"], mask=item["mask"])
        torch.cuda.synchronize()
        n, t0 = 0, time.perf_counter()
        for item in it:
            tr.step(item["tile"], mask=item["mask"])
            n += item["tile"].shape[0]
        torch.cuda.synchronize()
        res["train_from_store"] = {"tiles_per_s": round(n / (time.perf_counter() - t0), 1), "skipped_steps": int(tr.n_skipped)}
    finally:
        shutil.rmtree(root, ignore_errors=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
