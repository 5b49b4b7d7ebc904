#!/usr/bin/env python3
"""HIP-event timing of the hot TCN block kernels at BASELINE configs[1] (256 tiles of 5x32x32x64): per launch, per dilation.
Usage: python tools/tcn_bwd_bench.py [--reps 30] [--old]   (--old: route the backward through the 8-wave mask/ragged kernel)"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vq-vae_amd"))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--old", action="store_true")
    a = ap.parse_args()
    from frl_hip import ops, _lib
    dev = "cuda:0"
    B, T, HW, C = a.batch, 5, 1024, 64
    g = torch.Generator().manual_seed(0)
    w = [torch.randn(C, C, 3, generator=g) / (3 * C) ** 0.5, torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5,
         torch.randn(C, generator=g) * 0.2, torch.randn(C, C, 1, generator=g) / C ** 0.5, torch.randn(C, generator=g) * 0.1]
    args = tuple(t.to(dev) for t in w) + (None, None)
    x = torch.randn(B, T, HW, C, generator=g).to(torch.bfloat16).to(dev)
    dy = torch.randn(B, T, HW, C, generator=g).to(torch.bfloat16).to(dev)
    if a.old:
        _lib.load().frl_tcn_hot_force_generic_tiles(1)
    out = {}
    rows = B * HW * T
    for dil in (1, 2, 4):
        for name, fn in (("fwd", lambda: ops.tcn_block_fwd(x, *args, dil, 8)), ("bwd", lambda: ops.tcn_block_bwd(x, dy, *args, dil, 8))):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.reps)]
            for e0, e1 in ev:
                e0.record()
                fn()
                e1.record()
            torch.cuda.synchronize()
            ts = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in ev)
            med = ts[len(ts) // 2]
            nb = rows * 128 * (2 if name == "fwd" else 3)
            out[f"{name}_dil{dil}"] = {"us_median": round(med, 1), "us_min": round(ts[0], 1), "GB/s": round(nb / med / 1e3, 1),
                                       "hbm_frac": round(nb / med / 1e3 / 8000.0, 4)}
    print(json.dumps({"kernel": "old 8-wave" if a.old else "dispatch", "whole_call_incl_pack_and_slab_reduce": out}, indent=1))


if __name__ == "__main__":
    main()
