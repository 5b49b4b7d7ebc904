#!/usr/bin/env python3
"""HIP-event timing of the weight-gradient calls of one BASELINE configs[1] train step, shape by shape: the kernel itself (event pair
inside the library) and the whole C-ABI call (+ slab reduction).  Usage: python tools/wgrad_bench.py [--reps 20]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vq-vae_amd"))
import torch  # noqa: E402

# (name, rows, Cin, Cout, act) of the 1x1 convolutions, (name, B, H, W, Cin, Cout, act) of the 3x3 ones
PW = [("encoder conv1 64->128", 262144, 64, 128, 0), ("encoder conv2 128->64", 262144, 128, 64, 0), ("mix_head_A 64->32", 262144, 64, 32, 0),
      ("mix_head_B 64->256", 262144, 64, 256, 0), ("FiLM l1 64->32 relu", 262144, 64, 32, 1), ("FiLM l2 32->12", 262144, 32, 12, 0),
      ("phase_head 64->12 (rows x5)", 1310720, 64, 12, 0)]
C3 = [("mix_backbone 128->64 relu", 256, 32, 32, 128, 64, 1), ("gate_net.0 64->64 relu", 256, 32, 32, 64, 64, 1), ("gate_net.2 64->64 sigmoid", 256, 32, 32, 64, 64, 2)]


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for e0, e1 in ev:
        e0.record()
        fn()
        e1.record()
    torch.cuda.synchronize()
    ts = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in ev)
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--max-wgs", type=int, default=0, help="frl_wgrad_set_max_workgroups before the runs (0: library default)")
    a = ap.parse_args()
    from frl_hip import ops, _lib
    if a.max_wgs:
        _lib.load().frl_wgrad_set_max_workgroups(a.max_wgs)
    dev = "cuda:0"
    g = torch.Generator().manual_seed(0)
    out = {}
    for name, p, cin, cout, act in PW:
        x = torch.randn(p, cin, generator=g).to(torch.bfloat16).to(dev)
        dy = torch.randn(p, cout, generator=g).to(torch.bfloat16).to(dev)
        y = torch.randn(p, cout, generator=g).to(torch.bfloat16).to(dev) if act else None
        fn = lambda: ops.conv1x1_bwd_weight(dy, x, y, act, want_bias=True)  # noqa: E731
        call = timed(fn, a.reps)
        ops.kernel_timing(True)
        ops.kernel_timing_report()
        for _ in range(a.reps):
            fn()
        rep = ops.kernel_timing_report()
        ops.kernel_timing(False)
        nb = p * (cin + cout * (2 if act else 1)) * 2
        out[name] = {"call_us": round(call, 1), "bytes_MB": round(nb / 1e6, 1), "hbm_floor_us": round(nb / 8e6, 1),
                     "kernels_us": {k[:40]: round(1e3 * ms / c, 1) for k, (c, ms) in rep.items()}}
        del x, dy, y
    for name, b, h, w, cin, cout, act in C3:
        x = torch.randn(b, h, w, cin, generator=g).to(torch.bfloat16).to(dev)
        dy = torch.randn(b, h, w, cout, generator=g).to(torch.bfloat16).to(dev)
        y = torch.randn(b, h, w, cout, generator=g).to(torch.bfloat16).to(dev) if act else None
        fn = lambda: ops.conv3x3_bwd_weight(dy, x, y, act)  # noqa: E731
        call = timed(fn, a.reps)
        ops.kernel_timing(True)
        ops.kernel_timing_report()
        for _ in range(a.reps):
            fn()
        rep = ops.kernel_timing_report()
        ops.kernel_timing(False)
        nb = b * h * w * (cin + cout * (2 if act else 1)) * 2
        out[name] = {"call_us": round(call, 1), "bytes_MB": round(nb / 1e6, 1), "hbm_floor_us": round(nb / 8e6, 1),
                     "GFLOP": round(2 * 9 * cin * cout * b * h * w / 1e9, 1), "kernels_us": {k[:40]: round(1e3 * ms / c, 1) for k, (c, ms) in rep.items()}}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
