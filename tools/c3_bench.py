"""A/B of the conv3x3 forward / bwd-data kernel: 16-row tiles vs 32-row tiles (frl_conv3x3_tile32), same process, interleaved rounds.
Usage: python tools/c3_bench.py  ->  JSON lines on stdout (median / min of 20 rounds per shape and setting, us)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "vq-vae_amd"))
from frl_hip import _lib, ops  # noqa: E402

dev = "cuda:0"
g = torch.Generator().manual_seed(0)
lib = _lib.load()


def timeit(fn, n=20):
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return round(ts[len(ts) // 2], 1), round(ts[0], 1)


for (cin, cout, masked) in [(64, 64, False), (128, 64, False), (64, 64, True), (64, 128, True)]:
    x = torch.randn(256, 32, 32, cin, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).to(dev)
    b = torch.randn(cout, generator=g).to(dev)
    if masked:                                   # bwd-data: dy [.., cin_of_the_forward = our cin], weights of the forward conv [cin, cout]
        wf = (torch.randn(cin, cout, 3, 3, generator=g) * 0.05).to(dev)
        y = torch.randn(256, 32, 32, cin, generator=g).to(torch.bfloat16).to(dev)
        fn = lambda: ops.conv3x3_bwd_data(x, wf, y, ops.ACT_RELU)
    else:
        fn = lambda: ops.conv3x3_fwd(x, w, b, ops.ACT_RELU)
    outs, res = {}, {}
    for _ in range(2):
        for tall in (0, 1):
            lib.frl_conv3x3_tile32(tall)
            fn()
            res.setdefault(tall, []).append(timeit(fn))
            outs[tall] = fn()
    lib.frl_conv3x3_tile32(1)
    print(json.dumps({"cin": cin, "cout": cout, "bwd_data_masked": masked, "tile16_us(median,min)": res[0], "tile32_us(median,min)": res[1],
                      "equal": bool(torch.equal(outs[0], outs[1]))}), flush=True)
