"""conv3x3 backward-data / backward-weight with and without the activation mask on the incoming gradient (what folding the mask into the
producer of that gradient would save): median us of 20 rounds per shape."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "vq-vae_amd"))
from frl_hip import ops  # noqa: E402

dev = "cuda:0"
g = torch.Generator().manual_seed(0)


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
    return round(ts[len(ts) // 2], 1)


for cin_f, cout_f, act in [(64, 64, ops.ACT_SIGMOID), (64, 64, ops.ACT_RELU), (128, 64, ops.ACT_RELU)]:   # forward conv cin_f -> cout_f
    dy = torch.randn(256, 32, 32, cout_f, generator=g).to(torch.bfloat16).to(dev)
    y = torch.rand(256, 32, 32, cout_f, generator=g).to(torch.bfloat16).to(dev)
    x = torch.randn(256, 32, 32, cin_f, generator=g).to(torch.bfloat16).to(dev)
    wf = (torch.randn(cout_f, cin_f, 3, 3, generator=g) * 0.05).to(dev)
    r = {}
    for name, fn in (("bwd_data masked", lambda: ops.conv3x3_bwd_data(dy, wf, y, act)), ("bwd_data plain", lambda: ops.conv3x3_bwd_data(dy, wf, None, ops.ACT_NONE)),
                     ("bwd_weight masked", lambda: ops.conv3x3_bwd_weight(dy, x, y, act)), ("bwd_weight plain", lambda: ops.conv3x3_bwd_weight(dy, x, None, ops.ACT_NONE))):
        fn()
        r[name] = timeit(fn)
    print(json.dumps({"forward": f"{cin_f}->{cout_f}", "act": act, **r}), flush=True)
