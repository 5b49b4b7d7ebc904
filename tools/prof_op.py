"""Runs one hot-path op in isolation (for rocprofv3 --pmc / --kernel-trace passes).  usage: prof_op.py <op> [iters]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vq-vae_amd"))
import torch  # noqa: E402
from frl_hip import ops  # noqa: E402

op = sys.argv[1]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
B, T, HW, C = 256, 5, 1024, 64
bf = torch.bfloat16


def rnd(*s, dtype=bf, scale=1.0):
    return (torch.randn(*s, generator=g, device=dev) * scale).to(dtype)


if op in ("tcn_bwd", "tcn_fwd"):
    x, dy = rnd(B, T, HW, C), rnd(B, T, HW, C)
    cw, cb = rnd(64, 64, 3, dtype=torch.float32, scale=0.07), rnd(64, dtype=torch.float32, scale=0.1)
    gw, gb = torch.ones(64, device=dev), torch.zeros(64, device=dev)
    gtw, gtb = rnd(64, 64, 1, dtype=torch.float32, scale=0.12), rnd(64, dtype=torch.float32, scale=0.1)
    for i in range(iters):
        if op == "tcn_bwd":
            ops.tcn_block_bwd(x, dy, cw, cb, gw, gb, gtw, gtb, None, None, 2, 8)
        else:
            ops.tcn_block_fwd(x, cw, cb, gw, gb, gtw, gtb, None, None, 2, 8)
elif op == "vq":
    z = rnd(B * HW, 64)
    e = rnd(512, 64, dtype=torch.float32)
    for i in range(iters):
        ops.vq_assign(z, e)
elif op == "conv3x3":
    x = rnd(B, 32, 32, 128)
    w, b_ = rnd(64, 128, 3, 3, dtype=torch.float32, scale=0.03), rnd(64, dtype=torch.float32)
    for i in range(iters):
        ops.conv3x3_fwd(x, w, b_, 1)
elif op == "smooth_bwd":
    x, ds = rnd(B, 32, 32, 64), rnd(B, 32, 32, 64)
    a, bb = torch.softmax(rnd(B, 32, 32, 8, 4, dtype=torch.float32), 3).reshape(B, 32, 32, 32).to(bf), torch.softmax(rnd(B, 32, 32, 64, 4, dtype=torch.float32), 4).reshape(B, 32, 32, 256).to(bf)
    for i in range(iters):
        ops.edge_smooth_bwd(ds, x, a, bb, 4, 3)
torch.cuda.synchronize()
print("done", op)
