"""Per-kernel A/B of the backward epilogue sums at configs[1] shapes (one process, one box)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vq-vae_amd"))
from frl_hip import ops
from frl_hip.ops import ACT_NONE, ACT_RELU
dev = "cuda:0"
bf = torch.bfloat16
def rnd(*s): return torch.randn(*s, device=dev).to(bf)
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
B, H, W = 256, 32, 32
w = torch.randn(64, 64, 3, 3, device=dev) * 0.1
dy, y, add, sub = rnd(B, H, W, 64), rnd(B, H, W, 64), rnd(B, H, W, 64), rnd(B, H, W, 64)
print("c3 bwd-data plain             %.1f us" % timeit(lambda: ops.conv3x3_bwd_data(dy, w, y, ACT_RELU)))
print("c3 bwd-data plain + 2 adds    %.1f us" % timeit(lambda: ops.add(sub, ops.conv3x3_bwd_data(dy, w, y, ACT_RELU) + add, -1.0)))
print("c3 bwd-data add epilogue      %.1f us" % timeit(lambda: ops.conv3x3_bwd_data(dy, w, y, ACT_RELU, add=add)))
print("c3 bwd-data add+sub epilogue  %.1f us" % timeit(lambda: ops.conv3x3_bwd_data(dy, w, y, ACT_RELU, add=add, sub_from=sub)))
P = B * H * W
wb = torch.randn(256, 64, device=dev) * 0.1
db, addf = rnd(P, 256), rnd(P, 64)
print("pw bwd-data 256->64 plain     %.1f us" % timeit(lambda: ops.conv1x1_bwd_data(db, wb, None, ACT_NONE)))
print("pw bwd-data plain + add       %.1f us" % timeit(lambda: ops.conv1x1_bwd_data(db, wb, None, ACT_NONE) + addf))
print("pw bwd-data add epilogue      %.1f us" % timeit(lambda: ops.conv1x1_bwd_data(db, wb, None, ACT_NONE, add=addf)))
dg, addx = rnd(B, H, W, 128), rnd(B, H, W, 64)
print("sobel_bwd plain               %.1f us" % timeit(lambda: ops.sobel_bwd(dg)))
print("sobel_bwd plain + add         %.1f us" % timeit(lambda: ops.sobel_bwd(dg) + addx))
print("sobel_bwd add epilogue        %.1f us" % timeit(lambda: ops.sobel_bwd(dg, add=addx)))
