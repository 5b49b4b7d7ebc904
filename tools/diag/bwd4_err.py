"""tcn_hot_bwd4 against tcn_hot_bwd3 and the round-1 8-wave kernel on a multi-tile case: max / mean differences and launch time."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "vq-vae_amd"))
import torch
from frl_hip import ops, _lib

lib = _lib.load()
B, HW, dil = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (21, 1024, 1)
g = torch.Generator().manual_seed(1)
w = dict(conv_w=torch.randn(64, 64, 3, generator=g) / 192 ** 0.5, conv_b=torch.randn(64, generator=g) * 0.1, gn_w=torch.rand(64, generator=g) + 0.5,
         gn_b=torch.randn(64, generator=g) * 0.2, gate_w=torch.randn(64, 64, 1, generator=g) / 8, gate_b=torch.randn(64, generator=g) * 0.1)
args = tuple(w[k].cuda() for k in ("conv_w", "conv_b", "gn_w", "gn_b", "gate_w", "gate_b")) + (None, None)
x = torch.randn(B, 5, HW, 64, generator=g).bfloat16().cuda()
dy = torch.randn(B, 5, HW, 64, generator=g).bfloat16().cuda()
out = {}
for v in (4, 3):
    lib.frl_tcn_hot_bwd_variant(v)
    out[v] = ops.tcn_block_bwd(x, dy, *args, dil, 8)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.tcn_block_bwd(x, dy, *args, dil, 8)
    e1.record()
    torch.cuda.synchronize()
    print(f"variant {v}: {e0.elapsed_time(e1) * 100:.1f} us per call (pack + kernel + slab reduce)")
lib.frl_tcn_hot_bwd_variant(4)
lib.frl_tcn_hot_force_generic_tiles(1)
out[2] = ops.tcn_block_bwd(x, dy, *args, dil, 8)
lib.frl_tcn_hot_force_generic_tiles(0)
for a, b in ((4, 3), (4, 2), (3, 2)):
    for k in out[a]:
        d = (out[a][k].double() - out[b][k].double()).abs()
        ref = out[b][k].double().abs()
        print(f"{a} vs {b} {k:8s} max {d.max().item() / ref.max().item():.2e}  mean {d.mean().item() / ref.mean().item():.2e}  "
              f"differing {float((d > 0).double().mean()):.4f}")
