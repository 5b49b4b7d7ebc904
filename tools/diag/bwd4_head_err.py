"""tcn_hot_bwd4 HEAD variant (dy = dh W_h formed in the kernel) against the head's bwd-data launch + the plain kernel: differences and time."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "vq-vae_amd"))
import torch
from frl_hip import ops

B, HW, ch = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (2, 1024, 12)
g = torch.Generator().manual_seed(1)
w = dict(conv_w=torch.randn(64, 64, 3, generator=g) / 192 ** 0.5, conv_b=torch.randn(64, generator=g) * 0.1, gn_w=torch.rand(64, generator=g) + 0.5,
         gn_b=torch.randn(64, generator=g) * 0.2, gate_w=torch.randn(64, 64, 1, generator=g) / 8, gate_b=torch.randn(64, generator=g) * 0.1)
args = tuple(w[k].cuda() for k in ("conv_w", "conv_b", "gn_w", "gn_b", "gate_w", "gate_b"))
x = torch.randn(B, 5, HW, 64, generator=g).bfloat16().cuda()
dh = torch.randn(B, 5, HW, ch, generator=g).bfloat16().cuda()
hw_ = (torch.randn(ch, 64, generator=g) / 8).cuda()
print("supported", ops.tcn_block_bwd_head_supported(x, dh, hw_, 4))
a = ops.tcn_block_bwd_head(x, dh, hw_, *args, 4)
dy = ops.conv1x1_bwd_data(dh, hw_, None, ops.ACT_NONE)
b = ops.tcn_block_bwd(x, dy, *args, None, None, 4, 8)
dy32 = (dh.float() @ hw_).float()
print("dy check", (dy.float() - dy32).abs().max().item(), dy32.abs().max().item())
for k in a:
    d = (a[k].double() - b[k].double()).abs()
    ref = b[k].double().abs()
    print(f"{k:8s} max {d.max().item() / ref.max().item():.2e}  mean {d.mean().item() / ref.mean().item():.2e}")
for name, fn in (("head", lambda: ops.tcn_block_bwd_head(x, dh, hw_, *args, 4)), ("plain", lambda: ops.tcn_block_bwd(x, dy, *args, None, None, 4, 8))):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(name, f"{e0.elapsed_time(e1) * 100:.1f} us per call")
