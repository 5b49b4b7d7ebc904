"""tcn_hot_bwd4: subgroup-0 tile share per variant (with dx / without dx / head), same process, interleaved rounds: median us per call."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "vq-vae_amd"))
import torch
from frl_hip import ops, _lib

lib = _lib.load()
g = torch.Generator().manual_seed(1)
w = dict(conv_w=torch.randn(64, 64, 3, generator=g) / 192 ** 0.5, conv_b=torch.randn(64, generator=g) * 0.1, gn_w=torch.rand(64, generator=g) + 0.5,
         gn_b=torch.randn(64, generator=g) * 0.2, gate_w=torch.randn(64, 64, 1, generator=g) / 8, gate_b=torch.randn(64, generator=g) * 0.1)
args = tuple(w[k].cuda() for k in ("conv_w", "conv_b", "gn_w", "gn_b", "gate_w", "gate_b"))
x = torch.randn(256, 5, 1024, 64, generator=g).bfloat16().cuda()
dy = torch.randn(256, 5, 1024, 64, generator=g).bfloat16().cuda()
dh = torch.randn(256, 5, 1024, 12, generator=g).bfloat16().cuda()
hw_ = (torch.randn(12, 64, generator=g) / 8).cuda()
calls = {0: lambda: ops.tcn_block_bwd(x, dy, *args, None, None, 2, 8), 1: lambda: ops.tcn_block_bwd(x, dy, *args, None, None, 1, 8, want_dx=False),
         2: lambda: ops.tcn_block_bwd_head(x, dh, hw_, *args, 4)}
for var, fn in calls.items():
    res = {}
    for rnd in range(4):
        for share in (17, 18, 19, 20, 21):
            lib.frl_tcn_hot_bwd4_share(var, share)
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(share, []).append(e0.elapsed_time(e1) * 125)
    lib.frl_tcn_hot_bwd4_share(var, 19)
    print(("with dx", "without dx", "head")[var], {k: round(sorted(v)[len(v) // 2], 1) for k, v in res.items()})
