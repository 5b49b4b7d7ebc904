"""tcn_chain_fwd_kernel: tiles handed out by the LDS counter vs the fixed per-wave sequence, same process, interleaved rounds (median us)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "vq-vae_amd"))
import torch
from frl_hip import ops, _lib
from frl_hip.models.blocks import Conv2dParams, TCNEncoder

lib = _lib.load()
torch.manual_seed(0)
tcn = TCNEncoder(64, [64, 64, 64], 3, [1, 2, 4], 0.0, 8).cuda()
head = Conv2dParams(64, 12, 1).cuda()
x = torch.randn(256, 5, 1024, 64).bfloat16().cuda()
blocks = [(l.conv.weight, l.conv.bias, l.norm.weight, l.norm.bias, l.gate.weight, l.gate.bias, l.dilation, 8, False) for l in tcn.layers]
res = {0: [], 1: []}
outs = {}
for rnd in range(6):
    for st in (0, 1):
        lib.frl_tcn_chain_static_tiles(st)
        outs[st] = ops.tcn_chain_fwd(x, blocks, head.weight, head.bias)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.tcn_chain_fwd(x, blocks, head.weight, head.bias)
        e1.record()
        torch.cuda.synchronize()
        res[st].append(e0.elapsed_time(e1) * 100)
lib.frl_tcn_chain_static_tiles(0)
print("equal outputs:", all(torch.equal(a, b) for a, b in zip(outs[0], outs[1])))
for st in (0, 1):
    r = sorted(res[st])
    print("static" if st else "dynamic", [round(v, 1) for v in r], "median", round(r[len(r) // 2], 1), "us per call (incl. weight pack)")
