"""dec_mse_bwd (12-channel latents, 1.31 M rows): two 4-wave subgroups per workgroup vs the lockstep 8-wave workgroup, same process, interleaved."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "vq-vae_amd"))
import torch
from frl_hip import ops, _lib

lib = _lib.load()
g = torch.Generator().manual_seed(0)
P, cz = 256 * 5 * 1024, 12
z = torch.randn(P, cz, generator=g).bfloat16().cuda()
tgt = torch.randn(P, 64, generator=g).bfloat16().cuda()
w1 = (torch.randn(128, cz, generator=g) / cz ** 0.5).cuda(); b1 = torch.zeros(128).cuda()
w2 = (torch.randn(64, 128, generator=g) / 128 ** 0.5).cuda(); b2 = torch.zeros(64).cuda()
stats, _ = ops.decoder_mse_fwd(z, w1, b1, w2, b2, tgt, None)
gs = torch.ones(2, device="cuda")
res = {0: [], 1: []}
for rnd in range(5):
    for on in (1, 0):
        lib.frl_decoder_mse_bwd_subgroups(on)
        ops.decoder_mse_bwd(z, w1, b1, w2, b2, tgt, None, gs, stats)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            ops.decoder_mse_bwd(z, w1, b1, w2, b2, tgt, None, gs, stats)
        e1.record()
        torch.cuda.synchronize()
        res[on].append(e0.elapsed_time(e1) * 125)
lib.frl_decoder_mse_bwd_subgroups(1)
for on in (1, 0):
    r = sorted(res[on])
    print("subgroups" if on else "lockstep ", [round(v, 1) for v in r], "median", round(r[len(r) // 2], 1), "us per call (incl. pack + slab reduce)")
