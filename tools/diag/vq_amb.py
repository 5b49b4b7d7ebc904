import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vq-vae_amd"))
from frl_hip import ops
g = torch.Generator().manual_seed(0)
for (n, k, d, zs, es) in ((262144, 8192, 128, 1.0, 1.0), (262144, 8192, 128, 1.0, 0.1), (262144, 512, 64, 1.0, 1.0), (65536, 1024, 64, 1.0, 1.0)):
    z = (torch.randn(n, d, generator=g) * zs).to(torch.bfloat16).cuda()
    e = (torch.randn(k, d, generator=g) * es).cuda()
    idx, zq, stats, counts = ops.vq_assign(z, e)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
    for a, b in ev:
        a.record(); ops.vq_assign(z, e); b.record()
    torch.cuda.synchronize()
    print(n, k, d, zs, es, "re-evaluated rows", int(stats[2].item()), "=", round(100 * stats[2].item() / n, 2), "%  call us", round(min(a.elapsed_time(b) for a, b in ev) * 1e3, 1))
