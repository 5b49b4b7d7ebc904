"""Diagnostic: conv1x1 weight-gradient / forward call time with warm (Infinity-Cache resident) and cold (flushed) inputs."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "vq-vae_amd"))
from frl_hip import ops
dev = "cuda:0"
flush = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
def timed(fn, cold):
    ts = []
    for _ in range(6):
        if cold: flush.fill_(1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]
P = 262144
for (ci, co) in ((64, 64), (128, 64), (64, 256)):
    dy = torch.randn(P, co, device=dev).bfloat16(); x = torch.randn(P, ci, device=dev).bfloat16()
    w = torch.randn(co, ci, device=dev)
    for name, fn in (("wgrad", lambda: ops.conv1x1_bwd_weight(dy, x, None, 0)), ("fwd", lambda: ops.conv1x1_fwd(x, w, None, 0))):
        fn()
        print(f"{name} {ci}->{co}: warm {timed(fn, False):.1f} us, cold {timed(fn, True):.1f} us")
