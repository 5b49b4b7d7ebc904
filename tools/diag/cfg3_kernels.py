"""Per-kernel time of one eager train step at BASELINE configs[3] (K = 8192, d = 128): where do its 30 ms go?"""
import os, sys, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vq-vae_amd"))
import bench
from frl_hip import ops
from frl_hip.data import SyntheticTileStream
sys.argv = ["bench.py"]
args = bench.parse()
dev = torch.device("cuda", 0)
model, trainer = bench.build_trainer(args, dev, torch.bfloat16, 8192, 128, 0)
stream = SyntheticTileStream(256, 5, 32, args.features, device=dev, dtype=torch.bfloat16, seed=1234)
for _ in range(3):
    trainer.step(stream.next())
torch.cuda.synchronize()
ops.kernel_timing(True); ops.kernel_timing_report()
n = 3
for _ in range(n):
    trainer.step(stream.next())
torch.cuda.synchronize()
rep = ops.kernel_timing_report(); ops.kernel_timing(False)
tot = 0
for k, (c, ms) in sorted(rep.items(), key=lambda kv: -kv[1][1]):
    tot += ms / n
    print(f"{k[:60]:62s} {c / n:6.1f} launches  {1e3 * ms / c:9.1f} us  {ms / n:8.3f} ms/step")
print("total", round(tot, 3), "ms/step")
