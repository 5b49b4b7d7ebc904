"""Per-kernel time of one eager train step at a secondary BASELINE configuration: python tools/diag/cfg3_kernels.py [3|4]
(configs[3]: K = 8192, d = 128; configs[4] on one GPU: T = 10, 64 x 64 tiles, two codebooks of 1024)."""
import os, sys, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vq-vae_amd"))
import bench
from frl_hip import ops
from frl_hip.data import SyntheticTileStream
sys.argv_saved = list(sys.argv)
sys.argv = ["bench.py"]
args = bench.parse()
dev = torch.device("cuda", 0)
which = sys.argv_saved[1] if len(sys.argv_saved) > 1 else "3"
if which == "4":
    model, trainer = bench.build_trainer(args, dev, torch.bfloat16, 1024, 64, 1024)
    stream = SyntheticTileStream(32, 10, 64, args.features, device=dev, dtype=torch.bfloat16, seed=1234)
else:
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
