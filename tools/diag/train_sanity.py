"""300 graphed train steps at BASELINE configs[1] on a fixed pool of synthetic tiles: the loss must stay finite and fall."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vq-vae_amd"))
import bench
from frl_hip.data import SyntheticTileStream
sys.argv = ["bench.py"]
args = bench.parse()
dev = torch.device("cuda", 0)
DT = torch.float32 if os.environ.get("SANITY_F32") else torch.bfloat16
NB = int(os.environ.get("SANITY_BATCH", "256"))
model, trainer = bench.build_trainer(args, dev, DT, 512, 64, 0)
stream = SyntheticTileStream(NB, 5, 32, args.features, device=dev, dtype=DT, seed=1234)
step = trainer.step_graphed if trainer.graph_supported() else trainer.step
losses = []
for i in range(300):
    out = step(stream.next())
    if i % 25 == 0 or i == 299:
        losses.append((i, round(float(out["loss"].detach()), 4), round(float(out["perplexity"]), 1)))
print(losses)
print("skipped steps:", trainer.skipped if hasattr(trainer, "skipped") else None)
assert all(l == l and abs(l) < 1e6 for _, l, _ in losses) and losses[-1][1] < losses[0][1]
print("train sanity ok")
