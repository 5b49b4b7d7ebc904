"""A/B of the graphed configs[1] train step in ONE process on ONE box (box-to-box spread is larger than most single changes):
   python tools/diag/ab_step.py spatial_fuse      -> EdgeAwareSmoothingConv2D.fuse False vs True
   python tools/diag/ab_step.py encoder_fuse      -> Conv2DEncoder.fuse False vs True"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vq-vae_amd"))
import bench
from frl_hip.data import SyntheticTileStream
from frl_hip.models import blocks

what = sys.argv[1] if len(sys.argv) > 1 else "spatial_fuse"
cls = {"spatial_fuse": blocks.EdgeAwareSmoothingConv2D, "encoder_fuse": blocks.Conv2DEncoder}[what]
sys.argv = ["bench.py"]
args = bench.parse()
dev = torch.device("cuda", 0)
stream = SyntheticTileStream(256, 5, 32, args.features, device=dev, dtype=torch.bfloat16, seed=1234)
runs = {}
for flag in (False, True):
    cls.fuse = flag
    model, trainer = bench.build_trainer(args, dev, torch.bfloat16, 512, 64, 0)
    if os.environ.get("AB_SERIAL"):
        model.concurrent_phase = False
    step = trainer.step_graphed if trainer.graph_supported() else trainer.step
    for _ in range(8):
        step(stream.next())
    runs[flag] = step
res = {False: [], True: []}
for rep in range(6):
    for flag in (False, True):
        cls.fuse = flag
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(40):
            runs[flag](stream.next())
        torch.cuda.synchronize()
        res[flag].append((time.perf_counter() - t0) / 40 * 1e3)
for flag in (False, True):
    v = sorted(res[flag])
    print(what, flag, "ms/step min %.4f median %.4f" % (v[0], v[len(v) // 2]))
