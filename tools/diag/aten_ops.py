"""Lists the ATen ops (tiny torch kernels around the HIP calls) of one train step: python3 tools/diag/aten_ops.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vq-vae_amd"))
from frl_hip.models import VQVAE  # noqa: E402
from frl_hip.training.trainer import VQVAETrainer  # noqa: E402

dev = "cuda:0"
torch.manual_seed(0)
m = VQVAE(in_features=64, codebook_size=512, emb_dim=64, type_encoder_dropout=0.0, phase_tcn_dropout=0.0, compute_dtype=torch.bfloat16).to(dev)
with torch.no_grad():
    m.quant.codebook.copy_(torch.randn(512, 64, generator=torch.Generator().manual_seed(7)))
tr = VQVAETrainer(m, lr=1e-4, total_steps=1000)
x = torch.randn(256, 5, 32, 32, 64, device=dev).bfloat16()
for _ in range(5):
    tr.step(x)
torch.cuda.synchronize()
steps = 4
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU], with_stack=True) as prof:
    for _ in range(steps):
        tr.step(x)
    torch.cuda.synchronize()
rows = [(e.key, e.count / steps) for e in prof.key_averages() if e.key.startswith("aten::")]
rows.sort(key=lambda r: -r[1])
for k, c in rows[:40]:
    print(f"{k:40s} {c:6.1f} / step")
print()
for e in prof.key_averages(group_by_stack_n=4):
    if e.key in ("aten::clone", "aten::zeros", "aten::fill_", "aten::mul", "aten::add", "aten::copy_", "aten::zero_", "aten::zeros_like", "aten::isfinite", "aten::to"):
        st = [s for s in e.stack if "frl_hip" in s or "torch/autograd" in s][:2]
        print(f"{e.key:18s} {e.count / steps:5.1f}  {' <- '.join(s.split('/')[-1] for s in st)}")
