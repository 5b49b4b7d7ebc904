"""Error map of the fused / modular two-layer encoder backward against (a) the exact float64 chain and (b) float64 chains with bf16 rounding
emulated at candidate places.  Prints max-relative errors per parameter.  Diagnostic only."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vq-vae_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import frl_oracle as O  # noqa: E402
from frl_hip.models.blocks import Conv2DEncoder  # noqa: E402


def q(t):
    return t.to(torch.bfloat16).double()


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def rel(a, b):
    return ((a - b).abs().max() / b.abs().max()).item()


class RF(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t):
        return q(t)

    @staticmethod
    def backward(ctx, g):
        return g


class RB(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t):
        return t.clone()

    @staticmethod
    def backward(ctx, g):
        return q(g)


B, H, W = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (3, 32, 32)))
g = torch.Generator().manual_seed(B * H + W)
torch.manual_seed(B * H + W)
enc = Conv2DEncoder(64, [128, 64], num_groups=8)
enc.fuse_min_samples = 1
with torch.no_grad():
    for prm in enc.parameters():
        if prm.dim() == 1:
            prm.copy_(torch.randn(prm.shape, generator=g) * 0.3 + (1.0 if prm.mean() > 0.5 else 0.0))
x = q(torch.randn(B, 64, H, W, generator=g) * 1.3 + 0.2)
dz = q(torch.randn(B, 64, H, W, generator=g))
base = {n: (q(p.detach()) if p.dim() == 4 else p.detach().double()) for n, p in enc.named_parameters()}
names = list(base)
convs = [n for n in names if base[n].dim() == 4]
gam = [n for n in names if n.endswith("weight") and base[n].dim() == 1]
bet = [n for n in names if n.endswith("bias") and base[n].dim() == 1]


def ref(round_c1=False, round_h=False, round_dc=False, round_dh=False, round_c2=False):
    ps = {n: v.clone().requires_grad_(True) for n, v in base.items()}
    c1 = F.conv2d(x, ps[convs[0]])
    if round_dc:
        c1 = RB.apply(c1)
    if round_c1:
        c1 = RF.apply(c1)
    h = F.relu(O.group_norm(c1, 8, ps[gam[0]], ps[bet[0]]))
    if round_dh:
        h = RB.apply(h)
    if round_h:
        h = RF.apply(h)
    c2 = F.conv2d(h, ps[convs[1]])
    if round_dc:
        c2 = RB.apply(c2)
    if round_c2:
        c2 = RF.apply(c2)
    z = O.group_norm(c2, 8, ps[gam[1]], ps[bet[1]])
    z.backward(dz)
    return {n: ps[n].grad for n in names}


refs = {"exact": ref(), "h": ref(round_h=True), "h+dc": ref(round_h=True, round_dc=True), "h+dc+dh": ref(round_h=True, round_dc=True, round_dh=True),
        "all+c1c2": ref(True, True, True, True, True)}
enc = enc.cuda().train()
xd, dzd = nhwc(x).to(torch.bfloat16).cuda(), nhwc(dz).to(torch.bfloat16).cuda()
out = {}
for fuse in (True, False):
    enc.fuse = fuse
    enc.zero_grad(set_to_none=True)
    enc(xd).backward(dzd)
    out[fuse] = {n: p.grad.detach().cpu().double().reshape(base[n].shape) for n, p in enc.named_parameters()}
for n in names:
    row = [f"{n:18s}"]
    for fuse in (True, False):
        row.append(("fused  " if fuse else "modular") + " " + " ".join(f"{k}={rel(out[fuse][n], r[n]):.4f}" for k, r in refs.items()))
    row.append(f"fused-vs-modular={rel(out[True][n], out[False][n]):.4f}")
    print(" | ".join(row))
n0 = convs[0]
print("ref-vs-ref on", n0, {k: round(rel(r[n0], refs["exact"][n0]), 4) for k, r in refs.items()})
