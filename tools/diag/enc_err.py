import sys, os, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vq-vae_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import frl_oracle as O
from frl_hip.models.blocks import Conv2DEncoder
def q(t): return t.to(torch.bfloat16).double()
def nhwc(t): return t.permute(0, 2, 3, 1).contiguous()
def rel(a, b): return ((a - b).abs().max() / b.abs().max()).item()
B, H, W = 3, 32, 32
g = torch.Generator().manual_seed(B * H + W)
enc = Conv2DEncoder(64, [128, 64], num_groups=8)
with torch.no_grad():
    for prm in enc.parameters():
        if prm.dim() == 1: prm.copy_(torch.randn(prm.shape, generator=g) * 0.3 + (1.0 if prm.mean() > 0.5 else 0.0))
x = q(torch.randn(B, 64, H, W, generator=g) * 1.3 + 0.2); dz = q(torch.randn(B, 64, H, W, generator=g))
ps = {n: (q(p.detach()) if p.dim() == 4 else p.detach().double()).requires_grad_(True) for n, p in enc.named_parameters()}
names = list(ps)
convs = [n for n in names if ps[n].dim() == 4]; gam = [n for n in names if n.endswith("weight") and ps[n].dim() == 1]; bet = [n for n in names if n.endswith("bias") and ps[n].dim() == 1]
y = F.relu(O.group_norm(F.conv2d(x, ps[convs[0]]), 8, ps[gam[0]], ps[bet[0]]))
zr = O.group_norm(F.conv2d(y, ps[convs[1]]), 8, ps[gam[1]], ps[bet[1]]); zr.backward(dz)
enc = enc.cuda().train(); xd, dzd = nhwc(x).to(torch.bfloat16).cuda(), nhwc(dz).to(torch.bfloat16).cuda()
out = {}
for fuse in (True, False):
    enc.fuse = fuse; enc.zero_grad(set_to_none=True); z = enc(xd); z.backward(dzd)
    out[fuse] = {n: p.grad.detach().cpu().double().reshape(ps[n].shape) for n, p in enc.named_parameters()}
for n in names:
    print(n, "fused", round(rel(out[True][n], ps[n].grad), 4), "modular", round(rel(out[False][n], ps[n].grad), 4), "fused-vs-modular", round(rel(out[True][n], out[False][n]), 4))
d = (out[True][convs[0]] - ps[convs[0]].grad).abs().reshape(128, 64)
print("dW1 err by 16-row block:", [round(d[i*16:(i+1)*16].max().item(), 2) for i in range(8)], "by 16-col block:", [round(d[:, i*16:(i+1)*16].max().item(), 2) for i in range(4)], "max|ref|", ps[convs[0]].grad.abs().max().item())
