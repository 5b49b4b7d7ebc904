"""GPU parity of GroupNorm, 3x3 conv, stencils, fused TCN block and streaming ops vs the CPU oracle (float64 autograd)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import frl_oracle as O  # noqa: E402

DEV = "cuda:0"
MODES = [(torch.float32, 3e-6, 2e-5), (torch.bfloat16, 3e-2, 3e-2)]  # (dtype, activation tol, weight-grad tol) relative to max|ref|


def rel_err(got, ref):
    ref = ref.double()
    return (got.detach().cpu().double() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-6)


def q(t, dtype):
    """Quantise to the storage dtype and return float64 master of the quantised value."""
    return t.to(dtype).double()


def nhwc(t):  # [B,C,H,W] -> [B,H,W,C]
    return t.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("dtype,atol,wtol", MODES)
@pytest.mark.parametrize("B,HW,C,G,relu", [(3, 1024, 128, 8, True), (2, 64, 16, 4, True), (2, 64, 8, 4, False), (5, 100, 64, 8, False)])
def test_groupnorm(dtype, atol, wtol, B, HW, C, G, relu):
    from frl_hip import ops
    g = torch.Generator().manual_seed(B * HW + C)
    x = q(torch.randn(B, HW, C, generator=g) * 1.5 + 0.3, dtype).requires_grad_(True)
    gam = (torch.rand(C, generator=g) + 0.5).double().requires_grad_(True)
    bet = (torch.randn(C, generator=g) * 0.2).double().requires_grad_(True)
    dy = q(torch.randn(B, HW, C, generator=g), dtype)
    ref = O.group_norm(x.permute(0, 2, 1), G, gam, bet).permute(0, 2, 1)
    if relu:
        ref = F.relu(ref)
    ref.backward(dy)
    xd, dyd = x.detach().to(dtype).to(DEV), dy.to(dtype).to(DEV)
    gd, bd = gam.detach().float().to(DEV), bet.detach().float().to(DEV)
    y, mean, rstd = ops.groupnorm_fwd(xd, gd, bd, G, 1e-5, relu)
    assert rel_err(y.float(), ref) <= atol
    dx, dg, db = ops.groupnorm_bwd(dyd, xd, gd, bd, mean, rstd, G, relu)
    assert rel_err(dx.float(), x.grad) <= atol * 2
    assert rel_err(dg, gam.grad) <= wtol and rel_err(db, bet.grad) <= wtol


@pytest.mark.parametrize("dtype,atol,wtol", MODES)
@pytest.mark.parametrize("B,H,W,cin,cout,act", [(2, 32, 32, 128, 64, 1), (1, 8, 8, 16, 8, 1), (2, 8, 8, 8, 8, 2), (1, 13, 21, 64, 64, 0),
                                                (1, 16, 16, 64, 128, 2),
                                                # bf16 band kernel of the weight gradient: bands side by side (W = 64), two output slices,
                                                # no mask, and more bands than workgroups (two per workgroup, idle workgroups write zeros)
                                                (3, 16, 64, 64, 128, 2), (1, 8, 32, 64, 64, 0), (70, 32, 32, 64, 64, 1)])
def test_conv3x3(dtype, atol, wtol, B, H, W, cin, cout, act):
    from frl_hip import ops
    g = torch.Generator().manual_seed(H * W + cin + cout)
    x = q(torch.randn(B, cin, H, W, generator=g), dtype).requires_grad_(True)
    w = (torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5))
    wq = (q(w, dtype) if dtype == torch.bfloat16 else w.double()).requires_grad_(True)
    b = (torch.randn(cout, generator=g) * 0.1).double().requires_grad_(True)
    y = F.conv2d(x, wq, b, padding=1)
    y = F.relu(y) if act == 1 else torch.sigmoid(y) if act == 2 else y
    dy = q(torch.randn(B, cout, H, W, generator=g), dtype)
    xd = nhwc(x.detach()).to(dtype).to(DEV)
    wd, bd = w.float().to(DEV), b.detach().float().to(DEV)
    yd = ops.conv3x3_fwd(xd, wd, bd, act)
    assert rel_err(yd.float(), nhwc(y.detach())) <= atol
    # backward through the same activation using the DEVICE output as the mask (as autograd does)
    yq = q(yd.float().cpu(), dtype).permute(0, 3, 1, 2)
    x2 = x.detach().clone().requires_grad_(True)
    w2 = wq.detach().clone().requires_grad_(True)
    b2 = b.detach().clone().requires_grad_(True)
    pre = F.conv2d(x2, w2, b2, padding=1)
    dpre = dy * ((yq > 0).double() if act == 1 else (yq * (1 - yq)) if act == 2 else 1.0)
    pre.backward(dpre)
    dyd = nhwc(dy).to(dtype).to(DEV)
    dx = ops.conv3x3_bwd_data(dyd, wd, yd if act else None, act)
    assert rel_err(dx.float(), nhwc(x2.grad)) <= atol * 2
    for scalar in (False, True):
        dw, db = ops.conv3x3_bwd_weight(dyd, xd, yd if act else None, act, scalar_frags=scalar)
        assert rel_err(dw, w2.grad) <= wtol * 2 and rel_err(db, b2.grad) <= wtol * 2


@pytest.mark.parametrize("B,H,W", [(3, 32, 32), (5, 16, 16), (2, 4, 8)])
def test_fused_two_layer_encoder_matches_float64_and_the_modular_path(B, H, W):
    """Conv2DEncoder 64 -> 128 -> 64 (bf16): the one-launch-per-direction kernels (csrc/enc_fused.hip) against float64 autograd of
    conv1x1 -> GroupNorm(8) -> ReLU -> conv1x1 -> GroupNorm(8), and against the modular kernels on the same inputs."""
    from frl_hip.models.blocks import Conv2DEncoder
    g = torch.Generator().manual_seed(B * H + W)
    torch.manual_seed(B * H + W)                                    # (the constructor draws the convolution weights from the global generator)
    enc = Conv2DEncoder(64, [128, 64], num_groups=8)
    enc.fuse_min_samples = 1                                        # (the production threshold keeps small batches on the modular kernels)
    with torch.no_grad():
        for prm in enc.parameters():
            if prm.dim() == 1:
                prm.copy_(torch.randn(prm.shape, generator=g) * 0.3 + (1.0 if prm.mean() > 0.5 else 0.0))
    x = q(torch.randn(B, 64, H, W, generator=g) * 1.3 + 0.2, torch.bfloat16)
    dz = q(torch.randn(B, 64, H, W, generator=g), torch.bfloat16)
    names = [n for n, _ in enc.named_parameters()]
    # float64 reference (weights as the kernels see them: rounded to bf16)
    ps = {n: (q(p.detach(), torch.bfloat16) if p.dim() == 4 else p.detach().double()).requires_grad_(True) for n, p in enc.named_parameters()}
    convs = [n for n in names if ps[n].dim() == 4]
    gam = [n for n in names if n.endswith("weight") and ps[n].dim() == 1]
    bet = [n for n in names if n.endswith("bias") and ps[n].dim() == 1]
    y = F.relu(O.group_norm(F.conv2d(x, ps[convs[0]]), 8, ps[gam[0]], ps[bet[0]]))
    zr = O.group_norm(F.conv2d(y, ps[convs[1]]), 8, ps[gam[1]], ps[bet[1]])
    zr.backward(dz)
    # The same chain in float64 with bf16 rounding emulated where the kernels round: the hidden activation h (operand of the second
    # convolution), the gradients of the two convolution outputs (operands of dW = dy^T x and W^T dy) and d h = W2^T dy2.  The modular
    # kernels additionally STORE the convolution outputs y1, y2 (the GroupNorm inputs) as bf16 tensors; the fused kernels keep them in
    # float32.  That one rounding matters: the GroupNorm backward removes the projections of the gradient onto 1 and xhat, dW1 is a small
    # remainder of cancelling sums, and a 2^-9 perturbation of xhat moves it by 3-9 % of max |dW1| (tools/diag/enc_err.py).  So:
    # the fused path is held tightly to the chain WITHOUT the y1 / y2 rounding (and through it to the exact one), the modular path
    # tightly to the chain WITH it -- what is left in both cases is accumulation order, not operand precision.
    class _RoundFwd(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return q(t, torch.bfloat16)

        @staticmethod
        def backward(ctx, g_):
            return g_

    class _RoundBwd(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return t.clone()

        @staticmethod
        def backward(ctx, g_):
            return q(g_, torch.bfloat16)

    def emulated(round_prenorm):
        pe = {n: ps[n].detach().clone().requires_grad_(True) for n in names}
        c1 = _RoundBwd.apply(F.conv2d(x, pe[convs[0]]))
        if round_prenorm:
            c1 = _RoundFwd.apply(c1)
        h = _RoundFwd.apply(_RoundBwd.apply(F.relu(O.group_norm(c1, 8, pe[gam[0]], pe[bet[0]]))))
        c2 = _RoundBwd.apply(F.conv2d(h, pe[convs[1]]))
        if round_prenorm:
            c2 = _RoundFwd.apply(c2)
        O.group_norm(c2, 8, pe[gam[1]], pe[bet[1]]).backward(dz)
        return {n: pe[n].grad for n in names}

    ref_fused, ref_modular = emulated(False), emulated(True)
    enc = enc.to(DEV).train()
    xd, dzd = nhwc(x).to(torch.bfloat16).to(DEV), nhwc(dz).to(torch.bfloat16).to(DEV)
    out = {}
    for fuse in (True, False):
        enc.fuse = fuse
        enc.zero_grad(set_to_none=True)
        z = enc(xd)
        z.backward(dzd)
        out[fuse] = (z.detach().float().cpu(), {n: p.grad.detach().cpu().double().reshape(ps[n].shape) for n, p in enc.named_parameters()})
    assert enc._fused_layers(xd) is None and out[True][0].shape == out[False][0].shape   # (fuse is off now)
    for fuse in (True, False):
        z, grads = out[fuse]
        assert rel_err(z, nhwc(zr.detach())) <= 3e-2, fuse
        for n in names:
            assert rel_err(grads[n], (ref_fused if fuse else ref_modular)[n]) <= 5e-3, (fuse, n)      # every sample size
    # the fused path against the EXACT float64 chain: only operand rounding of h, dh, dy is left (was 7e-2 with y1 / y2 rounded)
    for n in names:
        assert rel_err(out[True][1][n], ps[n].grad) <= 1.5e-2, n
    assert rel_err(out[True][0], out[False][0].double()) <= 1.6e-2


@pytest.mark.parametrize("dtype,atol,wtol", MODES)
@pytest.mark.parametrize("B,H,W,C", [(2, 32, 32, 64), (1, 8, 8, 8), (1, 9, 13, 16)])
def test_sobel(dtype, atol, wtol, B, H, W, C):
    from frl_hip import ops
    g = torch.Generator().manual_seed(C + H)
    x = q(torch.randn(B, C, H, W, generator=g), dtype).requires_grad_(True)
    sx = (torch.tensor([[-1., 0., 1.], [-2., 0., 2.], [-1., 0., 1.]]) / 4).double().reshape(1, 1, 3, 3).expand(C, 1, 3, 3)
    sy = (torch.tensor([[-1., -2., -1.], [0., 0., 0.], [1., 2., 1.]]) / 4).double().reshape(1, 1, 3, 3).expand(C, 1, 3, 3)
    ref = torch.cat([F.conv2d(x, sx, padding=1, groups=C), F.conv2d(x, sy, padding=1, groups=C)], 1)
    dg = q(torch.randn(B, 2 * C, H, W, generator=g), dtype)
    ref.backward(dg)
    gd = ops.sobel_fwd(nhwc(x.detach()).to(dtype).to(DEV))
    assert rel_err(gd.float(), nhwc(ref.detach())) <= atol
    dx = ops.sobel_bwd(nhwc(dg).to(dtype).to(DEV))
    assert rel_err(dx.float(), nhwc(x.grad)) <= atol * 2


def _smooth_ref(x, al, bl, R, dil):
    b, c, hh, ww = x.shape
    k = 8
    tmpl = [[[0., 0., 0.], [1 / 3, 1 / 3, 1 / 3], [0., 0., 0.]], [[0., 1 / 3, 0.], [0., 1 / 3, 0.], [0., 1 / 3, 0.]],
            [[1 / 3, 0., 0.], [0., 1 / 3, 0.], [0., 0., 1 / 3]], [[0., 0., 1 / 3], [0., 1 / 3, 0.], [1 / 3, 0., 0.]]]
    a = torch.softmax(al.reshape(b, k, R, hh, ww), dim=1)
    bw = torch.softmax(bl.reshape(b, c, R, hh, ww), dim=2)
    slot = torch.zeros(b, c, R, hh, ww, dtype=x.dtype)
    for i in range(4):
        filt = torch.tensor(tmpl[i], dtype=x.dtype).reshape(1, 1, 3, 3).expand(c, 1, 3, 3)
        fine = F.conv2d(x, filt, padding=1, groups=c)
        coarse = F.conv2d(x, filt, padding=dil, dilation=dil, groups=c)
        slot = slot + fine.unsqueeze(2) * a[:, 2 * i].unsqueeze(1) + coarse.unsqueeze(2) * a[:, 2 * i + 1].unsqueeze(1)
    sm = (bw * slot).sum(2)
    return sm, x - sm


@pytest.mark.parametrize("dtype,atol,wtol", MODES)
@pytest.mark.parametrize("B,H,W,C,R", [(2, 32, 32, 64, 4), (1, 8, 8, 8, 4), (1, 7, 11, 16, 2), (1, 8, 8, 48, 4)])
def test_edge_smooth_stencil(dtype, atol, wtol, B, H, W, C, R):
    from frl_hip import ops
    g = torch.Generator().manual_seed(C * R + H)
    x = q(torch.randn(B, C, H, W, generator=g), dtype).requires_grad_(True)
    al = q(torch.randn(B, 8 * R, H, W, generator=g), dtype).requires_grad_(True)
    bl = q(torch.randn(B, C * R, H, W, generator=g), dtype).requires_grad_(True)
    sm, res = _smooth_ref(x, al, bl, R, 3)
    xd, ald, bld = (nhwc(t.detach()).to(dtype).to(DEV) for t in (x, al, bl))
    smd, resd, asoft, bsoft = ops.edge_smooth_fwd(xd, ald, bld, R, 3)
    assert rel_err(smd.float(), nhwc(sm.detach())) <= atol
    assert rel_err(resd.float(), nhwc(res.detach())) <= atol
    ds = q(torch.randn(B, C, H, W, generator=g), dtype)
    sm.backward(ds)
    dx, da, db = ops.edge_smooth_bwd(nhwc(ds).to(dtype).to(DEV), xd, asoft, bsoft, R, 3)
    assert rel_err(dx.float(), nhwc(x.grad)) <= atol * 3
    assert rel_err(da.float(), nhwc(al.grad)) <= atol * 3
    assert rel_err(db.float(), nhwc(bl.grad)) <= atol * 3
    # additive term folded into the kernel's dx store (the residual branch of EdgeSmoothFn.backward): da / db unchanged bit for bit
    r = q(torch.randn(B, C, H, W, generator=g), dtype)
    dx2, da2, db2 = ops.edge_smooth_bwd(nhwc(ds).to(dtype).to(DEV), xd, asoft, bsoft, R, 3, dx_add=nhwc(r).to(dtype).to(DEV))
    assert rel_err(dx2.float(), nhwc(x.grad + r)) <= atol * 3
    assert torch.equal(da2, da) and torch.equal(db2, db)


@pytest.mark.parametrize("B,H,W,dil", [(2, 32, 32, 3), (1, 16, 24, 3), (3, 9, 13, 2), (1, 5, 7, 3), (2, 20, 16, 3), (1, 12, 64, 2)])
def test_fused_mixing_heads_softmax_bank_match_float64(B, H, W, dil):
    """csrc/smooth_fused.hip (bf16, 64 channels, 64 hidden features, rank 4): the two 1x1 heads, both softmaxes and the directional bank
    in one forward launch, and the backward pair (heads recomputed; d feat, d x, head parameter gradients produced in-kernel, no [P,288]
    tensor anywhere) against float64 autograd of spatial.py:300-331 on the same bf16-rounded inputs.  Shapes: the hot 32x32 tiles,
    pixel counts that are not multiples of the 16 / 128-pixel work items, images narrower than the coarse stencil, dilation 2; widths
    that take the LDS-tiled d x kernel (multiples of 16; 20 rows = a ragged last band) and widths that take its gather form."""
    from frl_hip import ops
    C, HID, R = 64, 64, 4
    g = torch.Generator().manual_seed(H * W + dil)
    dt = torch.bfloat16
    x = q(torch.randn(B, C, H, W, generator=g), dt).requires_grad_(True)
    feat = q(torch.randn(B, HID, H, W, generator=g).clamp_min(0.0), dt).requires_grad_(True)
    wa = (torch.randn(8 * R, HID, generator=g) * 0.3)
    ba = torch.randn(8 * R, generator=g) * 0.2
    wb = (torch.randn(C * R, HID, generator=g) * 0.3)
    bb = torch.randn(C * R, generator=g) * 0.2
    # the kernels round the weights to bf16 for the matrix cores: the float64 reference uses the same rounded values
    wa64, wb64 = q(wa, dt).requires_grad_(True), q(wb, dt).requires_grad_(True)
    ba64, bb64 = ba.double().requires_grad_(True), bb.double().requires_grad_(True)
    al = F.conv2d(feat, wa64.reshape(8 * R, HID, 1, 1), ba64)
    bl = F.conv2d(feat, wb64.reshape(C * R, HID, 1, 1), bb64)
    sm, res = _smooth_ref(x, al, bl, R, dil)
    xd, fd = nhwc(x.detach()).to(dt).to(DEV), nhwc(feat.detach()).to(dt).to(DEV)
    wad, bad, wbd, bbd = (t.float().to(DEV).contiguous() for t in (wa64.detach(), ba, wb64.detach(), bb))
    assert ops.smooth_heads_supported(xd, HID, R, wad, bad, wbd, bbd)
    smd, resd = ops.smooth_heads_fwd(xd, fd, wad, bad, wbd, bbd, dil)
    assert rel_err(smd.float(), nhwc(sm.detach())) <= 8e-3
    assert rel_err(resd.float(), nhwc(res.detach())) <= 8e-3
    ds = q(torch.randn(B, C, H, W, generator=g), dt)
    add = q(torch.randn(B, C, H, W, generator=g), dt)
    sm.backward(ds)
    dx, dfeat, dwa, dba, dwb, dbb = ops.smooth_heads_bwd(nhwc(ds).to(dt).to(DEV), xd, fd, wad, bad, wbd, bbd, dil, dx_add=nhwc(add).to(dt).to(DEV))
    assert rel_err(dx.float(), nhwc(x.grad + add)) <= 1.5e-2
    assert rel_err(dfeat.float(), nhwc(feat.grad)) <= 3e-2
    for got, ref in ((dwa, wa64.grad), (dba, ba64.grad), (dwb, wb64.grad), (dbb, bb64.grad)):
        assert rel_err(got, ref) <= 3e-2
    dx0 = ops.smooth_heads_bwd(nhwc(ds).to(dt).to(DEV), xd, fd, wad, bad, wbd, bbd, dil)[0]
    assert rel_err(dx0.float(), nhwc(x.grad)) <= 1.5e-2
    # the two forms of the d x kernel (LDS tiles / L2 gathers) add the same terms, associated differently: equal to a bf16 ulp
    from frl_hip import _lib
    was = _lib.load().frl_smooth_heads_force_gather(1)
    try:
        dx1 = ops.smooth_heads_bwd(nhwc(ds).to(dt).to(DEV), xd, fd, wad, bad, wbd, bbd, dil)[0]
    finally:
        _lib.load().frl_smooth_heads_force_gather(was)
    assert rel_err(dx1.float(), dx0.float().cpu()) <= 8e-3 and rel_err(dx1.float(), nhwc(x.grad)) <= 1.5e-2
    # not the hot configuration: the callers keep the modular kernels
    assert not ops.smooth_heads_supported(xd.float(), HID, R, wad, bad, wbd, bbd)
    assert not ops.smooth_heads_supported(xd, HID, R, wad, None, wbd, bbd)


@pytest.mark.parametrize("dtype,atol,wtol", MODES)
@pytest.mark.parametrize("B,T,HW,cin,cout,G,dil", [(2, 5, 64, 8, 8, 4, 1), (2, 5, 64, 8, 8, 4, 4), (1, 5, 1024, 64, 64, 8, 2), (3, 10, 50, 64, 64, 8, 4),
                                                   (1, 5, 37, 16, 8, 4, 1), (1, 15, 16, 64, 64, 8, 4), (1, 5, 32, 12, 64, 8, 2)])
def test_tcn_block(dtype, atol, wtol, B, T, HW, cin, cout, G, dil):
    from frl_hip import ops
    g = torch.Generator().manual_seed(T * HW + cin + cout + dil)
    prefix = "b."
    st = {}
    st[prefix + "conv.weight"] = torch.randn(cout, cin, 3, generator=g) / (3 * cin) ** 0.5
    st[prefix + "conv.bias"] = torch.randn(cout, generator=g) * 0.1
    st[prefix + "norm.weight"] = torch.rand(cout, generator=g) + 0.5
    st[prefix + "norm.bias"] = torch.randn(cout, generator=g) * 0.2
    st[prefix + "gate.weight"] = torch.randn(cout, cout, 1, generator=g) / cout ** 0.5
    st[prefix + "gate.bias"] = torch.randn(cout, generator=g) * 0.1
    if cin != cout:
        st[prefix + "projection.weight"] = torch.randn(cout, cin, 1, generator=g) / cin ** 0.5
        st[prefix + "projection.bias"] = torch.randn(cout, generator=g) * 0.1
    mm = {"conv.weight", "gate.weight", "projection.weight"}
    ref_st = {k: ((q(v, dtype) if (dtype == torch.bfloat16 and k[len(prefix):] in mm) else v.double()).requires_grad_(True))
              for k, v in st.items()}
    x = q(torch.randn(B, T, HW, cin, generator=g), dtype).requires_grad_(True)     # [B,T,HW,C]
    xr = x.permute(0, 2, 3, 1).reshape(B * HW, cin, T)                                # [N,C,T]
    yr = O.tcn_block_forward(ref_st, xr, dil, G, prefix)
    y_ref = yr.reshape(B, HW, cout, T).permute(0, 3, 1, 2)
    dy = q(torch.randn(B, T, HW, cout, generator=g), dtype)
    y_ref.backward(dy)
    dev = {k: v.float().to(DEV) for k, v in st.items()}
    args = (dev[prefix + "conv.weight"], dev[prefix + "conv.bias"], dev[prefix + "norm.weight"], dev[prefix + "norm.bias"],
            dev[prefix + "gate.weight"], dev[prefix + "gate.bias"], dev.get(prefix + "projection.weight"), dev.get(prefix + "projection.bias"))
    if args[6] is not None:
        args = args[:6] + (args[6].reshape(cout, cin).contiguous(), args[7])
    xd = x.detach().to(dtype).to(DEV)
    yd = ops.tcn_block_fwd(xd, *args, dil, G)
    assert rel_err(yd.float(), y_ref.detach()) <= atol * 2
    gr = ops.tcn_block_bwd(xd, dy.to(dtype).to(DEV), *args, dil, G)
    assert rel_err(gr["dx"].float(), x.grad) <= atol * 4
    names = dict(conv_w="conv.weight", conv_b="conv.bias", gn_w="norm.weight", gn_b="norm.bias", gate_w="gate.weight", gate_b="gate.bias",
                 proj_w="projection.weight", proj_b="projection.bias")
    for kk, nm in names.items():
        if kk in gr:
            assert rel_err(gr[kk].reshape(ref_st[prefix + nm].shape), ref_st[prefix + nm].grad) <= wtol * 4, kk


@pytest.mark.parametrize("dtype,atol,wtol", MODES)
def test_streaming_ops(dtype, atol, wtol):
    from frl_hip import ops
    g = torch.Generator().manual_seed(1)
    B, T, H, W, C = 2, 5, 8, 8, 12
    # mean over time
    tile = q(torch.randn(B, T, H, W, 16, generator=g), dtype)
    assert rel_err(ops.mean_time(tile.to(dtype).to(DEV)).float(), tile.mean(1)) <= atol
    # FiLM
    h = q(torch.randn(B, T, H * W, C, generator=g), dtype).requires_grad_(True)
    ga = q(torch.randn(B, H * W, C, generator=g), dtype).requires_grad_(True)
    be = q(torch.randn(B, H * W, C, generator=g), dtype).requires_grad_(True)
    out = ga.unsqueeze(1) * h + be.unsqueeze(1)
    do = q(torch.randn(B, T, H * W, C, generator=g), dtype)
    out.backward(do)
    hd, gd, bd = (t.detach().to(dtype).to(DEV) for t in (h, ga, be))
    assert rel_err(ops.film_modulate_fwd(hd, gd, bd).float(), out.detach()) <= atol
    dh, dg, db = ops.film_modulate_bwd(do.to(dtype).to(DEV), hd, gd)
    assert rel_err(dh.float(), h.grad) <= atol and rel_err(dg.float(), ga.grad) <= atol * 2 and rel_err(db.float(), be.grad) <= atol * 2
    # MSE with and without mask
    p = q(torch.randn(300, 64, generator=g), dtype).requires_grad_(True)
    t = q(torch.randn(300, 64, generator=g), dtype)
    for use_mask in (False, True):
        mask = (torch.rand(300, generator=g) > 0.3) if use_mask else None
        p.grad = None
        loss = O.reconstruction_loss_l2(p, t, mask.unsqueeze(1).expand(300, 64) if use_mask else None)
        (loss * 0.7).backward()
        md = mask.to(torch.uint8).to(DEV) if use_mask else None
        st = ops.mse_fwd(p.detach().to(dtype).to(DEV), t.to(dtype).to(DEV), md)
        assert abs(st[0].item() - loss.item()) <= 1e-5 * loss.item()
        dp = ops.mse_bwd(p.detach().to(dtype).to(DEV), t.to(dtype).to(DEV), md, torch.tensor([0.7], device=DEV), st)
        assert rel_err(dp.float(), p.grad) <= atol
    # gate blend
    for mg in (0.0, 0.55):
        sm = q(torch.randn(64, 16, generator=g), dtype).requires_grad_(True)
        rs = q(torch.randn(64, 16, generator=g), dtype).requires_grad_(True)
        gr_ = q(torch.rand(64, 16, generator=g), dtype).requires_grad_(True)
        gate = gr_.clamp(min=mg) if mg > 0 else gr_
        o = sm + gate * rs
        do, dge = q(torch.randn(64, 16, generator=g), dtype), q(torch.randn(64, 16, generator=g), dtype)
        (o * do + gate * dge).sum().backward()
        od, gd_ = ops.gate_blend_fwd(*(t.detach().to(dtype).to(DEV) for t in (sm, rs, gr_)), mg)
        assert rel_err(od.float(), o.detach()) <= atol and rel_err(gd_.float(), gate.detach()) <= atol
        dres, dgraw = ops.gate_blend_bwd(do.to(dtype).to(DEV), dge.to(dtype).to(DEV), rs.detach().to(dtype).to(DEV), gr_.detach().to(dtype).to(DEV), mg)
        assert rel_err(dres.float(), rs.grad) <= atol and rel_err(dgraw.float(), gr_.grad) <= atol
    a, b = q(torch.randn(128, 8, generator=g), dtype), q(torch.randn(128, 8, generator=g), dtype)
    assert rel_err(ops.add(a.to(dtype).to(DEV), b.to(dtype).to(DEV)).float(), a + b) <= atol


@pytest.mark.parametrize("B,T,HW,dil", [(1, 5, 1024, 1), (2, 5, 1024, 2), (3, 5, 100, 4), (1, 3, 77, 1), (1, 4, 64, 2)])
def test_tcn_block_bwd_fused_matches_oracle_and_unfused(B, T, HW, dil):
    """The fused bf16 backward (one launch: dx + all parameter gradients) vs float64 autograd and vs the unfused kernels."""
    from frl_hip import ops
    dtype, cin, cout, G = torch.bfloat16, 64, 64, 8
    g = torch.Generator().manual_seed(T * HW + dil)
    p = "b."
    st = {p + "conv.weight": torch.randn(cout, cin, 3, generator=g) / (3 * cin) ** 0.5, p + "conv.bias": torch.randn(cout, generator=g) * 0.1,
          p + "norm.weight": torch.rand(cout, generator=g) + 0.5, p + "norm.bias": torch.randn(cout, generator=g) * 0.2,
          p + "gate.weight": torch.randn(cout, cout, 1, generator=g) / cout ** 0.5, p + "gate.bias": torch.randn(cout, generator=g) * 0.1}
    ref_st = {k: ((q(v, dtype) if k.endswith(("conv.weight", "gate.weight")) else v.double()).requires_grad_(True)) for k, v in st.items()}
    x = q(torch.randn(B, T, HW, cin, generator=g), dtype).requires_grad_(True)
    yr = O.tcn_block_forward(ref_st, x.permute(0, 2, 3, 1).reshape(B * HW, cin, T), dil, G, p)
    dy = q(torch.randn(B, T, HW, cout, generator=g), dtype)
    yr.reshape(B, HW, cout, T).permute(0, 3, 1, 2).backward(dy)
    dev = {k: v.float().to(DEV) for k, v in st.items()}
    args = (dev[p + "conv.weight"], dev[p + "conv.bias"], dev[p + "norm.weight"], dev[p + "norm.bias"], dev[p + "gate.weight"], dev[p + "gate.bias"], None, None)
    xd, dyd = x.detach().to(dtype).to(DEV), dy.to(dtype).to(DEV)
    fused = ops.tcn_block_bwd(xd, dyd, *args, dil, G)
    unf = ops.tcn_block_bwd(xd, dyd, *args, dil, G, allow_fused=False)
    names = dict(conv_w="conv.weight", conv_b="conv.bias", gn_w="norm.weight", gn_b="norm.bias", gate_w="gate.weight", gate_b="gate.bias")
    assert rel_err(fused["dx"].float(), x.grad) <= 0.1
    assert rel_err(fused["dx"].float(), unf["dx"].float().cpu()) <= 0.05
    for kk, nm in names.items():
        ref = ref_st[p + nm].grad
        assert rel_err(fused[kk].reshape(ref.shape), ref) <= 0.1, kk
        assert rel_err(fused[kk].reshape(ref.shape), unf[kk].reshape(ref.shape).cpu()) <= 0.03, kk


@pytest.mark.parametrize("B,HW,dil", [(1, 1024, 1), (2, 1024, 2), (3, 100, 4), (1, 77, 2), (5, 13, 1), (1, 16, 4), (3, 64, 4), (2, 4096, 1),
                                      (3, 1024, 4)])
def test_tcn_hot_kernels_match_oracle_and_generic(B, HW, dil):
    """The (T=5, dilation)-specialised bf16 block kernels vs float64 autograd and vs the generic kernels (ragged pixel counts too)."""
    from frl_hip import ops, _lib
    dtype, cin, cout, G, T = torch.bfloat16, 64, 64, 8, 5
    assert _lib.load().frl_tcn_hot_supported(T, cin, cout, G, dil, 0, 1) == 1
    assert _lib.load().frl_tcn_hot_supported(T, cin, cout, G, 3, 0, 1) == 0 and _lib.load().frl_tcn_hot_supported(4, cin, cout, G, dil, 0, 1) == 0
    g = torch.Generator().manual_seed(B * HW + dil)
    p = "b."
    st = {p + "conv.weight": torch.randn(cout, cin, 3, generator=g) / (3 * cin) ** 0.5, p + "conv.bias": torch.randn(cout, generator=g) * 0.1,
          p + "norm.weight": torch.rand(cout, generator=g) + 0.5, p + "norm.bias": torch.randn(cout, generator=g) * 0.2,
          p + "gate.weight": torch.randn(cout, cout, 1, generator=g) / cout ** 0.5, p + "gate.bias": torch.randn(cout, generator=g) * 0.1}
    ref_st = {k: ((q(v, dtype) if k.endswith(("conv.weight", "gate.weight")) else v.double()).requires_grad_(True)) for k, v in st.items()}
    x = q(torch.randn(B, T, HW, cin, generator=g), dtype).requires_grad_(True)
    yr = O.tcn_block_forward(ref_st, x.permute(0, 2, 3, 1).reshape(B * HW, cin, T), dil, G, p).reshape(B, HW, cout, T).permute(0, 3, 1, 2)
    dy = q(torch.randn(B, T, HW, cout, generator=g), dtype)
    yr.backward(dy)
    dev = {k: v.float().to(DEV) for k, v in st.items()}
    args = (dev[p + "conv.weight"], dev[p + "conv.bias"], dev[p + "norm.weight"], dev[p + "norm.bias"], dev[p + "gate.weight"], dev[p + "gate.bias"], None, None)
    xd, dyd = x.detach().to(dtype).to(DEV), dy.to(dtype).to(DEV)
    y_hot = ops.tcn_block_fwd(xd, *args, dil, G)
    y_gen = ops.tcn_block_fwd(xd, *args, dil, G, allow_hot=False)
    assert rel_err(y_hot.float(), yr.detach()) <= 8e-3            # bf16 output rounding (2^-8 relative)
    assert rel_err(y_hot.float(), y_gen.float().cpu()) <= 8e-3
    hot = ops.tcn_block_bwd(xd, dyd, *args, dil, G)
    gen = ops.tcn_block_bwd(xd, dyd, *args, dil, G, allow_hot=False)
    assert rel_err(hot["dx"].float(), x.grad) <= 2e-2
    assert rel_err(hot["dx"].float(), gen["dx"].float().cpu()) <= 3e-2
    names = dict(conv_w="conv.weight", conv_b="conv.bias", gn_w="norm.weight", gn_b="norm.bias", gate_w="gate.weight", gate_b="gate.bias")
    for kk, nm in names.items():
        ref = ref_st[p + nm].grad
        assert rel_err(hot[kk].reshape(ref.shape), ref) <= 2e-2, kk
        assert rel_err(hot[kk].reshape(ref.shape), gen[kk].reshape(ref.shape).cpu()) <= 2e-2, kk


@pytest.mark.parametrize("variant", [4, 3])
@pytest.mark.parametrize("B,HW,dil", [(21, 1024, 1), (18, 1024, 2), (5, 4096, 4), (600, 64, 1), (3, 64, 2)])
def test_tcn_hot_bwd_staged_tiles_match_the_8_wave_kernel(B, HW, dil, variant):
    """frl_tcn_hot_bwd has three kernels (include/frl_hip.h): the LDS-staged ones (no mask, HW % 64 == 0; variant 4 = two independent
    4-wave subgroups per workgroup over 32-pixel tiles, variant 3 = 8 waves in lockstep over 64-pixel tiles) must agree with the round-1
    8-wave kernel on the same inputs to bf16 rounding of dx and to float32 summation order of the parameter gradients -- with MORE tiles
    than workgroups (256), so that every workgroup / subgroup walks several tiles (next-tile LDS-DMA, X / N buffer swap), an uneven
    number of tiles per subgroup, and a launch of three workgroups -- and twice in a row bit for bit (fixed-order reductions)."""
    from frl_hip import ops, _lib
    was = _lib.load().frl_tcn_hot_bwd_variant(variant)
    try:
        _staged_tiles_case(B, HW, dil)
    finally:
        _lib.load().frl_tcn_hot_bwd_variant(was)


def _staged_tiles_case(B, HW, dil):
    from frl_hip import ops, _lib
    dtype, cin, cout, G, T = torch.bfloat16, 64, 64, 8, 5
    assert B * HW // 64 > 256 or B * HW // 64 == 3
    g = torch.Generator().manual_seed(B * HW + dil)
    w = dict(conv_w=torch.randn(cout, cin, 3, generator=g) / (3 * cin) ** 0.5, conv_b=torch.randn(cout, generator=g) * 0.1,
             gn_w=torch.rand(cout, generator=g) + 0.5, gn_b=torch.randn(cout, generator=g) * 0.2,
             gate_w=torch.randn(cout, cout, 1, generator=g) / cout ** 0.5, gate_b=torch.randn(cout, generator=g) * 0.1)
    args = tuple(w[k].to(DEV) for k in ("conv_w", "conv_b", "gn_w", "gn_b", "gate_w", "gate_b")) + (None, None)
    xd = torch.randn(B, T, HW, cin, generator=g).to(dtype).to(DEV)
    dyd = torch.randn(B, T, HW, cout, generator=g).to(dtype).to(DEV)
    new = ops.tcn_block_bwd(xd, dyd, *args, dil, G)
    again = ops.tcn_block_bwd(xd, dyd, *args, dil, G)
    lib = _lib.load()
    lib.frl_tcn_hot_force_generic_tiles(1)
    try:
        old = ops.tcn_block_bwd(xd, dyd, *args, dil, G)
    finally:
        lib.frl_tcn_hot_force_generic_tiles(0)
    for k in new:
        assert torch.equal(new[k], again[k]), k
    # both round dx to bf16; variant 3 and the round-1 kernel also round the same dres = dy - dy g inside, variant 4 keeps dy g (rounded) and
    # subtracts late, so a quarter of its dx elements land on the neighbouring bf16 value: one ulp = 2^-7 of the element at worst
    d = (new["dx"].float() - old["dx"].float()).abs().cpu()
    assert d.max().item() <= 8e-3 * old["dx"].float().abs().max().item()
    assert d.mean().item() <= 2e-3 * old["dx"].float().abs().mean().item()
    for k in ("conv_w", "conv_b", "gn_w", "gn_b", "gate_w", "gate_b"):
        assert rel_err(new[k].cpu(), old[k].cpu()) <= 2e-3, k


@pytest.mark.parametrize("B,HW,dil,p", [(2, 1024, 1, 0.5), (3, 100, 4, 0.25), (1, 77, 2, 0.1)])
def test_tcn_hot_dropout1d_mask_matches_float64(B, HW, dil, p):
    """Training-mode Dropout1d of the block (tcn.py:53): with a GIVEN keep/scale mask per (pixel series, channel) the hot kernels
    (conv input = x .* mask, residual = x) must match float64 autograd of the same realisation, forward and backward."""
    from frl_hip import ops
    dtype, cin, cout, G, T = torch.bfloat16, 64, 64, 8, 5
    g = torch.Generator().manual_seed(B * HW + dil)
    pfx = "b."
    st = {pfx + "conv.weight": torch.randn(cout, cin, 3, generator=g) / (3 * cin) ** 0.5, pfx + "conv.bias": torch.randn(cout, generator=g) * 0.1,
          pfx + "norm.weight": torch.rand(cout, generator=g) + 0.5, pfx + "norm.bias": torch.randn(cout, generator=g) * 0.2,
          pfx + "gate.weight": torch.randn(cout, cout, 1, generator=g) / cout ** 0.5, pfx + "gate.bias": torch.randn(cout, generator=g) * 0.1}
    ref_st = {k: ((q(v, dtype) if k.endswith(("conv.weight", "gate.weight")) else v.double()).requires_grad_(True)) for k, v in st.items()}
    x = q(torch.randn(B, T, HW, cin, generator=g), dtype).requires_grad_(True)
    mask = q(((torch.rand(B, HW, cin, generator=g) >= p).float() / (1.0 - p)), dtype)            # bf16-rounded scale, as the module builds it
    # the kernel rounds x .* mask to bf16 before the MFMA: mirror that in the reference (straight-through for the gradient)
    xm = x * mask.unsqueeze(1)
    xm = xm + (q(xm.detach(), dtype) - xm.detach())
    xr = x.permute(0, 2, 3, 1).reshape(B * HW, cin, T)
    xmr = xm.permute(0, 2, 3, 1).reshape(B * HW, cin, T)
    w = ref_st[pfx + "conv.weight"]
    out = torch.nn.functional.conv1d(xmr, w, ref_st[pfx + "conv.bias"], padding=dil, dilation=dil)
    out = O.group_norm(out, G, ref_st[pfx + "norm.weight"], ref_st[pfx + "norm.bias"])
    gate = torch.sigmoid(torch.nn.functional.conv1d(out, ref_st[pfx + "gate.weight"], ref_st[pfx + "gate.bias"]))
    yr = (gate * torch.relu(out) + (1 - gate) * xr).reshape(B, HW, cout, T).permute(0, 3, 1, 2)
    # same thing through the oracle's own block with the mask argument (no bf16 rounding of the product): close to the above
    y_or = O.tcn_block_forward({k: v.detach() for k, v in ref_st.items()}, xr.detach(), dil, G, pfx, drop_mask=mask.reshape(B * HW, cin))
    assert rel_err(y_or.reshape(B, HW, cout, T).permute(0, 3, 1, 2), yr.detach()) <= 2e-2
    dy = q(torch.randn(B, T, HW, cout, generator=g), dtype)
    yr.backward(dy)
    dev = {k: v.float().to(DEV) for k, v in st.items()}
    args = (dev[pfx + "conv.weight"], dev[pfx + "conv.bias"], dev[pfx + "norm.weight"], dev[pfx + "norm.bias"], dev[pfx + "gate.weight"], dev[pfx + "gate.bias"], None, None)
    xd, dyd, md = x.detach().to(dtype).to(DEV), dy.to(dtype).to(DEV), mask.to(dtype).to(DEV)
    y = ops.tcn_block_fwd(xd, *args, dil, G, drop_mask=md)
    assert rel_err(y.float(), yr.detach()) <= 8e-3
    y0 = ops.tcn_block_fwd(xd, *args, dil, G)
    assert rel_err(y.float(), y0.float().cpu()) > 1e-2                                            # the mask really changes the output
    gr = ops.tcn_block_bwd(xd, dyd, *args, dil, G, drop_mask=md)
    assert rel_err(gr["dx"].float(), x.grad) <= 2e-2
    names = dict(conv_w="conv.weight", conv_b="conv.bias", gn_w="norm.weight", gn_b="norm.bias", gate_w="gate.weight", gate_b="gate.bias")
    for kk, nm in names.items():
        ref = ref_st[pfx + nm].grad
        assert rel_err(gr[kk].reshape(ref.shape), ref) <= 2e-2, kk
    with pytest.raises(NotImplementedError):
        ops.tcn_block_fwd(xd[:, :4].contiguous(), *args, dil, G, drop_mask=md)                     # T != 5: the raw generic kernels take no mask


@pytest.mark.parametrize("dtype,T,cin,cout,G,tol", [(torch.float32, 5, 32, 32, 8, 2e-5), (torch.bfloat16, 10, 64, 64, 8, 2e-2),
                                                    (torch.float32, 7, 8, 16, 4, 2e-5), (torch.float32, 5, 8, 8, 4, 2e-5)])
def test_gated_block_dropout1d_outside_the_hot_configuration(monkeypatch, dtype, T, cin, cout, G, tol):
    """Training-mode Dropout1d for shapes the hot kernels do not cover (float32 parity mode, T != 5, projection blocks): the module's
    two-input formulation on the generic kernels vs float64 autograd of the same mask realisation (tcn.py:89-110)."""
    from frl_hip.models import blocks
    B, HW, dil, p = 2, 96, 2, 0.3
    g = torch.Generator().manual_seed(T * cin + cout)
    blk = blocks.GatedResidualBlock(cin, cout, 3, dil, dropout_rate=p, num_groups=G).to(DEV)
    with torch.no_grad():
        for prm in blk.parameters():
            prm.copy_((torch.randn(prm.shape, generator=g) * (0.3 if prm.dim() > 1 else 0.2) + (1.0 if prm.dim() == 1 and prm is blk.norm.weight else 0.0)).to(DEV))
    mask = ((torch.rand(B, HW, cin, generator=g) >= p).float() / (1.0 - p))
    monkeypatch.setattr(blocks, "dropout_mask", lambda shape, rate, like: mask.to(like.dtype).to(like.device))
    blk.train()
    x = q(torch.randn(B, T, HW, cin, generator=g), dtype)
    xd = x.to(dtype).to(DEV).requires_grad_(True)
    y = blk(xd)
    dy = q(torch.randn(B, T, HW, cout, generator=g), dtype)
    y.backward(dy.to(dtype).to(DEV))
    # float64 reference of the same realisation
    st = {k: v.detach().double().cpu() for k, v in blk.state_dict().items()}
    if dtype == torch.bfloat16:
        for k in ("conv.weight", "gate.weight", "projection.weight"):
            if k in st:
                st[k] = q(st[k].float(), dtype)
    ref = {k: v.clone().requires_grad_(True) for k, v in st.items()}
    xr = x.double().requires_grad_(True)
    mk = q(mask, dtype)
    xm = xr * mk.unsqueeze(1)
    if dtype == torch.bfloat16:
        xm = xm + (q(xm.detach().float(), dtype) - xm.detach())
    to_nct = lambda a: a.permute(0, 2, 3, 1).reshape(B * HW, a.shape[-1], T)                       # noqa: E731
    out = torch.nn.functional.conv1d(to_nct(xm), ref["conv.weight"], ref["conv.bias"], padding=dil, dilation=dil)
    out = O.group_norm(out, G, ref["norm.weight"], ref["norm.bias"])
    gate = torch.sigmoid(torch.nn.functional.conv1d(out, ref["gate.weight"], ref["gate.bias"]))
    res = to_nct(xr)
    if "projection.weight" in ref:
        res = torch.nn.functional.conv1d(res, ref["projection.weight"], ref["projection.bias"])
    yr = (gate * torch.relu(out) + (1 - gate) * res).reshape(B, HW, cout, T).permute(0, 3, 1, 2)
    yr.backward(dy)
    assert rel_err(y.float(), yr.detach()) <= tol
    assert rel_err(xd.grad.float(), xr.grad) <= 10 * tol
    for name, prm in blk.named_parameters():
        assert rel_err(prm.grad.float().reshape(ref[name].shape), ref[name].grad) <= 10 * tol, name
    blk.eval()
    assert rel_err(blk(xd).float(), y.float().cpu()) > 1e-2                                         # eval mode: no dropout


def test_gated_block_dropout1d_f32_64_channels_is_refused_up_front():
    from frl_hip.models import blocks
    blk = blocks.GatedResidualBlock(64, 64, 3, 1, dropout_rate=0.2, num_groups=8).to(DEV).train()
    with pytest.raises(NotImplementedError):                                                        # before any kernel runs, not in backward
        blk(torch.randn(1, 5, 16, 64, device=DEV))


def test_channel_scale_dropout2d():
    """Dropout2d of the type encoder (conv2d_encoder.py:74,142-148): y = x * scale[b, c], gradient dx = dy * scale[b, c]."""
    from frl_hip import functional as Fh
    g = torch.Generator().manual_seed(5)
    for dtype, tol in ((torch.float32, 1e-6), (torch.bfloat16, 8e-3)):
        x = q(torch.randn(3, 8, 8, 16, generator=g), dtype)
        sc = q((torch.rand(3, 16, generator=g) >= 0.3).float() / 0.7, dtype)
        xd = x.to(dtype).to(DEV).requires_grad_(True)
        y = Fh.ChannelScaleFn.apply(xd, sc.to(dtype).to(DEV))
        assert rel_err(y.float(), x * sc.view(3, 1, 1, 16)) <= tol
        dy = q(torch.randn(3, 8, 8, 16, generator=g), dtype)
        y.backward(dy.to(dtype).to(DEV))
        assert rel_err(xd.grad.float(), dy * sc.view(3, 1, 1, 16)) <= tol


@pytest.mark.parametrize("B,HW,zp", [(2, 1024, 12), (3, 100, 8), (1, 37, 16)])
def test_phase_chain_forward_in_one_launch_equals_the_block_by_block_path(B, HW, zp, monkeypatch):
    """tcn_chain_fwd_kernel: the three hot GatedResidualBlocks (dilation 1, 2, 4) + the 1x1 phase head in one launch.  The blocks hand
    their outputs on in registers, rounded to bf16 exactly where the stand-alone kernels round on their store, so y1, y2, y3 must EQUAL
    the block-by-block launches bit for bit; the head agrees with the 1x1 convolution kernel to bf16 rounding; the backward (head kernels
    + three fused block backward kernels on the saved inputs) yields the gradients of the modular autograd chain."""
    from frl_hip import ops
    from frl_hip import functional as Fh
    from frl_hip.models.blocks import Conv2dParams, TCNEncoder
    torch.manual_seed(HW + zp)
    tcn = TCNEncoder(64, [64, 64, 64], 3, [1, 2, 4], 0.0, 8).to(DEV)
    head = Conv2dParams(64, zp, 1).to(DEV)
    with torch.no_grad():
        for l in tcn.layers:
            l.norm.weight.uniform_(0.5, 1.5)
            l.norm.bias.uniform_(-0.3, 0.3)
    x = torch.randn(B, 5, HW, 64, generator=torch.Generator().manual_seed(3)).to(torch.bfloat16).to(DEV)
    blocks = [(l.conv.weight, l.conv.bias, l.norm.weight, l.norm.bias, l.gate.weight, l.gate.bias, l.dilation, 8, False) for l in tcn.layers]
    assert ops.tcn_chain_supported(x, blocks, head.weight)
    y1, y2, y3, h = ops.tcn_chain_fwd(x, blocks, head.weight, head.bias)
    ref, outs = x, []
    for blk in blocks:
        ref = ops.tcn_block_fwd(ref, *blk[:6], None, None, blk[6], 8)
        outs.append(ref)
    for got, want in zip((y1, y2, y3), outs):
        assert torch.equal(got, want)
    href = ops.conv1x1_fwd(outs[2], head.weight.reshape(zp, 64), head.bias, ops.ACT_NONE)
    assert (h.float() - href.float()).abs().max().item() <= 8e-3 * href.float().abs().max().item()
    # autograd: chain node vs modular modules
    dh = torch.randn(B, 5, HW, zp, generator=torch.Generator().manual_seed(4)).to(torch.bfloat16).to(DEV)
    params = list(tcn.parameters()) + list(head.parameters())
    res = []
    monkeypatch.setenv("FRL_HIP_DISABLE", "headbwd")                 # first with the head's backward-data as its own launch: the modular bits
    for chained in (True, False, "head"):
        if chained == "head":
            monkeypatch.delenv("FRL_HIP_DISABLE")
        for p_ in params:
            p_.grad = None
        xin = x.clone().requires_grad_(True)
        if chained:
            flat = [t for blk in blocks for t in blk[:6]]
            out = Fh.TcnChainHeadFn.apply(xin, *flat, head.weight, head.bias, 8, 1e-5)
        else:
            out = head(tcn(xin))
        out.backward(dh)
        res.append((out.detach(), xin.grad.detach(), [p_.grad.detach().clone() for p_ in params]))
    assert (res[0][0].float() - res[1][0].float()).abs().max().item() <= 8e-3 * res[1][0].float().abs().max().item()
    assert torch.equal(res[0][1], res[1][1])
    for a, b in zip(res[0][2], res[1][2]):
        assert torch.equal(a, b)
    # default route: where the last block's kernel takes dh itself (HW % 64 == 0), dy = dh W_h stays float32 inside the kernel instead of
    # being rounded to bf16 on its way through HBM -- the same gradients to that rounding (2^-9 per element of dy), not the same bits
    head_in_kernel = ops.tcn_block_bwd_head_supported(y2, dh, head.weight.reshape(zp, 64), 4)
    assert head_in_kernel == (HW % 64 == 0)
    if head_in_kernel:
        d = (res[2][1].float() - res[0][1].float()).abs()
        assert d.max().item() <= 1.6e-2 * res[0][1].float().abs().max().item() and d.mean().item() <= 8e-3 * res[0][1].float().abs().mean().item()
        for a, b in zip(res[2][2], res[0][2]):
            assert rel_err(a, b.cpu()) <= 1e-2                       # (measured 2-4e-3: tools/diag/bwd4_head_err.py)
    else:
        assert torch.equal(res[2][1], res[0][1])
        for a, b in zip(res[2][2], res[0][2]):
            assert torch.equal(a, b)
    # the chain's input is data in the model (the tile itself): block 1 then runs the backward without its conv^T GEMM / dx store;
    # parameter gradients are the same bits
    if HW % 64 == 0:
        for p_ in params:
            p_.grad = None
        flat = [t for blk in blocks for t in blk[:6]]
        Fh.TcnChainHeadFn.apply(x, *flat, head.weight, head.bias, 8, 1e-5).backward(dh)
        for p_, b in zip(params, res[2][2]):
            assert torch.equal(p_.grad, b)
        g = ops.tcn_block_bwd(x, y1, *blocks[0][:6], None, None, 1, 8, want_dx=False)
        assert g["dx"] is None
    assert not ops.tcn_chain_supported(x.float(), blocks, head.weight)
    assert not ops.tcn_chain_supported(x, blocks[:2], head.weight)


@pytest.mark.parametrize("B,T,HW", [(2, 5, 1024), (3, 10, 100), (1, 1, 7)])
def test_fused_film_matches_float64(B, T, HW):
    """csrc/film_fused.hip (bf16, 64 -> 32 -> 12): FiLMLayer's two nets + the modulation gamma * h + beta over T as one launch, and its
    backward (d h, all eight parameter gradients contracted in-kernel) against float64 autograd of conditioning.py:82-102 /
    representation.py:369-372 on the same bf16-rounded inputs and weights.  Pixel counts that are not multiples of the 16 / 64-pixel work
    items; T = 1, 5, 10."""
    from frl_hip import ops
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(B * T + HW)
    zt = q(torch.randn(B, HW, 64, generator=g), dt)
    h = q(torch.randn(B, T, HW, 12, generator=g), dt).requires_grad_(True)
    shapes = [(32, 64), (32,), (12, 32), (12,), (32, 64), (32,), (12, 32), (12,)]
    raw = [torch.randn(*s, generator=g) * (0.3 if len(s) == 2 else 0.5) for s in shapes]
    p64 = [(q(t, dt) if t.dim() == 2 else t.double()).requires_grad_(True) for t in raw]    # weights as the matrix cores see them
    w1g, b1g, w2g, b2g, w1b, b1b, w2b, b2b = p64
    gamma = torch.relu(zt @ w1g.T + b1g) @ w2g.T + b2g
    beta = torch.relu(zt @ w1b.T + b1b) @ w2b.T + b2b
    gq, bq = q(gamma.detach(), dt), q(beta.detach(), dt)                # the kernel applies gamma / beta as it stores them (bf16)
    z = gamma.unsqueeze(1) * h + beta.unsqueeze(1)
    pd = [t.detach().float().to(DEV).contiguous() for t in p64]
    ztd, hd = zt.to(dt).to(DEV), h.detach().to(dt).to(DEV)
    assert ops.film_fused_supported(ztd, hd, 32)
    zd, gd, bd = ops.film_fused_fwd(ztd, hd, pd)
    assert rel_err(gd.float(), gamma.detach()) <= 8e-3 and rel_err(bd.float(), beta.detach()) <= 8e-3
    zq = gq.unsqueeze(1) * h.detach() + bq.unsqueeze(1)
    assert rel_err(zd.float(), zq) <= 6e-3
    dz = q(torch.randn(B, T, HW, 12, generator=g), dt)
    z.backward(dz)
    dh, grads = ops.film_fused_bwd(ztd, hd, dz.to(dt).to(DEV), pd)
    assert rel_err(dh.float(), h.grad) <= 1e-2
    for got, ref, name in zip(grads, p64, ["w1g", "b1g", "w2g", "b2g", "w1b", "b1b", "w2b", "b2b"]):
        assert rel_err(got, ref.grad) <= 2e-2, name
    assert not ops.film_fused_supported(ztd.float(), hd.float(), 32) and not ops.film_fused_supported(ztd, hd, 16)


@pytest.mark.parametrize("P,cz,use_mask", [(4096, 64, False), (5000, 12, True), (333, 12, False), (1000, 32, True), (130, 64, True), (70001, 12, True)])
def test_fused_decoder_mse(P, cz, use_mask):
    """Fused decoder + L2 loss (bf16) vs float64 autograd of the same chain, and vs the modular kernels.  Latents of <= 32 channels run the
    backward as two independent 4-wave subgroups per workgroup (70001 rows: more rounds than subgroups, an uneven split and a ragged last
    round); the lockstep kernel must give the same gradients up to float32 summation order."""
    from frl_hip import functional as Fh
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(P + cz)
    z = q(torch.randn(P, cz, generator=g), dtype).requires_grad_(True)
    tgt = q(torch.randn(P, 64, generator=g), dtype)
    w1 = torch.randn(128, cz, generator=g) / cz ** 0.5
    b1 = torch.randn(128, generator=g) * 0.1
    w2 = torch.randn(64, 128, generator=g) / 128 ** 0.5
    b2 = torch.randn(64, generator=g) * 0.1
    w1q, w2q = q(w1, dtype).requires_grad_(True), q(w2, dtype).requires_grad_(True)
    b1d, b2d = b1.double().requires_grad_(True), b2.double().requires_grad_(True)
    mask = (torch.rand(P, generator=g) > 0.25) if use_mask else None
    hid = torch.relu(z @ w1q.t() + b1d)
    hid_q = q(hid.detach(), dtype) + (hid - hid.detach())          # the kernel rounds the hidden activations to bf16
    xh = hid_q @ w2q.t() + b2d
    loss = O.reconstruction_loss_l2(xh, tgt, mask.unsqueeze(1).expand(P, 64) if use_mask else None)
    (0.8 * loss).backward()
    dev = lambda t: t.to(DEV)
    zd = z.detach().to(dtype).to(DEV).requires_grad_(True)
    params = [dev(w1).requires_grad_(True), dev(b1).requires_grad_(True), dev(w2).requires_grad_(True), dev(b2).requires_grad_(True)]
    l, xhat = Fh.decoder_mse(zd, params[0], params[1], params[2], params[3], tgt.to(dtype).to(DEV), mask.to(DEV) if use_mask else None, True)
    assert abs(l.item() - loss.item()) <= 2e-3 * loss.item()
    assert rel_err(xhat.float(), xh.detach()) <= 2e-2
    (0.8 * l).backward()
    assert rel_err(zd.grad.float(), z.grad) <= 3e-2
    for got, ref in zip(params, (w1q, b1d, w2q, b2d)):
        assert rel_err(got.grad, ref.grad) <= 3e-2
    if cz <= 32:
        from frl_hip import _lib
        lib = _lib.load()
        first = [zd.grad.clone()] + [p_.grad.clone() for p_ in params]
        zd.grad = None
        for p_ in params:
            p_.grad = None
        was = lib.frl_decoder_mse_bwd_subgroups(0)
        try:
            l2, _ = Fh.decoder_mse(zd, params[0], params[1], params[2], params[3], tgt.to(dtype).to(DEV), mask.to(DEV) if use_mask else None, True)
            (0.8 * l2).backward()
        finally:
            lib.frl_decoder_mse_bwd_subgroups(was)
        assert torch.equal(first[0], zd.grad)                          # dz does not depend on who computed the row
        for a, p_ in zip(first[1:], params):
            assert rel_err(a, p_.grad.cpu()) <= 1e-4


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-6), (torch.bfloat16, 8e-3)])
def test_backward_epilogue_sums_match_separate_launches(dtype, tol):
    """conv3x3 / conv1x1 bwd-data and the Sobel transpose with the accumulation of a second gradient stream (and, for the 3x3, the
    second output sub_from - dx) in their epilogues == the plain launch followed by torch arithmetic.  Shapes cover the 16-byte store
    route (64 channels) and the scalar route."""
    from frl_hip import ops
    from frl_hip.ops import ACT_NONE, ACT_RELU
    g = torch.Generator().manual_seed(77)

    def rnd(*shape):
        return torch.randn(*shape, generator=g).to(dtype).to(DEV)

    for (B, H, W, cin, cout) in [(2, 16, 32, 64, 64), (1, 9, 11, 12, 8), (1, 8, 8, 128, 64)]:
        w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.1).to(DEV)
        dy, y, add, sub = rnd(B, H, W, cout), rnd(B, H, W, cout), rnd(B, H, W, cin), rnd(B, H, W, cin)
        plain = ops.conv3x3_bwd_data(dy, w, y, ACT_RELU).float()
        dx = ops.conv3x3_bwd_data(dy, w, y, ACT_RELU, add=add)
        scale = plain.abs().max().item() + add.float().abs().max().item()
        assert (dx.float() - (plain + add.float())).abs().max().item() <= tol * scale
        dx2, out2 = ops.conv3x3_bwd_data(dy, w, y, ACT_RELU, add=add, sub_from=sub)
        assert torch.equal(dx2, dx)
        assert (out2.float() - (sub.float() - dx.float())).abs().max().item() <= tol * scale
        dx3, out3 = ops.conv3x3_bwd_data(dy, w, y, ACT_RELU, sub_from=sub)
        assert (dx3.float() - plain).abs().max().item() <= tol * scale
        assert (out3.float() - (sub.float() - dx3.float())).abs().max().item() <= tol * scale
    # gate blend in the epilogue of the sigmoid convolution == the convolution followed by the stand-alone blend kernel (min_gate = 0), bit for bit
    for (B, H, W, cin, cout) in [(2, 32, 32, 64, 64), (1, 9, 11, 12, 8)]:
        w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.1).to(DEV)
        bias = torch.randn(cout, generator=g).to(DEV)
        x, sm, res = rnd(B, H, W, cin), rnd(B, H, W, cout), rnd(B, H, W, cout)
        graw = ops.conv3x3_fwd(x, w, bias, ops.ACT_SIGMOID)
        out_ref, gate_ref = ops.gate_blend_fwd(sm, res, graw, 0.0)
        out, gate = ops.conv3x3_fwd_gate_blend(x, w, bias, sm, res)
        assert torch.equal(gate, graw) and torch.equal(gate, gate_ref) and torch.equal(out, out_ref)
    for (P, cin, cout) in [(4096, 64, 32), (333, 12, 20), (1000, 64, 256), (130, 8, 4)]:
        w = (torch.randn(cout, cin, generator=g) * 0.2).to(DEV)
        dy, add = rnd(P, cout), rnd(P, cin)
        plain = ops.conv1x1_bwd_data(dy, w, None, ACT_NONE).float()
        dx = ops.conv1x1_bwd_data(dy, w, None, ACT_NONE, add=add)
        assert (dx.float() - (plain + add.float())).abs().max().item() <= tol * (plain.abs().max().item() + 4.0)
    for (B, H, W, C) in [(2, 16, 16, 64), (1, 7, 9, 8)]:
        dg, add = rnd(B, H, W, 2 * C), rnd(B, H, W, C)
        plain = ops.sobel_bwd(dg).float()
        assert (ops.sobel_bwd(dg, add=add).float() - (plain + add.float())).abs().max().item() <= tol * (plain.abs().max().item() + 4.0)
    with pytest.raises(ValueError):
        ops.sobel_bwd(rnd(1, 4, 4, 16), add=rnd(1, 4, 4, 16))                 # add must have dx's shape, not dg's


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 4e-2)])
@pytest.mark.parametrize("B,H,W,C,hidden", [(2, 32, 32, 64, 64), (1, 9, 13, 16, 24)])
def test_spatial_smoothing_block_as_one_autograd_node_matches_the_modular_chain(dtype, tol, B, H, W, C, hidden):
    """EdgeAwareSmoothingConv2D: fuse=True (Fh.SpatialSmoothFn, gradient sums in kernel epilogues) against fuse=False (one autograd
    node per kernel, sums by autograd).  Outside the hot configuration the forward kernels are the same launches, so outputs are
    bit-equal; gradients agree to the rounding of the intermediate sums."""
    from frl_hip.models.blocks import EdgeAwareSmoothingConv2D
    torch.manual_seed(5 + C)
    m = EdgeAwareSmoothingConv2D(C, gate_hidden=hidden).to(DEV)
    m.set_min_gate(0.1)
    x0 = (torch.randn(B, H, W, C, generator=torch.Generator().manual_seed(9)) * 0.7).to(dtype).to(DEV)
    dout = torch.randn(B, H, W, C, generator=torch.Generator().manual_seed(10)).to(dtype).to(DEV)
    dgate = (torch.randn(B, H, W, C, generator=torch.Generator().manual_seed(11)) * 0.3).to(dtype).to(DEV)
    res = {}

    def run(fuse, dt):
        m.fuse = fuse
        m.zero_grad(set_to_none=True)
        x = x0.to(dt).clone().requires_grad_(True)
        out, gate = m(x, return_gate=True)
        torch.autograd.backward([out, gate], [dout.to(dt), dgate.to(dt)])
        return (out.detach().float(), gate.detach().float(), x.grad.detach().float(), {n: p.grad.detach().clone() for n, p in m.named_parameters()})

    for fuse in (False, True):
        res[fuse] = run(fuse, dtype)
    hot = dtype == torch.bfloat16 and C == 64 and hidden == 64
    if hot:
        # Hot configuration: the fused node runs heads + softmaxes + bank as one kernel per direction (float32 logits that never leave
        # the registers; heads recomputed in the backward), the modular chain rounds logits, soft-maxed maps and their gradients to bf16
        # in between.  Both are bf16 evaluations of the same function: each is held against the float32 evaluation of the modular chain on
        # the same (bf16-representable) inputs, and the fused node may not be further from it than 1.5 x the modular chain's own distance (both sit at 4-5 % on the gate-net bias sums).
        ref = run(False, torch.float32)

        def err(a, b):
            return (a.float() - b.float()).abs().max().item() / max(b.float().abs().max().item(), 1e-6)

        for i, name in ((0, "out"), (1, "gate"), (2, "dx")):
            ef, em = err(res[True][i], ref[i]), err(res[False][i], ref[i])
            assert ef <= max(1.5 * em, 1e-2) and ef <= tol, (name, ef, em)
        for n, gr in ref[3].items():
            ef, em = err(res[True][3][n], gr), err(res[False][3][n], gr)
            assert ef <= max(1.5 * em, 3e-2) and ef <= 2.5 * tol, (n, ef, em)
    else:
        assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
        gx_f, gx_m = res[True][2], res[False][2]
        assert (gx_f - gx_m).abs().max().item() <= tol * gx_m.abs().max().item()
        for n, gm in res[False][3].items():
            gf = res[True][3][n]
            assert (gf - gm).abs().max().item() <= tol * max(gm.abs().max().item(), 1e-6), n
    # the gate output may carry no gradient at all (return_gate=False callers)
    m.fuse = True
    x = x0.clone().requires_grad_(True)
    m(x).backward(dout)
    m.fuse = False
    x2 = x0.clone().requires_grad_(True)
    m(x2).backward(dout)
    assert (x.grad.float() - x2.grad.float()).abs().max().item() <= tol * x2.grad.float().abs().max().item()
