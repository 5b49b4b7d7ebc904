"""torch.autograd.Function wrappers: PyTorch carries the autograd graph, every forward/backward body is a HIP kernel.

Tensors are NHWC rows ([..., C] contiguous) in the compute dtype (float32 parity mode / bfloat16 performance mode);
parameters stay float32 in the reference's layouts and receive float32 gradients.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import ops
from .ops import ACT_NONE, ACT_RELU, ACT_SIGMOID  # noqa: F401


def _c(t: Optional[torch.Tensor]):
    return None if t is None else (t if t.is_contiguous() else t.contiguous())


class Conv1x1Fn(Function):
    """y = act(x W^T + b) over rows; W [Cout, Cin(,1,1)]."""

    @staticmethod
    def forward(ctx, x, w, bias, act):
        y = ops.conv1x1_fwd(x, w, bias, act)
        ctx.act = act
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, w, y if act != ACT_NONE else None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        dy = _c(dy)
        dx = ops.conv1x1_bwd_data(dy, w, y, ctx.act) if ctx.needs_input_grad[0] else None
        dw = db = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw, db = ops.conv1x1_bwd_weight(dy, x, y, ctx.act, want_bias=ctx.has_bias)
            dw = dw.reshape(w.shape)
        return dx, dw, db, None


class Conv3x3Fn(Function):
    """y = act(conv3x3_pad1(x, W) + b); x [B,H,W,Cin], W [Cout,Cin,3,3]."""

    @staticmethod
    def forward(ctx, x, w, bias, act):
        y = ops.conv3x3_fwd(x, w, bias, act)
        ctx.act = act
        ctx.save_for_backward(x, w, y if act != ACT_NONE else None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        dy = _c(dy)
        dx = ops.conv3x3_bwd_data(dy, w, y, ctx.act) if ctx.needs_input_grad[0] else None
        dw, db = ops.conv3x3_bwd_weight(dy, x, y, ctx.act)
        return dx, dw, db, None


class GroupNormFn(Function):
    """nn.GroupNorm over [B, ..., C] rows with optional fused ReLU."""

    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps, relu):
        y, mean, rstd = ops.groupnorm_fwd(x, gamma, beta, groups, eps, relu)
        ctx.groups, ctx.relu = groups, relu
        ctx.save_for_backward(x, gamma, beta, mean, rstd)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, gamma, beta, mean, rstd = ctx.saved_tensors
        dx, dg, db = ops.groupnorm_bwd(_c(dy), x, gamma, beta, mean, rstd, ctx.groups, ctx.relu)
        return dx, dg, db, None, None, None


class Encoder2Fn(Function):
    """Fused conv1x1 -> GroupNorm -> ReLU -> conv1x1 -> GroupNorm over the rows of each sample (csrc/enc_fused.hip); x is data (no dx)."""

    @staticmethod
    def forward(ctx, x, w1, g1, b1, w2, g2, b2, eps):
        z, stats = ops.encoder2_fwd(x, w1, g1, b1, w2, g2, b2, eps)
        ctx.save_for_backward(x, w1, g1, b1, w2, g2, b2, stats)
        return z

    @staticmethod
    @once_differentiable
    def backward(ctx, dz):
        x, w1, g1, b1, w2, g2, b2, stats = ctx.saved_tensors
        dw1, dg1, db1, dw2, dg2, db2 = ops.encoder2_bwd(x, _c(dz), w1, g1, b1, w2, g2, b2, stats)
        return None, dw1, dg1, db1, dw2, dg2, db2, None


class ScalarCombineFn(Function):
    """sum_i coef_i * term_i over device scalars in one launch (plus its finite flag and, optionally, a second un-differentiated
    combination of the same terms); backward: one launch for all term gradients."""

    @staticmethod
    def forward(ctx, coefs, mults, aux_coefs, *terms):
        res = ops.scalar_combine([t.detach() for t in terms], coefs, mults, aux_coefs)
        ctx.coefs = tuple(float(c) for c in coefs)
        ctx.mults = mults                                           # device scalars (no gradient: schedule values), read again in the backward
        ctx.mark_non_differentiable(*res[1:])
        return res if aux_coefs is not None else res + (None,)

    @staticmethod
    @once_differentiable
    def backward(ctx, g, g_ok, g_aux):
        gg = ops.scalar_fanout(g.reshape(1).float().contiguous(), ctx.coefs, ctx.mults)
        return (None, None, None) + tuple(gg[i] for i in range(len(ctx.coefs)))


def scalar_combine(terms, coefs, mults=None, aux_coefs=None):
    """-> (sum_i coef_i * mult_i * term_i [0-dim], ok [1] = 1.0 if finite else 0.0 | None[, aux = sum_i aux_coef_i * term_i, detached]); plain
    torch arithmetic off the GPU path.  mults: optional device float scalars (or None entries) that scale a term's weight at run time (a
    scheduled loss weight); aux_coefs: a second combination of the same terms from the same launch (no gradient flows through it)."""
    fits = all(torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 and t.numel() == 1 for t in terms) and 1 <= len(terms) <= 8
    if not fits:
        tot = None
        for i, (t, c) in enumerate(zip(terms, coefs)):
            v = t if c == 1.0 else c * t
            if mults is not None and mults[i] is not None:
                v = v * mults[i].reshape(()).to(v.device)
            tot = v if tot is None else tot + v
        if aux_coefs is None:
            return tot, None
        aux = sum(float(c) * t.detach() for t, c in zip(terms, aux_coefs))
        return tot, None, aux
    loss, ok, aux = ScalarCombineFn.apply(tuple(coefs), None if mults is None else tuple(mults), None if aux_coefs is None else tuple(aux_coefs),
                                          *[t.reshape(()) for t in terms])
    return (loss, ok) if aux_coefs is None else (loss, ok, aux)


class SobelFn(Function):
    @staticmethod
    def forward(ctx, x):
        return ops.sobel_fwd(x)

    @staticmethod
    @once_differentiable
    def backward(ctx, dg):
        return ops.sobel_bwd(_c(dg))


class EdgeSmoothFn(Function):
    """(x, a_logit, b_logit) -> (smoothed, residual = x - smoothed)."""

    @staticmethod
    def forward(ctx, x, a_logit, b_logit, rank, dil):
        sm, res, a_soft, b_soft = ops.edge_smooth_fwd(x, a_logit, b_logit, rank, dil)
        ctx.rank, ctx.dil = rank, dil
        ctx.save_for_backward(x, a_soft, b_soft)
        return sm, res

    @staticmethod
    @once_differentiable
    def backward(ctx, d_sm, d_res):
        x, a_soft, b_soft = ctx.saved_tensors
        d_sm, d_res = _c(d_sm), _c(d_res)
        d_tot = ops.add(d_sm, d_res, -1.0)                # residual = x - smoothed
        dx, da, db = ops.edge_smooth_bwd(d_tot, x, a_soft, b_soft, ctx.rank, ctx.dil, dx_add=d_res)   # + d_res: x feeds the residual directly
        return dx, da, db, None, None


class GateBlendFn(Function):
    """(smoothed, residual, gate_raw) -> (out = smoothed + clamp(gate) * residual, gate)."""

    @staticmethod
    def forward(ctx, smoothed, residual, gate_raw, min_gate):
        out, gate = ops.gate_blend_fwd(smoothed, residual, gate_raw, min_gate)
        ctx.min_gate = min_gate
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(residual, gate_raw)
        return out, gate

    @staticmethod
    @once_differentiable
    def backward(ctx, dout, dgate):
        residual, gate_raw = ctx.saved_tensors
        if dout is None:
            dout = torch.zeros_like(residual)
        dout = _c(dout)
        dres, dgraw = ops.gate_blend_bwd(dout, _c(dgate), residual, gate_raw, ctx.min_gate)
        return dout, dres, dgraw, None


class SpatialSmoothFn(Function):
    """EdgeAwareSmoothingConv2D.forward (spatial.py:278-339) as ONE autograd node over the same kernels as the modular chain
    SobelFn -> conv3x3 -> two 1x1 heads -> EdgeSmoothFn -> conv3x3 x2 -> GateBlendFn.  The point is the backward: three tensors of the
    block have two consumers each (x: Sobel + filter bank; feat: the two heads; residual: gate net + blend) and autograd sums their
    gradients with separate elementwise launches; here each sum rides in the epilogue of the kernel that produces its second term,
    and d_smoothed - d_residual leaves the gate net's bwd-data launch as a second output."""

    @staticmethod
    def forward(ctx, x, w_mb, b_mb, w_a, b_a, w_b, b_b, w_g0, b_g0, w_g2, b_g2, rank, dil, min_gate):
        g = ops.sobel_fwd(x)
        feat = ops.conv3x3_fwd(g, w_mb, b_mb, ACT_RELU)
        # hot configuration: heads + softmaxes + bank in ONE kernel, nothing of the [P,288] logits reaches memory (csrc/smooth_fused.hip);
        # the backward recomputes the heads from feat
        fused_heads = ops.smooth_heads_supported(x, feat.shape[-1], rank, w_a, b_a, w_b, b_b)
        if fused_heads:
            sm, res = ops.smooth_heads_fwd(x, feat, w_a, b_a, w_b, b_b, dil)
            a_soft = b_soft = None
        else:
            a_logit = ops.conv1x1_fwd(feat, w_a, b_a, ACT_NONE)
            b_logit = ops.conv1x1_fwd(feat, w_b, b_b, ACT_NONE)
            sm, res, a_soft, b_soft = ops.edge_smooth_fwd(x, a_logit, b_logit, rank, dil)
        g1 = ops.conv3x3_fwd(res, w_g0, b_g0, ACT_RELU)
        if min_gate <= 0.0 and b_g2 is not None and ops.fusion_enabled("blend"):
            out, gate_raw = ops.conv3x3_fwd_gate_blend(g1, w_g2, b_g2, sm, res)      # blend in the convolution's epilogue; no floor: gate IS gate_raw
            gate = gate_raw
        else:
            gate_raw = ops.conv3x3_fwd(g1, w_g2, b_g2, ACT_SIGMOID)
            out, gate = ops.gate_blend_fwd(sm, res, gate_raw, min_gate)
        ctx.cfg = (rank, dil, min_gate)
        ctx.has_bias = tuple(b is not None for b in (b_mb, b_a, b_b, b_g0, b_g2))
        ctx.set_materialize_grads(False)
        ctx.fused_heads = fused_heads
        ctx.save_for_backward(x, g, feat, a_soft, b_soft, res, g1, gate_raw, w_mb, w_a, w_b, w_g0, w_g2, b_a if fused_heads else None,
                              b_b if fused_heads else None)
        return out, gate

    @staticmethod
    @once_differentiable
    def backward(ctx, dout, dgate):
        x, g, feat, a_soft, b_soft, res, g1, gate_raw, w_mb, w_a, w_b, w_g0, w_g2, b_a, b_b = ctx.saved_tensors
        rank, dil, min_gate = ctx.cfg
        dout = torch.zeros_like(res) if dout is None else _c(dout)
        # Activation masks of the three 3x3 convolutions are folded into the kernels that PRODUCE their incoming gradients (gate-blend backward:
        # sigmoid' of the gate; epilogue of the gate net's second backward-data: relu' of g1; mixing-heads backward: relu' of feat), so the six
        # 3x3 backward calls run without a mask pass over their input (16-35 us per backward-data, 6-15 us per weight gradient)
        pre = ops.fusion_enabled("premask")
        dres, dgraw = ops.gate_blend_bwd(dout, _c(dgate), res, gate_raw, min_gate, sigmoid_mask=pre)
        a2, y2 = (ACT_NONE, None) if pre else (ACT_SIGMOID, gate_raw)
        if pre:
            dg1 = ops.conv3x3_bwd_data(dgraw, w_g2, None, ACT_NONE, out_y=g1, out_act=ACT_RELU)
        else:
            dg1 = ops.conv3x3_bwd_data(dgraw, w_g2, gate_raw, ACT_SIGMOID)
        dw_g2, db_g2 = ops.conv3x3_bwd_weight(dgraw, g1, y2, a2)
        a0, y0 = (ACT_NONE, None) if pre else (ACT_RELU, g1)
        # residual: blend term + gate-net term in one store; d(smoothed) - d(residual) (residual = x - smoothed) as the second output
        dres_tot, d_tot = ops.conv3x3_bwd_data(dg1, w_g0, y0, a0, add=dres, sub_from=dout)
        dw_g0, db_g0 = ops.conv3x3_bwd_weight(dg1, res, y0, a0)
        pre_f = pre and ctx.fused_heads
        if ctx.fused_heads:
            dx, dfeat, dw_a, db_a, dw_b, db_b = ops.smooth_heads_bwd(d_tot, x, feat, w_a, b_a, w_b, b_b, dil, dx_add=dres_tot, dfeat_relu=pre_f)
        else:
            dx, da, db = ops.edge_smooth_bwd(d_tot, x, a_soft, b_soft, rank, dil, dx_add=dres_tot)
            dfeat = ops.conv1x1_bwd_data(db, w_b, None, ACT_NONE, add=ops.conv1x1_bwd_data(da, w_a, None, ACT_NONE))
            dw_a, db_a = ops.conv1x1_bwd_weight(da, feat, None, ACT_NONE, want_bias=ctx.has_bias[1])
            dw_b, db_b = ops.conv1x1_bwd_weight(db, feat, None, ACT_NONE, want_bias=ctx.has_bias[2])
        af, yf = (ACT_NONE, None) if pre_f else (ACT_RELU, feat)
        dw_mb, db_mb = ops.conv3x3_bwd_weight(dfeat, g, yf, af)
        if ctx.needs_input_grad[0]:
            dx = ops.sobel_bwd(ops.conv3x3_bwd_data(dfeat, w_mb, yf, af), add=dx)
        else:
            dx = None
        hb = ctx.has_bias
        return (dx, dw_mb, db_mb if hb[0] else None, dw_a.reshape(w_a.shape), db_a if hb[1] else None, dw_b.reshape(w_b.shape),
                db_b if hb[2] else None, dw_g0, db_g0 if hb[3] else None, dw_g2, db_g2 if hb[4] else None, None, None, None)


class TcnBlockFn(Function):
    """Fused GatedResidualBlock on x [B,T,HW..,Cin]."""

    @staticmethod
    def forward(ctx, x, conv_w, conv_b, gn_w, gn_b, gate_w, gate_b, proj_w, proj_b, dilation, groups, eps, drop_mask=None):
        pw = None if proj_w is None else proj_w.reshape(proj_w.shape[0], proj_w.shape[1])
        y = ops.tcn_block_fwd(x, conv_w, conv_b, gn_w, gn_b, gate_w, gate_b, pw, proj_b, dilation, groups, eps, drop_mask=drop_mask)
        ctx.cfg = (dilation, groups, eps)
        ctx.drop_mask = drop_mask
        ctx.save_for_backward(x, conv_w, conv_b, gn_w, gn_b, gate_w, gate_b, proj_w, proj_b)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, conv_w, conv_b, gn_w, gn_b, gate_w, gate_b, proj_w, proj_b = ctx.saved_tensors
        dilation, groups, eps = ctx.cfg
        pw = None if proj_w is None else proj_w.reshape(proj_w.shape[0], proj_w.shape[1])
        g = ops.tcn_block_bwd(x, _c(dy), conv_w, conv_b, gn_w, gn_b, gate_w, gate_b, pw, proj_b, dilation, groups, eps,
                              drop_mask=ctx.drop_mask, want_dx=ctx.needs_input_grad[0])
        dpw = g["proj_w"].reshape(proj_w.shape) if proj_w is not None else None
        return (g["dx"] if ctx.needs_input_grad[0] else None, g["conv_w"], g["conv_b"], g["gn_w"], g["gn_b"], g["gate_w"],
                g["gate_b"], dpw, g.get("proj_b"), None, None, None, None)


class TcnChainHeadFn(Function):
    """Three hot GatedResidualBlocks (dilation 1, 2, 4) + the 1x1 phase head: ONE forward launch (ops.tcn_chain_fwd); the backward runs the
    head's two kernels and the three fused block backward kernels on the saved block inputs x, y1, y2 (and y3 for the head's weights).
    args: x, 3 x (conv_w, conv_b, gn_w, gn_b, gate_w, gate_b), head_w, head_b, groups, eps."""

    @staticmethod
    def forward(ctx, x, *args):
        params, (head_w, head_b, groups, eps) = args[:18], args[18:]
        blocks = [tuple(params[6 * i:6 * i + 6]) + (d, groups, False) for i, d in enumerate((1, 2, 4))]
        y1, y2, y3, h = ops.tcn_chain_fwd(x, blocks, head_w, head_b, eps)
        ctx.cfg = (groups, eps)
        ctx.save_for_backward(x, y1, y2, y3, head_w, *params)
        return h

    @staticmethod
    @once_differentiable
    def backward(ctx, dh):
        x, y1, y2, y3, head_w, *params = ctx.saved_tensors
        groups, eps = ctx.cfg
        dh = _c(dh)
        w2 = head_w.reshape(head_w.shape[0], head_w.shape[1])
        dw_h, db_h = ops.conv1x1_bwd_weight(dh, y3, None, ACT_NONE, want_bias=True)
        # the head's backward-data rides inside the last block's backward kernel where that kernel applies (dy = dh W_h never reaches HBM)
        head_in_kernel = groups == 8 and ops.tcn_block_bwd_head_supported(y2, dh, w2, 4)
        dy = None if head_in_kernel else ops.conv1x1_bwd_data(dh, w2, None, ACT_NONE)
        grads = [None] * 18
        for i, (xin, dil) in reversed(list(enumerate(zip((x, y1, y2), (1, 2, 4))))):
            cw, cb, gw, gb, tw, tb = params[6 * i:6 * i + 6]
            if i == 2 and head_in_kernel:
                g = ops.tcn_block_bwd_head(xin, dh, w2, cw, cb, gw, gb, tw, tb, dil, eps)
            else:
                g = ops.tcn_block_bwd(xin, dy, cw, cb, gw, gb, tw, tb, None, None, dil, groups, eps, want_dx=(i > 0 or ctx.needs_input_grad[0]))
            grads[6 * i:6 * i + 6] = [g["conv_w"], g["conv_b"], g["gn_w"], g["gn_b"], g["gate_w"], g["gate_b"]]
            dy = g["dx"]
        return (dy if ctx.needs_input_grad[0] else None,) + tuple(grads) + (dw_h.reshape(head_w.shape), db_h, None, None)


class FilmFn(Function):
    """z[b,t,p,c] = gamma[b,p,c] * h[b,t,p,c] + beta[b,p,c]."""

    @staticmethod
    def forward(ctx, h, gamma, beta):
        ctx.save_for_backward(h, gamma)
        return ops.film_modulate_fwd(h, gamma, beta)

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        h, gamma = ctx.saved_tensors
        return ops.film_modulate_bwd(_c(dout), h, gamma)


class FilmFusedFn(Function):
    """FiLMLayer + modulation in one launch per direction (csrc/film_fused.hip): (h [B,T,HW..,12], z_type [B,HW..,64] stop-gradient,
    the eight parameters of gamma_network / beta_network) -> (z = gamma * h + beta, gamma, beta)."""

    @staticmethod
    def forward(ctx, h, z_type, *params):
        z, gamma, beta = ops.film_fused_fwd(z_type, h, params)
        ctx.save_for_backward(h, z_type, *params)
        ctx.mark_non_differentiable(gamma, beta)
        return z, gamma, beta

    @staticmethod
    @once_differentiable
    def backward(ctx, dz, _dg, _db):
        h, z_type, *params = ctx.saved_tensors
        dh, grads = ops.film_fused_bwd(z_type, h, _c(dz), params)
        return (dh if ctx.needs_input_grad[0] else None, None) + tuple(grads)


class ChannelScaleFn(Function):
    """y[b, r, c] = x[b, r, c] * scale[b, c]  (Dropout2d mask on NHWC rows; the FiLM kernel with a single 'pixel' per sample)."""

    @staticmethod
    def forward(ctx, x, scale):
        b, c = scale.shape
        ctx.save_for_backward(scale)
        zero = torch.zeros_like(scale)
        ctx.zero = zero
        return ops.film_modulate_fwd(x.reshape(b, -1, 1, c), scale.reshape(b, 1, c), zero.reshape(b, 1, c)).reshape(x.shape)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (scale,) = ctx.saved_tensors
        b, c = scale.shape
        dy = _c(dy)
        return ops.film_modulate_fwd(dy.reshape(b, -1, 1, c), scale.reshape(b, 1, c), ctx.zero.reshape(b, 1, c)).reshape(dy.shape), None


class VQFn(Function):
    """(z [N,d], codebook [K,d]) -> (z_q [straight-through], L_codebook, L_commit, perplexity, idx, counts, stats).

    L_codebook = mean((sg[z] - z_q)^2) and L_commit = mean((z - sg[z_q])^2) are numerically equal in the forward;
    they are separate outputs so that autograd routes their upstream gradients to the codebook and to z respectively.
    """

    @staticmethod
    def forward(ctx, z, codebook, prep=None):
        idx, zq, stats, counts = ops.vq_assign(z, codebook, prep)
        d = z.shape[-1]
        n = z.numel() // d
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(z, codebook, idx, counts, zq)
        ctx.mark_non_differentiable(idx, counts)
        # stats = {sum ||z - z_q||^2, perplexity, rows re-evaluated, mean squared error}: the two loss terms are numerically equal
        mse = stats.narrow(0, 3, 1).reshape(())
        ctx.mark_non_differentiable(stats)
        return zq, mse, mse.clone(), stats.narrow(0, 1, 1).reshape(()), idx, counts, stats

    @staticmethod
    @once_differentiable
    def backward(ctx, g_zq, g_cb, g_cm, g_perp, g_idx, g_counts, g_stats):
        z, codebook, idx, counts, zq = ctx.saved_tensors
        if g_cm is not None and g_cb is not None:
            gs = torch.stack([g_cm.reshape(()).float(), g_cb.reshape(()).float()])      # one launch
        else:
            gs = torch.zeros(2, dtype=torch.float32, device=z.device)
            if g_cm is not None:
                gs[0] = g_cm
            if g_cb is not None:
                gs[1] = g_cb
        gz, ge, _ = ops.vq_bwd(_c(g_zq), z, codebook, idx, counts, gs, 1.0,
                               want_gz=ctx.needs_input_grad[0], want_ge=ctx.needs_input_grad[1], zq=zq)
        return gz, ge, None


class MseFn(Function):
    """Masked mean squared error -> f32 scalar tensor."""

    @staticmethod
    def forward(ctx, pred, target, mask):
        stats = ops.mse_fwd(pred, target, mask)
        ctx.save_for_backward(pred, target, mask, stats)
        return stats[0].clone()

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        pred, target, mask, stats = ctx.saved_tensors
        return ops.mse_bwd(pred, target, mask, g.reshape(1).float().contiguous(), stats), None, None


class DecoderMseFn(Function):
    """Fused conv1x1 -> ReLU -> conv1x1 -> masked L2 loss: (z, W1, b1, W2, b2, target, mask) -> (loss, xhat | None)."""

    @staticmethod
    def forward(ctx, z, w1, b1, w2, b2, target, mask, want_xhat):
        stats, xhat = ops.decoder_mse_fwd(z, w1, b1, w2, b2, target, mask, want_xhat)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(z, w1, b1, w2, b2, target, mask, stats)
        if xhat is not None:
            ctx.mark_non_differentiable(xhat)
        return stats.narrow(0, 0, 1).reshape(()), xhat                 # (a view of the kernel's output record: no copy launch)

    @staticmethod
    @once_differentiable
    def backward(ctx, g, g_xhat):
        z, w1, b1, w2, b2, target, mask, stats = ctx.saved_tensors
        if g is None:
            return (None,) * 8
        dz, dw1, db1, dw2, db2 = ops.decoder_mse_bwd(z, w1, b1, w2, b2, target, mask, g.reshape(1).float().contiguous(), stats)
        return dz, dw1.reshape(w1.shape), db1, dw2.reshape(w2.shape), db2, None, None, None


def decoder_mse(z, w1, b1, w2, b2, target, mask=None, want_xhat=False):
    if mask is not None:
        mask = mask.reshape(-1).to(torch.uint8).contiguous()
    return DecoderMseFn.apply(z, w1, b1, w2, b2, target, mask, want_xhat)


def conv1x1(x, w, bias=None, act=ACT_NONE):
    return Conv1x1Fn.apply(x, w, bias, act)


def conv3x3(x, w, bias=None, act=ACT_NONE):
    return Conv3x3Fn.apply(x, w, bias, act)


def encoder2(x, w1, g1, b1, w2, g2, b2, eps=1e-5):
    return Encoder2Fn.apply(x, w1, g1, b1, w2, g2, b2, eps)


def group_norm(x, gamma, beta, groups, eps=1e-5, relu=False):
    return GroupNormFn.apply(x, gamma, beta, groups, eps, relu)


def mse_loss(pred, target, mask=None):
    """mask: [rows] / [B,H,W] bool or uint8 (True = valid) broadcast over channels, or None."""
    if mask is not None:
        mask = mask.reshape(-1).to(torch.uint8).contiguous()
    return MseFn.apply(pred, target, mask)
