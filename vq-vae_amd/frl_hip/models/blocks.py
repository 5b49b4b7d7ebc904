"""Building blocks with the reference's module / parameter names, computing through HIP kernels (NHWC rows).

Each class mirrors one reference module so that `state_dict()` keys and tensor shapes are identical and reference
checkpoints load unchanged:
  Conv2DEncoder            frl/models/conv2d_encoder.py:19-159
  EdgeAwareSmoothingConv2D frl/models/spatial.py:165-343
  GatedResidualBlock/TCNEncoder  frl/models/tcn.py:24-302
  FiLMLayer                frl/models/conditioning.py:16-102
  Conv2DHead (decoder)     frl/models/heads.py:128-198
All `forward` methods take and return NHWC tensors in the compute dtype; layout conversion from the reference's
NCHW API happens once at the model boundary (models/representation.py).
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Union

import torch
import torch.nn as nn

from .. import functional as Fh
from .. import ops
from ..functional import ACT_NONE, ACT_RELU, ACT_SIGMOID


def _kaiming_uniform_(w: torch.Tensor, fan_in: int):
    bound = 1.0 / math.sqrt(fan_in) if fan_in > 0 else 0.0  # kaiming_uniform_(a=sqrt(5)) == U(+-1/sqrt(fan_in))
    with torch.no_grad():
        w.uniform_(-bound, bound)


class _Marker(nn.Module):
    """Parameter-free placeholder that keeps nn.Sequential indices aligned with the reference (ReLU / Sigmoid / Dropout)."""

    def __init__(self, kind: str, p: float = 0.0):
        super().__init__()
        self.kind, self.p = kind, p

    def extra_repr(self):
        return f"{self.kind}" + (f", p={self.p}" if self.kind.startswith("dropout") else "")


class Conv2dParams(nn.Module):
    """Holds nn.Conv2d-shaped parameters (weight [Co,Ci,k,k], optional bias); compute is done by the parent."""

    def __init__(self, cin: int, cout: int, k: int = 1, bias: bool = True):
        super().__init__()
        self.in_channels, self.out_channels, self.k = cin, cout, k
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        self.bias = nn.Parameter(torch.empty(cout)) if bias else None
        _kaiming_uniform_(self.weight, cin * k * k)
        if bias:
            _kaiming_uniform_(self.bias, cin * k * k)

    def forward(self, x, act: int = ACT_NONE):
        if self.k == 1:
            return Fh.conv1x1(x, self.weight, self.bias, act)
        if self.k == 3:
            return Fh.conv3x3(x, self.weight, self.bias, act)
        raise NotImplementedError("only 1x1 and 3x3 (pad 1) convolutions are on the hot path")


class Conv1dParams(nn.Module):
    """nn.Conv1d-shaped parameters (weight [Co,Ci,k], bias)."""

    def __init__(self, cin: int, cout: int, k: int):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k))
        self.bias = nn.Parameter(torch.empty(cout))
        _kaiming_uniform_(self.weight, cin * k)
        _kaiming_uniform_(self.bias, cin * k)


class GroupNormParams(nn.Module):
    def __init__(self, groups: int, channels: int, eps: float = 1e-5):
        super().__init__()
        self.num_groups, self.num_channels, self.eps = groups, channels, eps
        self.weight = nn.Parameter(torch.ones(channels))
        self.bias = nn.Parameter(torch.zeros(channels))

    def forward(self, x, relu: bool = False):
        return Fh.group_norm(x, self.weight, self.bias, self.num_groups, self.eps, relu)


def dropout_mask(shape, p: float, like: torch.Tensor) -> torch.Tensor:
    """Bernoulli keep-mask scaled by 1/(1-p) (torch.nn.Dropout*d semantics), in the dtype / on the device of `like`."""
    keep = torch.rand(shape, device=like.device) >= p
    return (keep.float() / (1.0 - p)).to(like.dtype)


def channel_dropout(x: torch.Tensor, p: float, training: bool, group_rows: int) -> torch.Tensor:
    """Dropout2d semantics on NHWC rows [B, ..., C]: whole channel maps of a sample are zeroed, the rest scaled by 1/(1-p)
    (conv2d_encoder.py:74,142-148).  p == 0 or eval -> identity (the parity configuration)."""
    if not training or p <= 0.0:
        return x
    if p >= 1.0:
        return torch.zeros_like(x)
    b, c = x.shape[0], x.shape[-1]
    return Fh.ChannelScaleFn.apply(x, dropout_mask((b, c), p, x))


class Conv2DEncoder(nn.Module):
    """[conv1x1(bias=False) -> GroupNorm -> ReLU -> Dropout2d] x (L-1) -> conv1x1 -> GroupNorm  (conv2d_encoder.py:100-127)."""

    def __init__(self, in_channels: int, channels: Sequence[int], kernel_size=1, padding=0,
                 dropout_rate: Union[float, Sequence[float]] = 0.0, num_groups: Union[int, Sequence[int]] = 8,
                 input_dropout_rate: float = 0.0):
        super().__init__()
        channels = list(channels)
        n = len(channels)
        assert n > 0
        ks = [kernel_size] * n if isinstance(kernel_size, int) else list(kernel_size)
        if any(k != 1 for k in ks):
            raise NotImplementedError("type encoder kernel_size must be 1 on the HIP path")
        dr = [dropout_rate] * n if isinstance(dropout_rate, (int, float)) else list(dropout_rate)
        ng = [num_groups] * n if isinstance(num_groups, int) else list(num_groups)
        self.in_channels, self.out_channels = in_channels, channels[-1]
        self.input_dropout = _Marker("dropout2d", input_dropout_rate)
        layers: List[nn.Module] = []
        prev = in_channels
        for i, (co, drop, g) in enumerate(zip(channels, dr, ng)):
            last = i == n - 1
            layers.append(Conv2dParams(prev, co, 1, bias=False))
            layers.append(GroupNormParams(g, co))
            if not last:
                layers.append(_Marker("relu"))
            if drop > 0 and not last:
                layers.append(_Marker("dropout2d", drop))
            prev = co
        self.layers = nn.ModuleList(layers)   # indices match the reference's nn.Sequential

    def set_input_dropout_rate(self, rate: float) -> None:
        self.input_dropout.p = rate

    fuse = True        # the two-layer bf16 configuration runs as one launch per direction (csrc/enc_fused.hip) when nothing prevents it
    fuse_min_samples = 96

    def _fused_layers(self, x: torch.Tensor):
        """(conv1, norm1, conv2, norm2) when the fused kernels apply to this call, else None."""
        mods = list(self.layers)
        convs = [m for m in mods if isinstance(m, Conv2dParams)]
        norms = [m for m in mods if isinstance(m, GroupNormParams)]
        if not self.fuse or len(convs) != 2 or len(norms) != 2 or x.dtype != torch.bfloat16 or not x.is_cuda or x.requires_grad:
            return None
        if self.training and (self.input_dropout.p > 0 or any(isinstance(m, _Marker) and m.kind == "dropout2d" and m.p > 0 for m in mods)):
            return None
        if any(c.bias is not None for c in convs) or norms[0].eps != norms[1].eps:
            return None
        if x.shape[0] < self.fuse_min_samples:                     # one workgroup per sample: a small batch would leave most CUs idle
            return None
        hw = x.numel() // (x.shape[0] * x.shape[-1])
        if not ops.encoder2_supported(convs[0].weight.shape[1], convs[0].weight.shape[0], convs[1].weight.shape[0], norms[0].num_groups,
                                         norms[1].num_groups, hw, x.dtype):
            return None
        return convs[0], norms[0], convs[1], norms[1]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        fused = self._fused_layers(x)
        if fused is not None:
            c1, n1, c2, n2 = fused
            return Fh.encoder2(x, c1.weight, n1.weight, n1.bias, c2.weight, n2.weight, n2.bias, n1.eps)
        hw = x.shape[1] * x.shape[2]
        x = channel_dropout(x, self.input_dropout.p, self.training, hw)
        i = 0
        mods = list(self.layers)
        while i < len(mods):
            conv, norm = mods[i], mods[i + 1]
            i += 2
            relu = i < len(mods) and isinstance(mods[i], _Marker) and mods[i].kind == "relu"
            x = norm(conv(x), relu=relu)
            if relu:
                i += 1
            if i < len(mods) and isinstance(mods[i], _Marker) and mods[i].kind == "dropout2d":
                x = channel_dropout(x, mods[i].p, self.training, hw)
                i += 1
        return x


class EdgeAwareSmoothingConv2D(nn.Module):
    """Directional filter bank + residual edge gate (spatial.py:278-339) on NHWC tensors."""

    def __init__(self, channels: int, num_layers: int = 2, kernel_size: int = 3, padding: int = 1, gate_hidden: int = 64,
                 gate_kernel_size: int = 3, num_directions: int = 4, coarse_dilation: int = 3, rank: int = 4):
        super().__init__()
        if num_directions != 4 or kernel_size != 3 or gate_kernel_size != 3:
            raise NotImplementedError("HIP stencil supports num_directions=4, 3x3 bank and 3x3 gate convolutions")
        self.channels, self.num_directions, self.coarse_dilation = channels, num_directions, coarse_dilation
        self.K, self.rank = num_directions * 2, rank
        t = [[[0., 0., 0.], [1 / 3, 1 / 3, 1 / 3], [0., 0., 0.]], [[0., 1 / 3, 0.], [0., 1 / 3, 0.], [0., 1 / 3, 0.]],
             [[1 / 3, 0., 0.], [0., 1 / 3, 0.], [0., 0., 1 / 3]], [[0., 0., 1 / 3], [0., 1 / 3, 0.], [1 / 3, 0., 0.]]]
        bank = torch.tensor(t).unsqueeze(1).unsqueeze(1).expand(4, channels, 1, 3, 3).contiguous()
        self.register_buffer("bank", bank)                                  # kept for state-dict parity; fixed in-kernel
        sx = torch.tensor([[-1., 0., 1.], [-2., 0., 2.], [-1., 0., 1.]]) / 4.0
        sy = torch.tensor([[-1., -2., -1.], [0., 0., 0.], [1., 2., 1.]]) / 4.0
        self.register_buffer("sobel_x", sx.reshape(1, 1, 3, 3).expand(channels, 1, 3, 3).contiguous())
        self.register_buffer("sobel_y", sy.reshape(1, 1, 3, 3).expand(channels, 1, 3, 3).contiguous())
        self.mix_backbone = nn.ModuleList([Conv2dParams(2 * channels, gate_hidden, 3), _Marker("relu")])
        self.mix_head_A = Conv2dParams(gate_hidden, self.K * rank, 1)
        self.mix_head_B = Conv2dParams(gate_hidden, channels * rank, 1)
        self.gate_net = nn.ModuleList([Conv2dParams(channels, gate_hidden, 3), _Marker("relu"),
                                       Conv2dParams(gate_hidden, channels, 3), _Marker("sigmoid")])
        self.min_gate: float = 0.0

    def set_min_gate(self, value: float) -> None:
        self.min_gate = float(value)

    fuse = True     # one autograd node for the block (Fh.SpatialSmoothFn: gradient sums in kernel epilogues); False = modular chain

    def forward(self, x: torch.Tensor, return_gate: bool = False):
        if self.fuse:
            mb, g0, g2 = self.mix_backbone[0], self.gate_net[0], self.gate_net[2]
            out, gate = Fh.SpatialSmoothFn.apply(x, mb.weight, mb.bias, self.mix_head_A.weight, self.mix_head_A.bias,
                                                 self.mix_head_B.weight, self.mix_head_B.bias, g0.weight, g0.bias, g2.weight, g2.bias,
                                                 self.rank, self.coarse_dilation, self.min_gate)
            return (out, gate) if return_gate else out
        g = Fh.SobelFn.apply(x)                                   # [B,H,W,2C] = cat[dx, dy]
        feat = self.mix_backbone[0](g, ACT_RELU)
        a_logit = self.mix_head_A(feat)
        b_logit = self.mix_head_B(feat)
        smoothed, residual = Fh.EdgeSmoothFn.apply(x, a_logit, b_logit, self.rank, self.coarse_dilation)
        g1 = self.gate_net[0](residual, ACT_RELU)
        gate_raw = self.gate_net[2](g1, ACT_SIGMOID)
        out, gate = Fh.GateBlendFn.apply(smoothed, residual, gate_raw, self.min_gate)
        return (out, gate) if return_gate else out


class GatedResidualBlock(nn.Module):
    """dropout -> conv1d(k=3, dil) -> GroupNorm -> gate(1x1) ; y = g * relu(n) + (1-g) * res  (tcn.py:78-111), fused."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int = 3, dilation: int = 1, dropout_rate: float = 0.0,
                 num_groups: int = 8, projection_channels: Optional[int] = None):
        super().__init__()
        if kernel_size != 3:
            raise NotImplementedError("fused TCN block implements kernel_size=3")
        if projection_channels not in (None, out_channels):
            raise NotImplementedError("projection_channels != out_channels is not on the hot path")
        self.in_channels, self.out_channels, self.dilation = in_channels, out_channels, dilation
        self.dropout = _Marker("dropout1d", dropout_rate)
        self.conv = Conv1dParams(in_channels, out_channels, 3)
        self.norm = GroupNormParams(num_groups, out_channels)
        self.gate = Conv1dParams(out_channels, out_channels, 1)
        self.needs_projection = in_channels != out_channels
        self.projection = Conv1dParams(in_channels, out_channels, 1) if self.needs_projection else _Marker("identity")

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x [B,T,HW..,Cin] -> [B,T,HW..,Cout]."""
        # Dropout1d on the conv input only (tcn.py:53,89-90): one keep/scale value per (pixel series, channel); the fused kernels
        # take it as a [B, HW.., C] mask so that the residual path still sees the untouched x
        mask = None
        if self.training and self.dropout.p > 0.0:
            mask = dropout_mask((x.shape[0],) + tuple(x.shape[2:]), min(self.dropout.p, 1.0 - 1e-6), x)
        pw = self.projection.weight if self.needs_projection else None
        pb = self.projection.bias if self.needs_projection else None
        if mask is not None and not ops.tcn_hot_supported(x, self.in_channels, self.out_channels, self.norm.num_groups, self.dilation,
                                                          self.needs_projection):
            return self._forward_masked_generic(x, mask, pw, pb)
        return Fh.TcnBlockFn.apply(x, self.conv.weight, self.conv.bias, self.norm.weight, self.norm.bias, self.gate.weight,
                                   self.gate.bias, pw, pb, self.dilation, self.norm.num_groups, self.norm.eps, mask)

    def _forward_masked_generic(self, x: torch.Tensor, mask: torch.Tensor, pw, pb) -> torch.Tensor:
        """Dropout1d for shapes outside the hot configuration (float32 parity mode, T != 5, other widths).  The generic kernels have
        one input, so the block is evaluated on the concatenation [x .* mask | x] with the conv weights padded by zeros over the second
        half and the residual projection padded by zeros over the first: conv sees the dropped-out series, the residual the untouched
        one (identity residual = an identity projection), exactly tcn.py:89-110.  Gradients flow through the same kernels."""
        shape = x.shape
        b, t, c = shape[0], shape[1], shape[-1]
        if 2 * c > 128:
            raise NotImplementedError("TCN Dropout1d outside the hot configuration needs 2 * C_in <= 128 channels")
        # the generic backward kernel keeps conv, gate and projection weights of the doubled input in LDS (csrc/tcn_bwd.hip)
        f32 = x.dtype == torch.float32
        pad = lambda n, lo: next(v for v in (16, 32, 64, 128) if v >= max(n, lo))          # noqa: E731
        pi, po = pad(2 * c, 16 if f32 else 32), pad(self.out_channels, 16 if f32 else 32)
        nfi, mbo = (pi // 4 if f32 else pi // 32), po // 16
        nfo = 4 * mbo if f32 else max(1, mbo // 2)
        lds = (3 * mbo * nfi + 2 * mbo * nfo + mbo * nfi) * 64 * (4 if f32 else 16) + 48 * 4 * mbo * 4
        if lds > 160 * 1024:
            raise NotImplementedError(f"TCN Dropout1d with {c} -> {self.out_channels} channels in {x.dtype} does not fit the generic backward "
                                      "kernel's LDS; use bf16 compute (or dropout 0.0) for this width")
        x4 = x.reshape(b, t, -1, c)
        m3 = mask.reshape(b, -1, c)
        xm = Fh.FilmFn.apply(x4, m3, torch.zeros_like(m3))                                 # x .* mask, broadcast over time
        x_aug = torch.cat([xm, x4], dim=-1)
        w = self.conv.weight
        w_aug = torch.cat([w, torch.zeros_like(w)], dim=1)
        if pw is None:
            pw = torch.eye(c, dtype=w.dtype, device=w.device).reshape(c, c, 1)
            pb = torch.zeros(c, dtype=w.dtype, device=w.device)
        p_aug = torch.cat([torch.zeros_like(pw), pw], dim=1)
        y = Fh.TcnBlockFn.apply(x_aug, w_aug, self.conv.bias, self.norm.weight, self.norm.bias, self.gate.weight, self.gate.bias,
                                p_aug, pb, self.dilation, self.norm.num_groups, self.norm.eps, None)
        return y.reshape(shape[:-1] + (self.out_channels,))


class TCNEncoder(nn.Module):
    """Stack of GatedResidualBlocks, pooling='none' (tcn.py:114-302) on [B,T,HW..,C] tensors."""

    def __init__(self, in_channels: int, channels: Sequence[int], kernel_size: int = 3, dilations: Optional[Sequence[int]] = None,
                 dropout_rate: float = 0.0, num_groups: int = 8, pooling: str = "none"):
        super().__init__()
        if pooling != "none":
            raise NotImplementedError("only pooling='none' is used by RepresentationModel (representation.py:165)")
        channels = list(channels)
        dilations = [1] * len(channels) if dilations is None else list(dilations)
        assert len(dilations) == len(channels)
        self.in_channels, self.out_channels, self.pooling = in_channels, channels[-1], pooling
        layers, prev = [], in_channels
        for co, d in zip(channels, dilations):
            layers.append(GatedResidualBlock(prev, co, kernel_size, d, dropout_rate, num_groups))
            prev = co
        self.layers = nn.ModuleList(layers)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        for layer in self.layers:
            x = layer(x)
        return x


class FiLMLayer(nn.Module):
    """gamma/beta = two independent conv1x1 -> ReLU -> conv1x1 nets (conditioning.py:55-80)."""

    def __init__(self, cond_dim: int, target_dim: int, hidden_dim: Optional[int] = None):
        super().__init__()
        hidden_dim = max(cond_dim, target_dim) // 2 if hidden_dim is None else hidden_dim
        self.cond_dim, self.target_dim = cond_dim, target_dim
        self.gamma_network = nn.ModuleList([Conv2dParams(cond_dim, hidden_dim, 1), _Marker("relu"), Conv2dParams(hidden_dim, target_dim, 1)])
        self.beta_network = nn.ModuleList([Conv2dParams(cond_dim, hidden_dim, 1), _Marker("relu"), Conv2dParams(hidden_dim, target_dim, 1)])
        with torch.no_grad():
            self.gamma_network[2].weight.normal_(0.0, 0.01)
            self.gamma_network[2].bias.fill_(1.0)
            self.beta_network[2].weight.normal_(0.0, 0.01)
            self.beta_network[2].bias.zero_()

    def forward(self, cond: torch.Tensor):
        g = self.gamma_network[2](self.gamma_network[0](cond, ACT_RELU))
        b = self.beta_network[2](self.beta_network[0](cond, ACT_RELU))
        return g, b


class Conv2DHead(nn.Module):
    """conv1x1 -> ReLU -> ... -> conv1x1 decoder (heads.py:128-198, kernel 1, activation 'none')."""

    def __init__(self, in_channels: int, channels: Sequence[int], out_channels: int):
        super().__init__()
        layers, prev = [], in_channels
        for ch in channels:
            layers += [Conv2dParams(prev, ch, 1), _Marker("relu")]
            prev = ch
        layers.append(Conv2dParams(prev, out_channels, 1))
        self.layers = nn.ModuleList(layers)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        mods = list(self.layers)
        for i in range(0, len(mods) - 1, 2):
            x = mods[i](x, ACT_RELU)
        return mods[-1](x)
