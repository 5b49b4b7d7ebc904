"""RepresentationModel with the reference's API (frl/models/representation.py:62-495) on HIP kernels.

Same constructor arguments, `from_config`, `forward`, `forward_phase`, `forward_phase_at_locations`, setters,
`from_checkpoint`, `VERSION`, sub-module names and state-dict keys as the reference, so reference checkpoints load with
`load_state_dict` and `train_representation.py`-style callers need no change.  Public methods accept the reference's
channels-first tensors; `*_nhwc` methods are the zero-copy entry points used by the VQ-VAE trainer (tiles are already
(time, y, x, feature)).
"""
from __future__ import annotations

import inspect
import logging
from pathlib import Path
from typing import List, Optional

import torch
import torch.nn as nn

from .. import _lib
from .. import functional as Fh
from .. import ops
from .blocks import Conv2DEncoder, Conv2dParams, EdgeAwareSmoothingConv2D, FiLMLayer, TCNEncoder

logger = logging.getLogger(__name__)


class RepresentationModel(nn.Module):
    VERSION = "4"   # representation.py:86

    def __init__(
        self,
        type_in_channels: int,
        phase_in_channels: int,
        z_type_dim: int = 64,
        z_phase_dim: int = 12,
        type_encoder_channels: List[int] = (128, 64),
        type_encoder_kernel_size: int = 1,
        type_encoder_padding: int = 0,
        type_encoder_dropout: float = 0.1,
        type_encoder_num_groups: int = 8,
        type_encoder_input_dropout: float = 0.0,
        spatial_conv_num_layers: int = 2,
        spatial_conv_kernel_size: int = 3,
        spatial_conv_padding: int = 1,
        spatial_conv_gate_hidden: int = 64,
        spatial_conv_gate_kernel_size: int = 3,
        spatial_conv_num_directions: int = 4,
        spatial_conv_coarse_dilation: int = 3,
        spatial_conv_rank: int = 4,
        phase_tcn_channels: List[int] = (64, 64, 64),
        phase_tcn_kernel_size: int = 3,
        phase_tcn_dilations: List[int] = (1, 2, 4),
        phase_tcn_dropout: float = 0.1,
        phase_tcn_num_groups: int = 8,
        type_proj_hidden_dim: Optional[int] = None,
        type_proj_output_dim: Optional[int] = None,
        type_proj_l2_normalize: bool = True,
        compute_dtype: torch.dtype = torch.float32,
    ) -> None:
        super().__init__()
        if list(type_encoder_channels)[-1] != z_type_dim:
            raise ValueError(f"type_encoder_channels[-1]={type_encoder_channels[-1]} must equal z_type_dim={z_type_dim}")
        if compute_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("compute_dtype must be torch.float32 (parity mode) or torch.bfloat16 (performance mode)")
        self.type_in_channels = type_in_channels
        self.phase_in_channels = phase_in_channels
        self.z_type_dim = z_type_dim
        self.z_phase_dim = z_phase_dim
        self.compute_dtype = compute_dtype
        self.encoder = Conv2DEncoder(type_in_channels, list(type_encoder_channels), type_encoder_kernel_size,
                                     type_encoder_padding, type_encoder_dropout, type_encoder_num_groups,
                                     type_encoder_input_dropout)
        self.spatial_conv = EdgeAwareSmoothingConv2D(z_type_dim, spatial_conv_num_layers, spatial_conv_kernel_size,
                                                     spatial_conv_padding, spatial_conv_gate_hidden,
                                                     spatial_conv_gate_kernel_size, spatial_conv_num_directions,
                                                     spatial_conv_coarse_dilation, spatial_conv_rank)
        self.phase_tcn = TCNEncoder(phase_in_channels, list(phase_tcn_channels), phase_tcn_kernel_size,
                                    list(phase_tcn_dilations), phase_tcn_dropout, phase_tcn_num_groups, pooling="none")
        self.phase_head = Conv2dParams(list(phase_tcn_channels)[-1], z_phase_dim, 1)
        self.phase_film = FiLMLayer(cond_dim=z_type_dim, target_dim=z_phase_dim)
        if type_proj_hidden_dim is not None and type_proj_output_dim is not None:
            raise NotImplementedError("type_projection (SimCLR head) is disabled in the live config "
                                      "(frl_repr_model_v1.yaml:64-69) and not on the VQ-VAE path")
        self.type_projection = None

    # ------------------------------------------------------------------ construction helpers
    @classmethod
    def from_config(cls, cfg: dict, type_in_channels: int, phase_in_channels: int, **extra) -> "RepresentationModel":
        """Same contract as representation.py:194-279 (version check, scalar-or-schedule input_dropout)."""
        cfg_version = str(cfg.get("version", ""))
        if cfg_version != cls.VERSION:
            raise ValueError(f"Config version={cfg_version!r} does not match RepresentationModel.VERSION={cls.VERSION!r}.")
        latents = cfg.get("latents", {})
        te, sc, pt, tp = (cfg.get(k, {}) for k in ("type_encoder", "spatial_conv", "phase_tcn", "type_projection"))
        idc = te.get("input_dropout", 0.0)
        input_dropout = float(idc.get("start", 0.0)) if isinstance(idc, dict) else float(idc)
        return cls(
            type_in_channels=type_in_channels, phase_in_channels=phase_in_channels,
            z_type_dim=latents.get("z_type_dim", 64), z_phase_dim=latents.get("z_phase_dim", 12),
            type_encoder_channels=te.get("channels", [128, 64]), type_encoder_kernel_size=te.get("kernel_size", 1),
            type_encoder_padding=te.get("padding", 0), type_encoder_dropout=te.get("dropout", 0.1),
            type_encoder_num_groups=te.get("num_groups", 8), type_encoder_input_dropout=input_dropout,
            spatial_conv_num_layers=sc.get("num_layers", 2), spatial_conv_kernel_size=sc.get("kernel_size", 3),
            spatial_conv_padding=sc.get("padding", 1), spatial_conv_gate_hidden=sc.get("gate_hidden", 64),
            spatial_conv_gate_kernel_size=sc.get("gate_kernel_size", 3), spatial_conv_num_directions=sc.get("num_directions", 4),
            spatial_conv_coarse_dilation=sc.get("coarse_dilation", 3), spatial_conv_rank=sc.get("rank", 4),
            phase_tcn_channels=pt.get("channels", [64, 64, 64]), phase_tcn_kernel_size=pt.get("kernel_size", 3),
            phase_tcn_dilations=pt.get("dilations", [1, 2, 4]), phase_tcn_dropout=pt.get("dropout", 0.1),
            phase_tcn_num_groups=pt.get("num_groups", 8),
            type_proj_hidden_dim=tp.get("hidden_dim") if tp.get("enabled", False) else None,
            type_proj_output_dim=tp.get("output_dim") if tp.get("enabled", False) else None,
            type_proj_l2_normalize=tp.get("l2_normalize", True), **extra)

    def set_spatial_min_gate(self, value: float) -> None:
        self.spatial_conv.set_min_gate(value)

    def set_input_dropout_rate(self, rate: float) -> None:
        self.encoder.set_input_dropout_rate(rate)

    def project_type(self, z: torch.Tensor) -> torch.Tensor:
        return z if self.type_projection is None else self.type_projection(z)

    # ------------------------------------------------------------------ helpers
    def _require_gpu(self, t: torch.Tensor):
        if not t.is_cuda:
            raise _lib.FrlHipError("frl_hip models run on the GPU only (HIP kernels; there is no CPU fallback)")

    def _rows(self, t: torch.Tensor) -> torch.Tensor:
        t = t.to(self.compute_dtype)
        return t if t.is_contiguous() else t.contiguous()

    # ------------------------------------------------------------------ NHWC entry points (no layout copies)
    def forward_nhwc(self, x: torch.Tensor, return_gate: bool = False):
        """x [B,H,W,C_type] -> z_type [B,H,W,d] (and gate)."""
        self._require_gpu(x)
        h = self.encoder(self._rows(x))
        return self.spatial_conv(h, return_gate=return_gate)

    def _phase_chain(self, x: torch.Tensor) -> torch.Tensor:
        """phase_tcn -> phase_head on [B,T,HW..,C] rows.  Hot configuration (bf16, three 64-channel blocks with dilation 1, 2, 4, T = 5,
        no active Dropout1d, a head of <= 16 channels): one forward launch for the whole chain (Fh.TcnChainHeadFn)."""
        tcn, head = self.phase_tcn, self.phase_head
        layers = list(tcn.layers)
        drop = any(self.training and l.dropout.p > 0.0 for l in layers)
        if len(layers) == 3 and not drop and head.k == 1 and head.bias is not None:
            blocks = [(l.conv.weight, l.conv.bias, l.norm.weight, l.norm.bias, l.gate.weight, l.gate.bias, l.dilation, l.norm.num_groups,
                       l.needs_projection) for l in layers]
            if ops.tcn_chain_supported(x, blocks, head.weight):
                flat = [t for blk in blocks for t in blk[:6]]
                return Fh.TcnChainHeadFn.apply(x, *flat, head.weight, head.bias, layers[0].norm.num_groups, layers[0].norm.eps)
        return head(tcn(x))

    def forward_phase_nhwc(self, x_phase: torch.Tensor, z_type: torch.Tensor, return_parts: bool = False):
        """x_phase [B,T,H,W,C_phase], z_type [B,H,W,d] (caller stop-grads) -> z_phase [B,T,H,W,zp]."""
        self._require_gpu(x_phase)
        h = self._phase_chain(self._rows(x_phase))
        zt = self._rows(z_type)
        film = self.phase_film
        gn, bn = film.gamma_network, film.beta_network
        if not zt.requires_grad and ops.film_fused_supported(zt, h, gn[0].out_channels) and all(m.bias is not None for m in (gn[0], gn[2], bn[0], bn[2])):
            # hot configuration: both FiLM nets and the modulation as one launch per direction (csrc/film_fused.hip); the conditioning
            # input is a stop-gradient (representation.py:350-351), which is what the fused backward assumes
            z, gamma, beta = Fh.FilmFusedFn.apply(h, zt, gn[0].weight, gn[0].bias, gn[2].weight, gn[2].bias, bn[0].weight, bn[0].bias,
                                                  bn[2].weight, bn[2].bias)
        else:
            gamma, beta = film(zt)
            z = Fh.FilmFn.apply(h, gamma, beta)
        return (z, gamma, beta, h) if return_parts else z

    # ------------------------------------------------------------------ reference (channels-first) API
    def forward(self, x: torch.Tensor, return_gate: bool = False):
        """Type pathway, representation.py:317-334: x [B,C,H,W] -> z_type [B,d,H,W] (, gate)."""
        xr = x.permute(0, 2, 3, 1)
        if return_gate:
            z, gate = self.forward_nhwc(xr, True)
            return z.permute(0, 3, 1, 2), gate.permute(0, 3, 1, 2)
        return self.forward_nhwc(xr).permute(0, 3, 1, 2)

    def forward_phase(self, x_phase: torch.Tensor, z_type: torch.Tensor) -> torch.Tensor:
        """Dense phase pathway, representation.py:336-374: [B,C,T,H,W], [B,d,H,W] -> [B,zp,T,H,W]."""
        z = self.forward_phase_nhwc(x_phase.permute(0, 2, 3, 4, 1), z_type.permute(0, 2, 3, 1))
        return z.permute(0, 4, 1, 2, 3)

    def forward_phase_at_locations(self, x_phase_pixels: torch.Tensor, z_type_pixels: torch.Tensor,
                                   return_film: bool = False, return_pre_film: bool = False):
        """Sparse phase pathway, representation.py:376-436: [N,C,T], [N,d] -> [N,T,zp] (+ gamma, beta [N,zp]; h [N,zp,T])."""
        xr = x_phase_pixels.permute(2, 0, 1).unsqueeze(0)          # [1,T,N,C]
        zr = z_type_pixels.unsqueeze(0)                            # [1,N,d]
        z, gamma, beta, h = self.forward_phase_nhwc(xr, zr, return_parts=True)
        z = z[0].permute(1, 0, 2)                                  # [N,T,zp]
        if return_film and return_pre_film:
            return z, gamma[0], beta[0], h[0].permute(1, 2, 0)
        if return_film:
            return z, gamma[0], beta[0]
        if return_pre_film:
            return z, h[0].permute(1, 2, 0)
        return z

    # ------------------------------------------------------------------ checkpoints
    @classmethod
    def from_checkpoint(cls, path, device="cuda", freeze: bool = True, **extra) -> "RepresentationModel":
        """representation.py:442-490.  Loaded with weights_only=True (plain dict of tensors / config values)."""
        checkpoint = torch.load(path, map_location=device, weights_only=True)
        ckpt_version = checkpoint.get("model_version")
        if ckpt_version != cls.VERSION:
            raise RuntimeError(f"Checkpoint model_version={ckpt_version!r} is not supported. "
                               f"RepresentationModel.VERSION={cls.VERSION!r}.")
        model = cls.from_config(checkpoint["model_config"], type_in_channels=checkpoint["type_in_channels"],
                                phase_in_channels=checkpoint["phase_in_channels"], **extra).to(device)
        model.load_state_dict(checkpoint["model_state_dict"], strict=False)
        if freeze:
            for p in model.parameters():
                p.requires_grad = False
            model.eval()
        logger.info(f"Loaded RepresentationModel v{cls.VERSION} from {path}")
        return model

    @staticmethod
    def source_file() -> Path:
        return Path(inspect.getfile(RepresentationModel))
