"""Sparse-location gathers of the live trainer (frl/utils/spatial.py:132-173) on the HIP path.

`extract_at_locations(feature [C, H, W], coords [N, 2]) -> [N, C]` and `extract_temporal_at_locations(feature [C, T, H, W], coords)
-> [N, T, C]` keep the reference's signatures (callers: frl/training/representation/step.py:526,562,602).  The kernel takes element
strides, so the reference's channels-first view of one of this library's NHWC tensors (`z[b].permute(2, 0, 1)`) is gathered in place,
row by row -- which is the coalesced direction for NHWC.  Backward: rows of the incoming gradient that address the same pixel are
summed in list order over a key-sorted list (`frl_segment_sum_rows`): bit-reproducible, no float atomics.
"""
from __future__ import annotations

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import ops


class _GatherFn(Function):
    @staticmethod
    def forward(ctx, feature, coords):
        c, h, w = feature.shape
        ctx.shape = (c, h, w)
        ctx.save_for_backward(coords)
        ctx.strides, ctx.dtype = feature.stride(), feature.dtype
        return ops.gather_locations(feature, coords)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (coords,) = ctx.saved_tensors
        c, h, w = ctx.shape
        r = torch.where(coords[:, 0] < 0, coords[:, 0] + h, coords[:, 0])
        q = torch.where(coords[:, 1] < 0, coords[:, 1] + w, coords[:, 1])
        keys, order = torch.sort(r * w + q, stable=True)
        rows = torch.zeros(h * w, c, dtype=torch.float32, device=g.device)              # NHWC rows of the gradient raster
        ops.segment_sum_rows(g.float().contiguous(), order, keys, rows)
        return rows.reshape(h, w, c).permute(2, 0, 1).to(ctx.dtype), None              # a [C, H, W] view, like the input


def extract_at_locations(feature: torch.Tensor, coords: torch.Tensor) -> torch.Tensor:
    """feature [C, H, W] (any strides), coords [N, 2] as (row, col) -> [N, C]."""
    if feature.dim() != 3 or coords.dim() != 2 or coords.shape[1] != 2:
        raise ValueError("extract_at_locations: feature must be [C, H, W] and coords [N, 2]")
    coords = coords.to(device=feature.device, dtype=torch.int64)
    _, h, w = feature.shape
    coords = torch.stack([ops.sanitize_indices(coords[:, 0], h, "extract_at_locations: row"),
                          ops.sanitize_indices(coords[:, 1], w, "extract_at_locations: col")], dim=1).contiguous()
    if feature.requires_grad:
        return _GatherFn.apply(feature, coords)
    return ops.gather_locations(feature, coords)


def extract_temporal_at_locations(feature: torch.Tensor, coords: torch.Tensor) -> torch.Tensor:
    """feature [C, T, H, W], coords [N, 2] -> [N, T, C] (one gather per time step; T <= 15 in this model)."""
    if feature.dim() != 4:
        raise ValueError("extract_temporal_at_locations: feature must be [C, T, H, W]")
    return torch.stack([extract_at_locations(feature[:, t], coords) for t in range(feature.shape[1])], dim=1)
