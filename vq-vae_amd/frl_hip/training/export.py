"""Codebook export: every code decoded back to the data scale.

Counterpart of scripts/export_codebook.py:76-183, whose helpers (`decode_codebook_sequences`, `denorm_continuous_KTC`,
`code_summary`) live in the `vqvae` package that the reference tree does not contain -- build definition, same bundle layout:
one `.npz` with `cont_KT` [K*T, C] (decoded features in ORIGINAL units), `code_id` [K*T], `year` [K*T], `codes_K3` [K, 3] =
(code id, usage, canopy) and a JSON `meta`; optional CSVs.  The type codebook is time-less, so T = 1 unless `years` are given (the
decoded vector is then repeated per year, which keeps the reference's (code, year) row indexing).

Decoding runs the type decoder's HIP kernels on the K codebook rows; de-normalisation inverts the per-feature presets of the tile
store on the host (float64): zscore x*sd+mean, robust_iqr x*iqr+q50, minmax x*(max-min)+min, linear_rescale its inverse map;
clamp / none / identity are left as they are (a clamp cannot be undone).
"""
from __future__ import annotations

import json
from pathlib import Path
from typing import Dict, Mapping, Optional, Sequence

import numpy as np
import torch

from ..data.normalization import NormPreset, norm_record, FLAG_RESCALE


def denormalize(values: np.ndarray, presets: Sequence[NormPreset], stats: Sequence[Optional[Mapping[str, float]]]) -> np.ndarray:
    """Inverse of the tile-ingest normalisation, feature by feature (last axis); clamps are not inverted."""
    out = np.asarray(values, dtype=np.float64).copy()
    if out.shape[-1] != len(presets):
        raise ValueError("one preset per feature")
    for c, (p, s) in enumerate(zip(presets, stats)):
        sub, div, mul, add, _, _, flags = norm_record(p, s)
        x = out[..., c]
        if flags & FLAG_RESCALE:
            x = (x - add) / mul if mul != 0.0 else np.full_like(x, np.nan)
        out[..., c] = x * div + sub
    return out


@torch.no_grad()
def decode_codebook(model) -> torch.Tensor:
    """[K, F] float32: the type decoder applied to every codebook vector (on the model's device, through the HIP kernels)."""
    cb = model.quant.codebook.detach()
    k, d = cb.shape
    dtype = getattr(model, "compute_dtype", torch.float32)
    was_training = model.training
    model.eval()
    try:
        x = model.decoder_type(cb.to(dtype).reshape(1, k, 1, d).contiguous())
    finally:
        model.train(was_training)
    return x.reshape(k, -1).float()


def export_codebook(model, out_prefix, feature_names: Sequence[str], presets: Optional[Sequence[NormPreset]] = None,
                    stats: Optional[Sequence[Optional[Mapping[str, float]]]] = None, usage: Optional[torch.Tensor] = None,
                    years: Optional[Sequence[int]] = None, csv: bool = False) -> Path:
    decoded = decode_codebook(model).cpu().numpy()
    k, c = decoded.shape
    if len(feature_names) != c:
        raise ValueError(f"{c} decoded features but {len(feature_names)} names")
    if presets is not None:
        decoded = denormalize(decoded, presets, stats if stats is not None else [None] * c)
    yrs = np.asarray([0] if years is None else list(years), dtype=np.int32)
    t = int(yrs.shape[0])
    cont_kt = np.repeat(decoded.astype(np.float32)[:, None, :], t, axis=1).reshape(k * t, c)
    code_id = np.repeat(np.arange(k, dtype=np.int32), t)
    year = np.tile(yrs, k)
    if usage is None:
        mgr = getattr(model, "codebook_manager", None)
        usage = mgr.usage() if mgr is not None and getattr(mgr, "window", None) is not None else None
    use = np.full(k, np.nan) if usage is None else np.asarray(usage.detach().cpu() if torch.is_tensor(usage) else usage, dtype=np.float64)
    codes_k3 = np.stack([np.arange(k, dtype=np.float64), use, np.full(k, np.nan)], axis=1)
    out_prefix = Path(out_prefix)
    out_prefix.parent.mkdir(parents=True, exist_ok=True)
    meta: Dict = {"cont_names": list(feature_names), "cat_names": [], "T": t, "K": k,
                  "shapes": {"cont_KT": list(cont_kt.shape), "cats_KT": [k * t, 0], "code_id": list(code_id.shape), "year": list(year.shape),
                             "codes_K3": list(codes_k3.shape)},
                  "notes": "cont_KT holds decoded codebook vectors in original units; canopy in codes_K3 is NaN (no canopy head)."}
    npz = out_prefix.with_suffix(".npz")
    np.savez_compressed(npz, cont_KT=cont_kt, cats_KT=np.zeros((k * t, 0), dtype=np.float32), code_id=code_id, year=year,
                        codes_K3=codes_k3, meta=json.dumps(meta))
    if csv:
        import pandas as pd
        df = pd.DataFrame(cont_kt, columns=list(feature_names))
        df.insert(0, "year", year.astype(int))
        df.insert(0, "code_id", code_id.astype(int))
        df.to_csv(out_prefix.with_name(out_prefix.name + "_cont_KT.csv"), index=False)
        pd.DataFrame(codes_k3, columns=["code_id", "code_usage", "canopy"]).astype({"code_id": int}).to_csv(
            out_prefix.with_name(out_prefix.name + "_codes_K3.csv"), index=False)
    return npz
