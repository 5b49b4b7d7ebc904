"""Per-feature normalisation presets -> the record table consumed by `frl_normalize_tiles` (include/frl_hip.h).

The reference normalises every channel on the host with numpy (FeatureBuilder._normalize_array,
frl/data/loaders/builders/feature_builder.py:487-548; presets of frl/data/normalization/normalization.py:116-254; preset
schema `normalization.presets` of frl/config/frl_binding_v1.yaml).  Here a preset plus the channel's statistics collapse into
one affine-and-clamp record per feature, applied on the device in the same float32 operation order:

    zscore          r = (x - mean) / sd            sd  < 1e-8 -> 1      (defaults mean 0, sd 1)
    robust_iqr      r = (x - q50) / (q75 - q25)    iqr < 1e-8 -> 1      (defaults q25 0, q50 0, q75 1)
    minmax          r = (x - min) / (max - min)    range <= 1e-8 -> 1
    linear_rescale  r = ((x - in_min) / in_range) * out_range + out_min,  in_range < 1e-8 -> 1
    clamp | none | identity     r = x
    then  np.clip(r, clamp.min, clamp.max)  when  clamp.enabled
"""
from __future__ import annotations

import struct
from dataclasses import dataclass
from typing import Dict, Mapping, Optional, Sequence

import numpy as np

FLAG_RESCALE, FLAG_LO, FLAG_HI = 1, 2, 4
PRESET_TYPES = ("zscore", "robust_iqr", "minmax", "linear_rescale", "clamp", "none", "identity")


@dataclass
class NormPreset:
    type: str = "identity"
    clamp: Optional[Mapping] = None            # {"enabled": bool, "min": float | None, "max": float | None}
    min: Optional[float] = None                # minmax with fixed bounds
    max: Optional[float] = None
    in_min: Optional[float] = None             # linear_rescale
    in_max: Optional[float] = None
    out_min: Optional[float] = None
    out_max: Optional[float] = None

    @classmethod
    def from_dict(cls, d: Optional[Mapping]) -> "NormPreset":
        if not d:
            return cls()
        unknown = set(d) - {"type", "clamp", "min", "max", "in_min", "in_max", "out_min", "out_max", "stats_source", "fields", "missing"}
        if unknown:
            raise ValueError(f"unknown normalisation preset keys {sorted(unknown)}")
        p = cls(**{k: d[k] for k in ("type", "clamp", "min", "max", "in_min", "in_max", "out_min", "out_max") if k in d})
        if p.type not in PRESET_TYPES:
            raise ValueError(f"unknown normalisation type '{p.type}' (one of {PRESET_TYPES})")
        return p


def norm_record(preset: NormPreset, stats: Optional[Mapping[str, float]] = None):
    """-> (sub, div, mul, add, lo, hi, flags) of one feature (python floats; rounded to float32 when packed)."""
    stats = stats or {}
    sub, div, mul, add, lo, hi, flags = 0.0, 1.0, 1.0, 0.0, 0.0, 0.0, 0
    if preset.type == "zscore":
        sub, div = float(stats.get("mean", 0.0)), float(stats.get("sd", 1.0))
        if div < 1e-8:
            div = 1.0
    elif preset.type == "robust_iqr":
        sub = float(stats.get("q50", 0.0))
        div = float(stats.get("q75", 1.0)) - float(stats.get("q25", 0.0))
        if div < 1e-8:
            div = 1.0
    elif preset.type == "minmax":
        lo_v = preset.min if preset.min is not None and preset.max is not None else stats.get("min")
        hi_v = preset.max if preset.min is not None and preset.max is not None else stats.get("max")
        if lo_v is None or hi_v is None:
            raise ValueError("minmax normalisation requires 'min' and 'max'")
        sub, div = float(lo_v), float(hi_v) - float(lo_v)
        if not div > 1e-8:
            div = 1.0
    elif preset.type == "linear_rescale":
        in_min = 0.0 if preset.in_min is None else float(preset.in_min)
        in_max = 1.0 if preset.in_max is None else float(preset.in_max)
        out_min = 0.0 if preset.out_min is None else float(preset.out_min)
        out_max = 1.0 if preset.out_max is None else float(preset.out_max)
        sub, div = in_min, in_max - in_min
        if div < 1e-8:
            div = 1.0
        mul, add, flags = out_max - out_min, out_min, flags | FLAG_RESCALE
    elif preset.type not in ("clamp", "none", "identity"):
        raise ValueError(f"unknown normalisation type '{preset.type}'")
    if preset.clamp and preset.clamp.get("enabled", False):
        if preset.clamp.get("min") is not None:
            lo, flags = float(preset.clamp["min"]), flags | FLAG_LO
        if preset.clamp.get("max") is not None:
            hi, flags = float(preset.clamp["max"]), flags | FLAG_HI
    if (flags & FLAG_LO) and (flags & FLAG_HI) and lo > hi:
        raise ValueError(f"clamp.min {lo} > clamp.max {hi}")
    if not all(np.isfinite(v) for v in (sub, div, mul, add, lo, hi)) or div == 0.0:
        raise ValueError("normalisation constants must be finite and the divisor non-zero")
    return sub, div, mul, add, lo, hi, flags


def norm_table(presets: Sequence[NormPreset], stats: Sequence[Optional[Mapping[str, float]]]) -> np.ndarray:
    """Byte image of `FrlNormRec table[F]` (6 float32 + flags + pad = 32 bytes per feature)."""
    if len(presets) != len(stats):
        raise ValueError("one statistics entry per feature")
    buf = bytearray()
    for p, s in zip(presets, stats):
        sub, div, mul, add, lo, hi, flags = norm_record(p, s)
        buf += struct.pack("<6f2i", *(np.float32(v) for v in (sub, div, mul, add, lo, hi)), flags, 0)
    return np.frombuffer(bytes(buf), dtype=np.uint8).copy()


def identity_table(features: int) -> np.ndarray:
    return norm_table([NormPreset()] * features, [None] * features)


def presets_from_meta(meta: Dict) -> tuple:
    """(presets, stats) lists in feature order from a tile-store `meta.json`."""
    names = meta["features"]
    norm = meta.get("normalization", {})
    st = meta.get("stats", {})
    return [NormPreset.from_dict(norm.get(n)) for n in names], [st.get(n) for n in names]
