"""Shared builders for the tile-ingest tests (CPU and GPU): a preset mix covering every normalisation type and a raw cube."""
import numpy as np

PRESET_CYCLE = [
    ({"type": "zscore"}, {"mean": 0.37, "sd": 2.5}),
    ({"type": "zscore", "clamp": {"enabled": True, "min": -1.5, "max": 2.0}}, {"mean": -3.0, "sd": 1e-9}),     # sd floor -> 1
    ({"type": "robust_iqr"}, {"q25": -0.7, "q50": 0.1, "q75": 0.9}),
    ({"type": "robust_iqr", "clamp": {"enabled": True, "min": -6.0, "max": 6.0}}, {"q25": 2.0, "q50": 2.0, "q75": 2.0}),
    ({"type": "linear_rescale", "in_min": -0.4, "in_max": 0.4, "out_min": -1.0, "out_max": 1.0}, None),
    ({"type": "minmax", "min": -2.0, "max": 5.0}, None),
    ({"type": "minmax"}, {"min": 0.25, "max": 7.75}),
    ({"type": "clamp", "clamp": {"enabled": True, "min": -0.5, "max": None}}, None),
    ({"type": "identity"}, None),
    ({"type": "none", "clamp": {"enabled": False, "min": -0.1, "max": 0.1}}, None),
    ({"type": "linear_rescale", "in_min": 3.0, "in_max": 3.0, "out_min": 0.0, "out_max": 10.0,
      "clamp": {"enabled": True, "min": None, "max": 4.0}}, None),
]


def preset_mix(features):
    presets = [PRESET_CYCLE[i % len(PRESET_CYCLE)][0] for i in range(features)]
    stats = [PRESET_CYCLE[i % len(PRESET_CYCLE)][1] for i in range(features)]
    return presets, stats


def raw_rows(shape, dtype, seed, nan_frac=0.02, inf_rows=True):
    """Random rows [..., F] with scattered NaNs (missing observations), a few +-inf and large magnitudes."""
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal(shape) * rng.choice([0.1, 1.0, 30.0], size=shape)).astype(np.float32)
    x[rng.random(shape) < nan_frac / max(1, shape[-1]) * 4] = np.nan
    flat = x.reshape(-1, shape[-1])
    if inf_rows and flat.shape[0] > 8:
        flat[3, 1] = np.inf
        flat[7, shape[-1] - 1] = -np.inf
    return x.astype(dtype)


def apply_records_np(raw, valid, table_bytes):
    """float32 numpy evaluation of the record form, operation for operation what the device kernel does."""
    rec = np.frombuffer(table_bytes.tobytes(), dtype=np.dtype([("sub", "<f4"), ("div", "<f4"), ("mul", "<f4"), ("add", "<f4"),
                                                               ("lo", "<f4"), ("hi", "<f4"), ("flags", "<i4"), ("pad", "<i4")]))
    x = raw.astype(np.float32)
    ok = np.isfinite(x).all(-1)
    if valid is not None:
        ok &= valid.astype(bool)
    with np.errstate(invalid="ignore", over="ignore"):
        r = (x - rec["sub"]) / rec["div"]
        r = np.where(rec["flags"] & 1, r * rec["mul"] + rec["add"], r).astype(np.float32)
        r = np.where((rec["flags"] & 2) != 0, np.maximum(r, rec["lo"]), r)
        r = np.where((rec["flags"] & 4) != 0, np.minimum(r, rec["hi"]), r)
    return np.where(ok[..., None], r, np.float32(0)).astype(np.float32), ok.astype(np.uint8)
