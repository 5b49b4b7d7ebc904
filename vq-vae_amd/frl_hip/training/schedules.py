"""Pure-function schedules of the trainer (no torch, no model state).

Mirrors frl/training/representation/curriculum.py:16-83 (input-dropout schedule, 0->1 curriculum ramp, smoothing
gate lock), the cosine LR of scripts/train_vqvae.py:250-253 / configs/vae_v0.yaml:13-19, the warm-up + cosine
LambdaLR of frl/training/representation/scheduler.py:142-153, and the linear beta ramp of configs/vae_v0.yaml:21-27.
"""
from __future__ import annotations

import math
from typing import Union


def compute_input_dropout_rate(schedule_cfg: Union[float, dict], epoch: int, total_epochs: int) -> float:
    """Scalar -> constant rate; dict(schedule=constant|linear|cosine, start, end, epochs) -> ramped rate."""
    if isinstance(schedule_cfg, (int, float)):
        return float(schedule_cfg)
    kind = schedule_cfg.get("schedule", "constant")
    if kind == "constant":
        return float(schedule_cfg.get("rate", 0.0))
    start = float(schedule_cfg.get("start", 0.0))
    end = float(schedule_cfg.get("end", 0.1))
    ramp = int(schedule_cfg.get("epochs", total_epochs))
    t = min(epoch / max(ramp, 1), 1.0)
    if kind == "linear":
        return start + t * (end - start)
    if kind == "cosine":
        return start + (end - start) * (1 - math.cos(math.pi * t)) / 2
    raise ValueError(f"Unknown input_dropout schedule: {kind!r}")


def ramp_weight(epoch: int, start_epoch: int, ramp_epochs: int) -> float:
    """0 before start_epoch, 1 at/after start_epoch + ramp_epochs, linear in between (exactly 0 at start_epoch)."""
    if epoch < start_epoch:
        return 0.0
    if epoch >= start_epoch + ramp_epochs:
        return 1.0
    return (epoch - start_epoch) / ramp_epochs


def compute_smoothing_min_gate(epoch: int, freeze_until_epoch: int, ramp_epochs: int) -> float:
    """Gate floor: 1.0 (identity) while frozen, then linearly released to 0."""
    return 1.0 - ramp_weight(epoch, freeze_until_epoch, ramp_epochs)


def cosine_lr(step: int, total_steps: int, lr: float, min_lr: float) -> float:
    prog = min(step / max(total_steps, 1), 1.0)
    return min_lr + (lr - min_lr) * 0.5 * (1.0 + math.cos(math.pi * prog))


def warmup_cosine_factor(step: int, warmup_steps: int, total_steps: int, eta_min_factor: float) -> float:
    """LambdaLR multiplier: max(step/warmup, 1e-8) during warm-up, then cosine 1 -> eta_min/lr."""
    if warmup_steps > 0 and step < warmup_steps:
        return max(step / warmup_steps, 1e-8)
    prog = (step - warmup_steps) / max(total_steps - warmup_steps, 1)
    return eta_min_factor + (1.0 - eta_min_factor) * 0.5 * (1.0 + math.cos(math.pi * prog))


def beta_schedule(epoch: int, cfg: dict) -> float:
    """configs/vae_v0.yaml beta_schedule block: linear start_value -> end_value over [start_epoch, end_epoch]."""
    if not cfg or not cfg.get("enabled", False):
        return float((cfg or {}).get("end_value", 1.0))
    s, e = cfg["start_epoch"], cfg["end_epoch"]
    if epoch <= s:
        return float(cfg["start_value"])
    if epoch >= e:
        return float(cfg["end_value"])
    return float(cfg["start_value"] + (cfg["end_value"] - cfg["start_value"]) * (epoch - s) / (e - s))
