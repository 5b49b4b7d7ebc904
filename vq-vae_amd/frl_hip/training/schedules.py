"""Schedules of the trainer as pure functions of the epoch / step (no torch, no model state).

Every schedule here is "a value that travels from `lo` to `hi` along a unit progress u in [0, 1] with some easing"; the
module is organised around that one primitive (`_progress`, `_ease`) and the public functions are thin named views of it:

  * epoch curricula of the live trainer -- input-dropout rate, 0 -> 1 loss ramp, smoothing-gate lock -- with the names, arguments
    and values of frl/training/representation/curriculum.py:16-83 (pinned against reference-generated samples in
    tests/golden/schedules.json);
  * learning rate: cosine lr -> min_lr of the legacy trainer (scripts/train_vqvae.py:250-253, configs/vae_v0.yaml:13-19) and the
    warm-up + cosine LambdaLR factor of frl/training/representation/scheduler.py:142-153;
  * commitment weight beta: linear ramp block of configs/vae_v0.yaml:21-27;
  * lambda_vq(step): the `--anneal_vq_*` flags of scripts/train_vqvae.py:433-456.  The `vqvae.annealers` module those flags feed is
    absent from the reference tree, so the curve shapes are this build's definition (documented on `LambdaVQSchedule`).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Mapping, Optional, Sequence, Tuple, Union


# ------------------------------------------------------------------------------------------------------------------
# the primitive
# ------------------------------------------------------------------------------------------------------------------
def _progress(pos: float, start: float, length: float) -> float:
    """Unit progress of `pos` through the window [start, start + length], clamped to [0, 1]; a zero-length window is a step."""
    if pos < start:
        return 0.0
    if length <= 0 or pos >= start + length:
        return 1.0
    return (pos - start) / length


def _ease(u: float, kind: str, k: float = 5.0) -> float:
    """Easing of a unit progress: linear | cosine (half-cosine, zero slope at both ends) | exponential (saturating, steepness k)."""
    if kind == "linear":
        return u
    if kind == "cosine":
        return (1 - math.cos(math.pi * u)) / 2
    if kind == "exponential":
        return (1.0 - math.exp(-k * u)) / (1.0 - math.exp(-k)) if k > 0 else u
    raise ValueError(f"unknown easing {kind!r}")


# ------------------------------------------------------------------------------------------------------------------
# epoch curricula (curriculum.py:16-83)
# ------------------------------------------------------------------------------------------------------------------
def ramp_weight(epoch: int, start_epoch: int, ramp_epochs: int) -> float:
    """Loss weight that is 0 up to and including `start_epoch` and reaches 1 `ramp_epochs` later."""
    return _progress(epoch, start_epoch, ramp_epochs)


def compute_smoothing_min_gate(epoch: int, freeze_until_epoch: int, ramp_epochs: int) -> float:
    """Floor of the spatial gate: 1 (the smoothing conv is the identity) while frozen, released linearly to 0 afterwards."""
    return 1.0 - _progress(epoch, freeze_until_epoch, ramp_epochs)


def compute_input_dropout_rate(schedule_cfg: Union[float, Mapping], epoch: int, total_epochs: int) -> float:
    """A number is a constant rate; a mapping {schedule: constant|linear|cosine, rate | start, end, epochs} ramps it over `epochs`
    epochs (default: the whole run) starting at epoch 0."""
    if not isinstance(schedule_cfg, Mapping):
        return float(schedule_cfg)
    kind = schedule_cfg.get("schedule", "constant")
    if kind == "constant":
        return float(schedule_cfg.get("rate", 0.0))
    if kind not in ("linear", "cosine"):
        raise ValueError(f"Unknown input_dropout schedule: {kind!r}")
    lo, hi = float(schedule_cfg.get("start", 0.0)), float(schedule_cfg.get("end", 0.1))
    u = min(epoch / max(int(schedule_cfg.get("epochs", total_epochs)), 1), 1.0)
    return lo + _ease(u, kind) * (hi - lo) if kind == "linear" else lo + (hi - lo) * _ease(u, kind)


# ------------------------------------------------------------------------------------------------------------------
# learning rate
# ------------------------------------------------------------------------------------------------------------------
def cosine_lr(step: int, total_steps: int, lr: float, min_lr: float) -> float:
    """lr at step 0, min_lr from total_steps on, half-cosine in between (train_vqvae.py:250-253)."""
    return min_lr + (lr - min_lr) * 0.5 * (1.0 + math.cos(math.pi * min(step / max(total_steps, 1), 1.0)))


def warmup_cosine_factor(step: int, warmup_steps: int, total_steps: int, eta_min_factor: float) -> float:
    """LambdaLR multiplier (scheduler.py:142-153): step / warmup (never below 1e-8) during warm-up, then cosine 1 -> eta_min / lr."""
    if 0 < warmup_steps and step < warmup_steps:
        return max(step / warmup_steps, 1e-8)
    u = (step - warmup_steps) / max(total_steps - warmup_steps, 1)
    return eta_min_factor + (1.0 - eta_min_factor) * 0.5 * (1.0 + math.cos(math.pi * u))


# ------------------------------------------------------------------------------------------------------------------
# commitment weight beta (configs/vae_v0.yaml:21-27)
# ------------------------------------------------------------------------------------------------------------------
def beta_schedule(epoch: int, cfg: Optional[Mapping]) -> float:
    """`beta_schedule` block: start_value -> end_value between start_epoch and end_epoch (schedule_type linear | cosine); a missing
    or disabled block means the end value."""
    cfg = cfg or {}
    hi = float(cfg.get("end_value", 1.0))
    if not cfg.get("enabled", False):
        return hi
    lo = float(cfg["start_value"])
    u = _progress(epoch, cfg["start_epoch"], cfg["end_epoch"] - cfg["start_epoch"])
    return float(lo + (hi - lo) * _ease(u, cfg.get("schedule_type", "linear")))


# ------------------------------------------------------------------------------------------------------------------
# lambda_vq(step) (scripts/train_vqvae.py:236-248, 433-456)
# ------------------------------------------------------------------------------------------------------------------
@dataclass
class LambdaVQSchedule:
    """Weight of the VQ loss as a function of the optimizer step, driven by the legacy flags (same names without the
    `anneal_vq_` prefix).  Build definition of the curves:

      disabled           lambda_vq, always
      constant           floor before `start`, ceil from `start` on
      linear | cosine | exponential
                         floor until `start`, eased floor -> ceil over `duration` steps (exponential: steepness `k`), ceil afterwards
      stepwise           `milestones` "step:value": the value of the last milestone reached, floor before the first
      warmup_hold_decay  from `start`: linear floor -> ceil over `warmup` steps, ceil for `hold` steps, linear ceil -> final over
                         `decay` steps, final afterwards (final = None: back to floor)
    `ceil = None` inherits lambda_vq as the target (train_vqvae.py:243-247 drops unset ceil / final before building the schedule).
    """
    lambda_vq: float = 1.0
    enable: bool = False
    schedule: str = "warmup_hold_decay"
    start: int = 0
    duration: int = 0
    floor: float = 0.0
    ceil: Optional[float] = 0.1
    k: float = 5.0
    warmup: int = 10000
    hold: int = 15000
    decay: int = 5000
    final: Optional[float] = 0.08
    milestones: List[Tuple[int, float]] = field(default_factory=list)

    KINDS = ("constant", "linear", "cosine", "exponential", "stepwise", "warmup_hold_decay")

    def __post_init__(self):
        if self.schedule not in self.KINDS:
            raise ValueError(f"anneal_vq_schedule must be one of {self.KINDS}, got {self.schedule!r}")
        self.milestones = sorted((int(s), float(v)) for s, v in self.milestones)

    def __call__(self, step: int) -> float:
        if not self.enable:
            return float(self.lambda_vq)
        lo = float(self.floor)
        hi = float(self.lambda_vq if self.ceil is None else self.ceil)
        if self.schedule == "constant":
            return hi if step >= self.start else lo
        if self.schedule in ("linear", "cosine", "exponential"):
            return lo + (hi - lo) * _ease(_progress(step, self.start, self.duration), self.schedule, self.k)
        if self.schedule == "stepwise":
            value = lo
            for at, v in self.milestones:
                if step >= at:
                    value = v
            return value
        end = lo if self.final is None else float(self.final)          # warmup_hold_decay
        t_hold = self.start + self.warmup
        t_decay = t_hold + self.hold
        if step < t_decay:
            return lo + (hi - lo) * _progress(step, self.start, self.warmup)
        return hi + (end - hi) * _progress(step, t_decay, self.decay)


def parse_milestones(items: Optional[Sequence[str]]) -> List[Tuple[int, float]]:
    """['1000:0.01', '8000:0.1'] -> [(1000, 0.01), (8000, 0.1)] (the `--anneal_vq_milestones` syntax, train_vqvae.py:455-456)."""
    out = []
    for it in items or []:
        if isinstance(it, str):
            s, v = it.split(":")
            out.append((int(s), float(v)))
        else:
            out.append((int(it[0]), float(it[1])))
    return out


def build_lambda_vq(lambda_vq: float, flags: Optional[Mapping] = None) -> LambdaVQSchedule:
    """flags: mapping with the reference's key names (`anneal_vq_enable`, `anneal_vq_schedule`, ...), e.g. vars(args) or the YAML."""
    flags = flags or {}
    kw = {}
    for f in ("enable", "schedule", "start", "duration", "floor", "ceil", "k", "warmup", "hold", "decay", "final"):
        if f"anneal_vq_{f}" in flags:
            kw[f] = flags[f"anneal_vq_{f}"]
    if flags.get("anneal_vq_milestones"):
        kw["milestones"] = parse_milestones(flags["anneal_vq_milestones"])
    return LambdaVQSchedule(lambda_vq=float(lambda_vq), **kw)
