"""Checkpoint files and rotation policy compatible with the reference trainer.

File content (frl/training/train_representation.py:602-614): a dict with `epoch` (the NEXT epoch to run), `model_version`,
`model_config`, `type_in_channels`, `phase_in_channels`, `model_state_dict`, `optimizer_state_dict`, `scheduler_state_dict` and the
epoch's metrics as top-level keys -- which is also what `RepresentationModel.from_checkpoint` reads back
(frl/models/representation.py:442-490).  Rotation (frl/training/representation/checkpointing.py:22-150): `encoder_last.pt` every
epoch, `encoder_epoch_NNN.pt` every n-th epoch (never pruned), and the k best epochs by the monitored metric as
`encoder_best_<rank>_epoch_NNN.pt`, re-ranked after every insertion; a non-finite metric never enters the list.
tests/golden/checkpoint_policy.json holds directory listings produced by the reference manager itself.

Files are written with `torch.save` and read with `torch.load(weights_only=True)`: they hold tensors and plain Python values only.
"""
from __future__ import annotations

import math
import os
import re
from dataclasses import dataclass
from pathlib import Path
from typing import Callable, Dict, List, Optional, Tuple

import torch


@dataclass
class CheckpointPolicy:
    monitor: str = "val/loss"
    mode: str = "min"                       # "min" | "max"
    save_last: bool = True
    save_every_n_epochs: int = 10
    save_top_k: int = 3
    monitor_start_epoch: int = 0

    def __post_init__(self):
        if self.mode not in ("min", "max"):
            raise ValueError("mode must be 'min' or 'max'")
        if self.save_every_n_epochs <= 0 or self.save_top_k <= 0:
            raise ValueError("save_every_n_epochs and save_top_k must be positive")


def _torch_save(state: dict, path) -> None:
    torch.save(state, path)


def _torch_load(path) -> dict:
    return torch.load(path, map_location="cpu", weights_only=True)


class CheckpointManager:
    """`save(epoch, state, metrics)` once per finished epoch; `restore_top_k()` after a restart."""

    _BEST = re.compile(r"^encoder_best_.*epoch_(\d+)\.pt$")

    def __init__(self, ckpt_dir, policy: CheckpointPolicy, save_fn: Callable = _torch_save, load_fn: Callable = _torch_load):
        self.dir, self.policy, self.save_fn, self.load_fn = Path(ckpt_dir), policy, save_fn, load_fn
        self.dir.mkdir(parents=True, exist_ok=True)
        self.best: List[Tuple[float, Path]] = []            # best first

    # ranking key: better values first, non-finite ones (only ever read back from disk) last
    def _key(self, value: float) -> float:
        if not math.isfinite(value):
            return math.inf
        return value if self.policy.mode == "min" else -value

    def restore_top_k(self) -> None:
        for path in sorted(self.dir.glob("encoder_best_*.pt")):
            try:
                value = float(self.load_fn(path).get(self.policy.monitor, math.nan))
            except Exception:
                continue                                    # unreadable file: not part of the list
            self.best.append((value, path))

    def save(self, epoch: int, state: dict, metrics: Dict[str, float]) -> None:
        pol = self.policy
        if pol.monitor not in metrics:
            raise KeyError(f"Checkpoint monitor '{pol.monitor}' not found in epoch_metrics. Available keys: {list(metrics.keys())}")
        value = float(metrics[pol.monitor])
        if pol.save_last:
            self.save_fn(state, self.dir / "encoder_last.pt")
        if (epoch + 1) % pol.save_every_n_epochs == 0:
            self.save_fn(state, self.dir / f"encoder_epoch_{epoch + 1:03d}.pt")
        if not math.isfinite(value) or epoch < pol.monitor_start_epoch:
            return
        self.best.sort(key=lambda e: self._key(e[0]))       # stable: equal values keep their order of arrival
        if len(self.best) >= pol.save_top_k and not self._key(value) < self._key(self.best[-1][0]):
            return                                          # not strictly better than the worst kept checkpoint
        fresh = self.dir / f"encoder_best_epoch_{epoch + 1:03d}.pt"
        self.save_fn(state, fresh)
        self.best.append((value, fresh))
        self.best.sort(key=lambda e: self._key(e[0]))
        for _, path in self.best[pol.save_top_k:]:
            if path.exists():
                path.unlink()
        del self.best[pol.save_top_k:]
        # re-rank: move everything aside first so that no target name is still occupied
        staged = []
        for rank, (val, path) in enumerate(self.best, 1):
            ep = self._BEST.match(path.name).group(1)
            aside = self.dir / f"_tmp_rank_{rank}_{ep}.pt"
            path.rename(aside)
            staged.append((val, aside, self.dir / f"encoder_best_{rank}_epoch_{ep}.pt"))
        self.best = []
        for val, aside, target in staged:
            aside.rename(target)
            self.best.append((val, target))


def build_checkpoint_state(model, optimizer, epoch: int, metrics: Optional[Dict[str, float]] = None, model_config: Optional[dict] = None,
                           scheduler_state: Optional[dict] = None) -> dict:
    """The reference's checkpoint dict for a finished `epoch` (0-based); tensors are moved to the CPU."""
    from ..models.representation import RepresentationModel
    state = {"epoch": epoch + 1, "model_version": RepresentationModel.VERSION, "model_config": model_config,
             "type_in_channels": getattr(model, "type_in_channels", None), "phase_in_channels": getattr(model, "phase_in_channels", None),
             "model_state_dict": {k: v.detach().cpu() for k, v in model.state_dict().items()},
             "optimizer_state_dict": _to_cpu(optimizer.state_dict()), "scheduler_state_dict": scheduler_state or {}}
    state.update({k: float(v) for k, v in (metrics or {}).items()})
    return state


def _to_cpu(obj):
    if torch.is_tensor(obj):
        return obj.detach().cpu()
    if isinstance(obj, dict):
        return {k: _to_cpu(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_cpu(v) for v in obj)
    return obj


def resume_from_checkpoint(model, optimizer, ckpt_dir, resume: Optional[str] = None, no_resume: bool = False,
                           manager: Optional[CheckpointManager] = None, device="cpu"):
    """Manual (`resume` path) or automatic (`encoder_last.pt` in `ckpt_dir`) restart, as the reference's `resume_from_checkpoint`
    (checkpointing.py:153-217): returns (start_epoch, resume_lr, scheduler_state); the scheduler state is only returned for the
    automatic path, which also rebuilds the manager's top-k list."""
    start_epoch, resume_lr, scheduler_state = 0, optimizer.param_groups[0]["lr"], None
    auto = Path(ckpt_dir) / "encoder_last.pt"
    path = resume if resume else (auto if (not no_resume and auto.exists()) else None)
    if path is None:
        return start_epoch, resume_lr, scheduler_state
    ckpt = torch.load(os.fspath(path), map_location=device, weights_only=True)
    model.load_state_dict(ckpt["model_state_dict"])
    optimizer.load_state_dict(ckpt["optimizer_state_dict"])
    start_epoch = int(ckpt["epoch"])
    resume_lr = optimizer.param_groups[0]["lr"]
    if not resume:
        if manager is not None:
            manager.restore_top_k()
        scheduler_state = ckpt.get("scheduler_state_dict")
    return start_epoch, resume_lr, scheduler_state
