"""Config surface of the VQ-VAE trainer.

`configs/vae_v0.yaml` (reference configs/vae_v0.yaml:1-44) has no consumer in the reference tree; this loader accepts its
flat keys (batch_size, num_epochs, beta, optimizer{name,lr,weight_decay,scheduler{name,T_max_epochs,eta_min}},
beta_schedule{...}, num_workers, pin_memory, run_root, experiment_name, ckpt_dir, ...) plus the VQ flags of the legacy
CLI (scripts/train_vqvae.py:410-436: codebook_size, emb_dim, beta, quantizer, ema_decay, ema_eps, lambda_vq).
Model YAMLs (frl/config/frl_repr_model_v1.yaml) are read raw and handed to RepresentationModel.from_config.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict

import yaml


@dataclass
class OptimizerConfig:
    name: str = "adam"
    lr: float = 1e-4
    weight_decay: float = 0.0
    scheduler: Dict[str, Any] = field(default_factory=lambda: {"name": "cosine", "T_max_epochs": 150, "eta_min": 1e-6})


@dataclass
class VAEConfig:
    zarr_path: str = ""
    patch_size: int = 256
    batch_size: int = 4
    num_epochs: int = 200
    beta: float = 0.1
    lambda_cat: float = 1.0
    optimizer: OptimizerConfig = field(default_factory=OptimizerConfig)
    beta_schedule: Dict[str, Any] = field(default_factory=dict)
    num_workers: int = 0
    pin_memory: bool = True
    run_root: str = "runs"
    experiment_name: str = "vae_v0"
    ckpt_dir: str = "checkpoints"
    # VQ keys of the legacy CLI
    codebook_size: int = 256
    emb_dim: int = 64
    quantizer: str = "st"
    ema_decay: float = 0.99
    ema_eps: float = 1e-5
    lambda_vq: float = 1.0
    hidden: int = 128                     # decoder width (train_vqvae.py:413)
    min_lr: float = -1.0                  # < 0: optimizer.scheduler.eta_min (vae_v0.yaml) / 3e-5 of the legacy CLI when that is absent too
    lambda_cont: float = 1.0              # weight of the continuous reconstruction (= lambda_recon of the tile VQ-VAE)
    lambda_canopy: float = 1.0            # legacy head weight, accepted for CLI / YAML compatibility (no canopy head on the tile path)
    clip_grad: float = 1.0
    steps_per_epoch: int = 0              # 0: len(loader) -- the caller passes it to build_trainer_from_config
    anneal_vq: Dict[str, Any] = field(default_factory=dict)    # every `anneal_vq_*` key, verbatim (train_vqvae.py:433-456)
    extra: Dict[str, Any] = field(default_factory=dict)


def load_vae_config(path: str) -> VAEConfig:
    with open(path) as f:
        raw = yaml.safe_load(f) or {}
    cfg = VAEConfig()
    known = set(VAEConfig.__dataclass_fields__) - {"optimizer", "extra", "anneal_vq"}
    for k, v in raw.items():
        if k == "optimizer":
            o = dict(v or {})
            cfg.optimizer = OptimizerConfig(name=o.get("name", "adam"), lr=float(o.get("lr", 1e-4)),
                                            weight_decay=float(o.get("weight_decay", 0.0)),
                                            scheduler=dict(o.get("scheduler", {}) or {}))
            if "eta_min" in cfg.optimizer.scheduler:
                cfg.optimizer.scheduler["eta_min"] = float(cfg.optimizer.scheduler["eta_min"])
        elif k.startswith("anneal_vq_"):
            cfg.anneal_vq[k] = v
        elif k in known:
            setattr(cfg, k, v)
        else:
            cfg.extra[k] = v        # debug_window*, full_block_dims, ...: zarr-windowing keys, kept verbatim
    return cfg


def load_model_config(path: str) -> dict:
    with open(path) as f:
        return yaml.safe_load(f)


def build_trainer_from_config(cfg: VAEConfig, steps_per_epoch: int, in_features: int = 64, device=None, compute_dtype=None,
                              model_kwargs: Dict[str, Any] = None):
    """The consumer `configs/vae_v0.yaml` lacks in the reference: VAEConfig -> (VQVAE, VQVAETrainer, CheckpointManager, run_dir).

    Wiring (reference sources of each rule):
      * model: codebook_size / emb_dim / beta / hidden / quantizer / ema_* as the legacy constructor call (scripts/train_vqvae.py:183-195);
      * optimizer: AdamW, two groups with the codebook free of weight decay, betas (0.9, 0.95) (train_vqvae.py:221-228), lr and
        weight_decay from `optimizer`; clip_grad as the global-norm clip (train_vqvae.py:334);
      * LR: cosine from lr to eta_min over T_max_epochs * steps_per_epoch optimizer steps, stepped per batch (vae_v0.yaml:13-19,
        train_vqvae.py:250-253);
      * beta ramp per epoch from `beta_schedule` (vae_v0.yaml:21-27) -- call trainer.set_epoch(e) at every epoch start;
      * lambda_vq(step) from the `anneal_vq_*` keys (train_vqvae.py:236-248, 433-456);
      * checkpoints under run_root / experiment_name / ckpt_dir (vae_v0.yaml:39-44) with the reference's rotation policy.
    """
    import os

    import torch

    from .models import VQVAE
    from .training.checkpointing import CheckpointManager, CheckpointPolicy
    from .training.schedules import build_lambda_vq
    from .training.trainer import VQVAETrainer

    if cfg.optimizer.name.lower() not in ("adam", "adamw"):
        raise ValueError(f"optimizer.name must be adam or adamw, got {cfg.optimizer.name!r}")
    sched = cfg.optimizer.scheduler or {}
    if sched.get("name", "cosine") != "cosine":
        raise ValueError(f"optimizer.scheduler.name must be cosine, got {sched.get('name')!r}")
    mk = dict(model_kwargs or {})
    if compute_dtype is not None:
        mk["compute_dtype"] = compute_dtype
    model = VQVAE(in_features=in_features, codebook_size=cfg.codebook_size, emb_dim=cfg.emb_dim, beta=cfg.beta, hidden=cfg.hidden,
                  quantizer=cfg.quantizer, ema_decay=cfg.ema_decay, ema_eps=cfg.ema_eps, lambda_recon=cfg.lambda_cont,
                  lambda_vq=cfg.lambda_vq, **mk)
    if device is not None:
        model = model.to(device)
    epochs_lr = int(sched.get("T_max_epochs", cfg.num_epochs))
    spe = int(cfg.steps_per_epoch or steps_per_epoch)
    if spe <= 0:
        raise ValueError("steps_per_epoch must be positive")
    min_lr = cfg.min_lr if cfg.min_lr >= 0 else float(sched.get("eta_min", 3e-5))
    bs = cfg.beta_schedule if (cfg.beta_schedule or {}).get("enabled", False) else None
    trainer = VQVAETrainer(model, lr=cfg.optimizer.lr, min_lr=min_lr, weight_decay=cfg.optimizer.weight_decay, max_norm=cfg.clip_grad,
                           total_steps=epochs_lr * spe, beta_schedule_cfg=bs,
                           lambda_vq_schedule=build_lambda_vq(cfg.lambda_vq, cfg.anneal_vq) if cfg.anneal_vq else None)
    trainer.set_epoch(0)
    run_dir = os.path.join(cfg.run_root, cfg.experiment_name)
    ckpt = CheckpointManager(os.path.join(run_dir, cfg.ckpt_dir), CheckpointPolicy(monitor="train/loss"))
    return model, trainer, ckpt, run_dir
