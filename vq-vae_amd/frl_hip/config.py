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
    extra: Dict[str, Any] = field(default_factory=dict)


def load_vae_config(path: str) -> VAEConfig:
    with open(path) as f:
        raw = yaml.safe_load(f) or {}
    cfg = VAEConfig()
    known = set(VAEConfig.__dataclass_fields__) - {"optimizer", "extra"}
    for k, v in raw.items():
        if k == "optimizer":
            o = dict(v or {})
            cfg.optimizer = OptimizerConfig(name=o.get("name", "adam"), lr=float(o.get("lr", 1e-4)),
                                            weight_decay=float(o.get("weight_decay", 0.0)),
                                            scheduler=dict(o.get("scheduler", {}) or {}))
            if "eta_min" in cfg.optimizer.scheduler:
                cfg.optimizer.scheduler["eta_min"] = float(cfg.optimizer.scheduler["eta_min"])
        elif k in known:
            setattr(cfg, k, v)
        else:
            cfg.extra[k] = v        # debug_window*, full_block_dims, ...: zarr-windowing keys, kept verbatim
    return cfg


def load_model_config(path: str) -> dict:
    with open(path) as f:
        return yaml.safe_load(f)
