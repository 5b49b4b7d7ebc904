"""Train-step skeleton of the reference (frl/training/representation/step.py:121-125,1057-1090; loops.py:97-110):
zero_grad -> forward -> total loss -> isfinite guard -> backward -> clip_grad_norm_(1.0) -> AdamW.step -> per-batch LR step.

Optimizer wiring follows the legacy VQ-VAE trainer (scripts/train_vqvae.py:221-253): two AdamW groups (codebook without
weight decay), betas (0.9, 0.95), cosine LR from lr to min_lr.  With torch.distributed initialised the gradients are
all-reduced by `parallel.BucketedGradAllReduce` (RCCL on GPUs, gloo in the CPU tests).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.distributed as dist

from ..parallel import BucketedGradAllReduce
from .schedules import LambdaVQSchedule, beta_schedule, cosine_lr


class VQVAETrainer:
    def __init__(self, model, lr: float = 1e-4, min_lr: float = 1e-6, weight_decay: float = 0.01, max_norm: float = 1.0,
                 total_steps: int = 1000, betas=(0.9, 0.95), check_finite: bool = True, fused_optimizer: bool = True,
                 beta_schedule_cfg: Optional[dict] = None, lambda_vq_schedule: Optional[LambdaVQSchedule] = None):
        self.model = model
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        cb = [p for n, p in named if "quant" in n and "codebook" in n]
        rest = [p for n, p in named if not ("quant" in n and "codebook" in n)]
        groups = [{"params": rest, "weight_decay": weight_decay}]
        if cb:
            groups.append({"params": cb, "weight_decay": 0.0})
        on_gpu = bool(rest) and rest[0].is_cuda
        self.hip_opt = bool(fused_optimizer and on_gpu)
        if self.hip_opt:                                   # clip + AdamW in two HIP launches (csrc/optim.hip)
            from .optim import HipAdamW
            self.opt = HipAdamW(groups, lr=lr, betas=betas)
        else:
            self.opt = torch.optim.AdamW(groups, lr=lr, betas=betas)
        self.params = [p for _, p in named]
        self.lr, self.min_lr, self.total_steps, self.max_norm = lr, min_lr, total_steps, max_norm
        self.check_finite = check_finite
        self.beta_schedule_cfg = beta_schedule_cfg
        self.lambda_vq_schedule = lambda_vq_schedule           # lambda_vq(step) of scripts/train_vqvae.py:236-248,324 (None: constant)
        self.step_idx = 0
        self.epoch = 0
        self.skipped = 0
        distributed = dist.is_available() and dist.is_initialized()
        if distributed and dist.get_world_size() > 1:
            # as DistributedDataParallel does at construction: every rank starts from rank 0's parameters and buffers, whatever it seeded
            with torch.no_grad():
                for t in list(model.parameters()) + list(model.buffers()):
                    dist.broadcast(t.data, src=0)
        self.reducer = BucketedGradAllReduce(named) if distributed else None
        # codebook state that forward() would mutate (EMA running averages, the manager's usage window) is committed after the
        # isfinite guard instead: one non-finite batch must not poison it (step.py:1057-1074 skips the whole batch)
        if check_finite and hasattr(model, "set_defer_codebook_hooks"):
            model.set_defer_codebook_hooks(True)
        if self.reducer is not None and on_gpu and getattr(model, "concurrent_phase", False) and hasattr(model, "phase_stream"):
            self.reducer.extra_streams.append(model.phase_stream(rest[0].device))

    @property
    def n_skipped(self) -> int:
        """Batches skipped by the isfinite guard (reads the device counter when the HIP optimizer is in use)."""
        return self.opt.applied_and_skipped[1] if self.hip_opt else self.skipped

    def set_epoch(self, epoch: int):
        """Per-epoch curricula: beta schedule of configs/vae_v0.yaml:21-27."""
        self.epoch = epoch
        if self.beta_schedule_cfg is not None and hasattr(self.model, "quant"):
            self.model.quant.beta = beta_schedule(epoch, self.beta_schedule_cfg)

    def _all_finite(self, loss: torch.Tensor) -> bool:
        ok = torch.isfinite(loss.detach()).float()
        if self.reducer is not None:            # every rank must take the same branch or the collectives deadlock
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        return bool(ok.item() > 0)

    def step(self, tile: torch.Tensor, mask: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        lr_now = cosine_lr(self.step_idx, self.total_steps, self.lr, self.min_lr)
        for g in self.opt.param_groups:
            g["lr"] = lr_now
        if self.lambda_vq_schedule is not None:                # loss = lambda_recon L + lambda_vq(step) (L_codebook + beta L_commit)
            self.model.lambda_vq = self.lambda_vq_schedule(self.step_idx)
        self.model.train()
        self.opt.zero_grad(set_to_none=True)
        out = self.model.forward_tiles(tile, mask)
        loss = out["loss"]
        ok = None
        if self.check_finite:
            if self.hip_opt:
                # step.py:1057-1074 (skip the batch on a non-finite loss) evaluated on the device: the flag gates the optimizer
                # kernels, so the host never waits for the loss and keeps queueing the next step
                # x * 0 == 0 holds exactly for finite x and fails for NaN / +-inf: isfinite in three tiny kernels instead of six
                ok = (loss.detach().float() * 0.0 == 0.0).float().reshape(1)
                if self.reducer is not None and self.reducer.active:
                    self.reducer.flag_src = 1.0 - ok               # rides in the last gradient bucket: every rank takes the same decision
            elif not self._all_finite(loss):
                self.skipped += 1
                if self.reducer is not None:
                    self.reducer.reset()
                if hasattr(self.model, "drop_codebook_hooks"):
                    self.model.drop_codebook_hooks()
                self.step_idx += 1                                     # loops.py:110: the scheduler steps after every batch, skipped or not
                out["lr"] = lr_now
                return out
        loss.backward()
        if self.reducer is not None:
            self.reducer.finish(scatter=not self.hip_opt)   # HipAdamW reads the flat buckets in place
        if self.hip_opt:
            grads = None
            if self.reducer is not None and self.reducer.active:   # averaged gradients are read straight from the all-reduce buckets
                fg = self.reducer.flat_grads()
                # a parameter without a gradient (unused branch) stays without one: torch.optim.AdamW -- and HipAdamW on a single GPU --
                # skip it entirely, so its zero-filled bucket slot must not turn into a weight-decay-only update here
                grads = [fg[id(p)] if p.grad is not None else None for p in self.opt.params]
                if ok is not None:
                    ok = (self.reducer.flag_result() == 0).float()   # 1 <=> no rank reported a non-finite loss
            out["grad_norm"] = self.opt.step(self.max_norm, grads, ok)
            if getattr(self.model, "defer_codebook_hooks", False):
                self.model.commit_codebook_hooks(ok)                   # EMA / usage window: gated by the same device flag
        else:
            out["grad_norm"] = torch.nn.utils.clip_grad_norm_(self.params, self.max_norm)
            self.opt.step()
            if getattr(self.model, "defer_codebook_hooks", False):
                self.model.commit_codebook_hooks(None)                 # (this path checked the loss on the host already)
        mgr = getattr(self.model, "codebook_manager", None)
        if mgr is not None and hasattr(mgr, "after_step"):                   # dead-code revival every `reset_every` steps, on the device
            mgr.after_step(self.model.quant, self.opt)
        self.step_idx += 1
        out["lr"] = lr_now
        return out
