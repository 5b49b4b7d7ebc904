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
                 beta_schedule_cfg: Optional[dict] = None, lambda_vq_schedule: Optional[LambdaVQSchedule] = None,
                 pack_cache: bool = True, defer_reductions: bool = True):
        self.model = model
        self.defer_reductions = bool(defer_reductions)       # single process: one launch for all weight-gradient slab reductions
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
        # Weight-image cache (ops.PackCache): the packed MFMA fragment images of all conv weights live in an arena of this trainer; the
        # calls of a step find them there (no per-call packing launch) and ONE launch behind the optimizer rewrites all of them.
        self.pack_cache = None
        self._pack_versions = None
        if pack_cache and self.hip_opt:
            from .. import ops
            self.pack_cache = ops.PackCache(rest[0].device)
        self.lr, self.min_lr, self.total_steps, self.max_norm = lr, min_lr, total_steps, max_norm
        self.check_finite = check_finite
        self.beta_schedule_cfg = beta_schedule_cfg
        self.lambda_vq_schedule = lambda_vq_schedule           # lambda_vq(step) of scripts/train_vqvae.py:236-248,324 (None: constant)
        if lambda_vq_schedule is not None and self.hip_opt and hasattr(model, "lambda_vq_dev"):
            # the scheduled weight lives in a device scalar the loss head reads at run time: a captured step follows the schedule
            model.lambda_vq_dev = torch.full((1,), float(lambda_vq_schedule(0)), dtype=torch.float32, device=rest[0].device)
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

    def opt_lr_dev_sync(self) -> bool:
        """Eager steps after a graph was captured: the optimizer reads the learning rate from its device word, keep it current."""
        if self.hip_opt and self.opt.lr_dev is not None:
            self.opt.lr_dev.fill_(float(self.opt.param_groups[0]["lr"]))
            return True
        return False

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

    # ------------------------------------------------------------------------------------------------------------------
    # The whole step as ONE hipGraph launch.  A step is ~190 kernel launches through ctypes plus ~400 small ATen calls; at 256 tiles
    # the host needs about as long to queue them as the GPU needs to run them, so every kernel that gets faster moves the step from
    # GPU-bound to host-bound.  Captured once (forward on both streams, autograd backward, gradient clip + AdamW, codebook hooks,
    # fragment-image refresh), a step costs the host one copy into the static input, one word for the learning rate and one replay.
    # Everything the step decides per batch already lives on the device (isfinite flag, update counters); the learning rate moves
    # there too (HipAdamW.lr_dev).  Host-side values that are baked into the capture (beta, lambda_recon / lambda_vq) re-capture
    # when they change; a lambda_vq(step) schedule that changes every step keeps the eager path.
    # ------------------------------------------------------------------------------------------------------------------
    def graph_supported(self) -> bool:
        """The step can be captured: device-side optimizer and isfinite guard; a lambda_vq(step) schedule needs the model's device scalar;
        data parallel needs gradients on the GPU (RCCL collectives are captured with the step, gloo ones cannot be)."""
        if not (self.hip_opt and self.check_finite):
            return False
        if self.lambda_vq_schedule is not None and getattr(self.model, "lambda_vq_dev", None) is None:
            return False
        if self.reducer is not None and self.reducer.active:
            return bool(self.reducer.on_gpu and (self.reducer.world == 1 or dist.get_backend(self.reducer.group) == "nccl"))
        return True

    def _set_lambda_vq(self) -> None:
        """loss = lambda_recon L + lambda_vq(step) (L_codebook + beta L_commit): host attribute (reports, eager CPU paths) and device scalar"""
        if self.lambda_vq_schedule is None:
            return
        v = float(self.lambda_vq_schedule(self.step_idx))
        self.model.lambda_vq = v
        if getattr(self.model, "lambda_vq_dev", None) is not None:
            self.model.lambda_vq_dev.fill_(v)

    def _graph_key(self, tile, mask):
        q = getattr(self.model, "quant", None)
        lam = None if getattr(self.model, "lambda_vq_dev", None) is not None else getattr(self.model, "lambda_vq", None)   # (device scalar: not baked)
        return (tuple(tile.shape), tile.dtype, None if mask is None else (tuple(mask.shape), mask.dtype),
                getattr(q, "beta", None), lam, getattr(self.model, "lambda_recon", None),
                getattr(self.model, "concurrent_phase", None), self.max_norm)

    def step_graphed(self, tile: torch.Tensor, mask: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """Same step as `step`, replayed from a captured hipGraph (falls back to `step` when graph_supported() is False).
        The returned tensors are the graph's static outputs: they are overwritten by the next call."""
        if not self.graph_supported():
            return self.step(tile, mask)
        # A graph reads its input at a fixed address.  Inputs that come from a small ring of device buffers (TilePrefetcher slots,
        # the benchmark's tile pool) get one graph per buffer -- all of them share one memory pool, they never run concurrently --
        # so that no copy into a staging tensor is needed; beyond MAX_GRAPHS distinct buffers the input is copied into a staging tile of the trainer's own.
        self._set_lambda_vq()
        key = self._graph_key(tile, mask)
        graphs = self.__dict__.setdefault("_graphs", {})
        if graphs and next(iter(graphs.values()))["key"] != key:
            graphs.clear()                                          # shapes / baked scalars changed: start over
            self._graph_pool = None
        slot = (tile.data_ptr(), None if mask is None else mask.data_ptr())
        g = graphs.get(slot)
        if g is None:
            if sum(1 for k in graphs if k != "staging") < self.MAX_GRAPHS:
                g = graphs[slot] = self._capture(tile, mask, key)
            else:
                # more distinct input addresses than graphs (a loader that yields a fresh tensor per step, a ring of more than MAX_GRAPHS
                # slots): ONE further graph on buffers this trainer owns; the caller's tensors are only ever read
                g = graphs.get("staging")
                if g is None:
                    g = graphs["staging"] = self._capture(tile.clone(), None if mask is None else mask.clone(), key)
        # State edited behind the trainer's back since the last step (load_state_dict on resume, manual edits) -- the eager path checks
        # parameter versions in step() and the quantizer re-keys its image in forward(); a replay runs no Python, so check here.
        self._images_current()
        if hasattr(self.model, "_quantizers"):
            for qz in self.model._quantizers():
                if qz._prepared is not None:
                    qz.prepared(qz._prepared[0][2], 1 << 20)       # rebuilt (eagerly, into the same buffer) only when the key moved
        lr_now = cosine_lr(self.step_idx, self.total_steps, self.lr, self.min_lr)
        # the scalar travels as a kernel argument of the fill: no host word that a later step could overwrite before the copy engine
        # has read it (the un-synchronised loop of bench.py queues many steps ahead of the device)
        self.opt.lr_dev.fill_(lr_now)
        if g["tile"].data_ptr() != tile.data_ptr():
            g["tile"].copy_(tile, non_blocking=True)
        if mask is not None and g["mask"].data_ptr() != mask.data_ptr():
            g["mask"].copy_(mask, non_blocking=True)
        g["graph"].replay()
        mgr = getattr(self.model, "codebook_manager", None)
        if mgr is not None and hasattr(mgr, "after_step"):         # host-scheduled (every reset_every steps): stays outside the graph
            # the rows / guard flag a revival would draw from are static outputs of THE GRAPH THAT JUST RAN: hand them to the manager
            # before every after_step (it forgets them after a revival, and another graph's replay leaves its own buffers current)
            if g.get("mgr_z") is not None:
                mgr._z, mgr._z_ok = g["mgr_z"], g["mgr_ok"]
            if mgr.after_step(self.model.quant, self.opt):
                for qz in self.model._quantizers():
                    if qz._prepared is not None:
                        qz.prepared(qz._prepared[0][2], 1 << 20)
        self.step_idx += 1
        out = g["out"]
        out["lr"] = lr_now
        return out

    MAX_GRAPHS = 8

    def _capture(self, tile, mask, key):
        dev = tile.device
        static_tile, static_mask = tile, mask                      # the caller's buffers ARE the static inputs (kept alive by the graph record)
        if self.opt.lr_dev is None:
            self.opt.lr_dev = torch.zeros(1, dtype=torch.float32, device=dev)
        self.opt.lr_dev.fill_(cosine_lr(self.step_idx, self.total_steps, self.lr, self.min_lr))
        mgr = getattr(self.model, "codebook_manager", None)
        saved_after = None
        if mgr is not None and hasattr(mgr, "after_step"):         # the revival pass is scheduled by the host: not part of the body
            saved_after, mgr.after_step = mgr.after_step, (lambda *a, **k: False)
        try:
            # Warm-up on a side stream (allocator pools, workspaces, lazily built tables), with the state restored afterwards: the
            # capture must not consume optimizer steps.  Three eager steps move parameters, moments and counters -- snapshot them.
            snap = self._snapshot()
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(2):
                    self._step_body(static_tile, static_mask)
            torch.cuda.current_stream(dev).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, pool=getattr(self, "_graph_pool", None)):
                out = self._step_body(static_tile, static_mask)
            if getattr(self, "_graph_pool", None) is None:
                self._graph_pool = graph.pool()
            mgr_z, mgr_ok = (getattr(mgr, "_z", None), getattr(mgr, "_z_ok", None)) if mgr is not None else (None, None)
            self._restore(snap)
        finally:
            if saved_after is not None:
                mgr.after_step = saved_after
        return dict(key=key, graph=graph, tile=static_tile, mask=static_mask, out=out, mgr_z=mgr_z, mgr_ok=mgr_ok)

    def _snapshot(self):
        m, o = self.model, self.opt
        st = dict(params=[p.detach().clone() for p in o.params], m=[t.clone() for t in o.exp_avg], v=[t.clone() for t in o.exp_avg_sq],
                  counters=o.counters.clone(), step_count=o.step_count, lag=list(o.lag),
                  buffers=[b.detach().clone() for b in m.buffers()])
        mgr = getattr(m, "codebook_manager", None)
        if mgr is not None and hasattr(mgr, "window"):
            # (None before the manager's first update: the warm-up steps allocate it, and it must come back EMPTY)
            st["window"] = None if mgr.window is None else mgr.window.clone()
            st["revived"] = None if getattr(mgr, "revived", None) is None else mgr.revived.clone()
        return st

    def _restore(self, st):
        m, o = self.model, self.opt
        with torch.no_grad():
            for p, q in zip(o.params, st["params"]):
                p.copy_(q)
            for a, b in zip(o.exp_avg, st["m"]):
                a.copy_(b)
            for a, b in zip(o.exp_avg_sq, st["v"]):
                a.copy_(b)
            o.counters.copy_(st["counters"])
            for b, q in zip(m.buffers(), st["buffers"]):
                b.copy_(q)
        o.step_count, o.lag = st["step_count"], list(st["lag"])
        o._key = None
        self._images_refresh()                                     # the weight images follow the restored parameters
        mgr = getattr(m, "codebook_manager", None)
        if mgr is not None and "window" in st and mgr.window is not None:
            if st["window"] is None:
                mgr.window.zero_()
            else:
                mgr.window.copy_(st["window"])
            if getattr(mgr, "revived", None) is not None:
                if st["revived"] is None:
                    mgr.revived.zero_()
                else:
                    mgr.revived.copy_(st["revived"])
        if hasattr(m, "_quantizers"):
            for qz in m._quantizers():
                qz.drop_ema()
                if qz._prepared is not None:
                    # state that crosses steps must keep ONE address for every graph: the image written behind the optimizer of step n
                    # (by whichever graph ran it) is read by the forward of step n + 1 (possibly another graph).  Keep the buffer,
                    # rebuild its content for the restored codebook.
                    dtype = qz._prepared[0][2]
                    qz._prepared = ((None, None, dtype), qz._prepared[1])
                    qz.prepared(dtype, 1 << 20)

    def _images_current(self) -> None:
        """Weights modified by anyone but this trainer's optimizer since the last step (load_state_dict, manual edits): rewrite the images."""
        if self.pack_cache is None:
            return
        vers = [p._version for p in self.params]
        if vers != self._pack_versions:
            self.pack_cache.refresh()
            self._pack_versions = vers

    def _images_refresh(self) -> None:
        if self.pack_cache is not None:
            self.pack_cache.refresh()
            self._pack_versions = [p._version for p in self.params]

    @staticmethod
    def _finite_flag(out):
        """device float [1]: 1.0 when the loss is finite.  The model's loss head delivers it with the loss (one launch); otherwise
        x * 0 == 0 (exact for finite x, false for NaN / +-inf) in three tiny kernels."""
        ok = out.get("loss_ok")
        if ok is not None:
            return ok
        return (out["loss"].detach().float() * 0.0 == 0.0).float().reshape(1)

    def _backward(self, loss):
        """loss.backward(); single-process training parks the slab reductions of the weight-gradient kernels and runs them in one launch
        (ops.deferred_reductions).  Data parallel keeps them where they are: the bucket hooks read each gradient the moment autograd
        delivers it."""
        if self.defer_reductions and self.hip_opt and not (self.reducer is not None and self.reducer.active):
            from .. import ops
            with ops.deferred_reductions(self.opt.params):
                loss.backward()
        else:
            loss.backward()

    def _step_body(self, tile, mask):
        """forward -> device isfinite flag -> backward -> clip + AdamW -> codebook hooks -> fragment-image refresh (no host-side
        schedule, no host sync): the part of `step` that a graph can hold."""
        self.model.train()
        self.opt.zero_grad(set_to_none=True)
        dp = self.reducer is not None and self.reducer.active
        if dp:
            self.reducer.reset()                                   # (hooks count down per step; the training stream is the capturing one)

        def fwd_bwd():
            out_ = self.model.forward_tiles(tile, mask)
            ok_ = self._finite_flag(out_)
            if dp:
                self.reducer.flag_src = 1.0 - ok_                  # rides in the last gradient bucket: every rank takes the same decision
            self._backward(out_["loss"])                           # data parallel: the bucket hooks pack + all-reduce on the side stream
            return out_, ok_

        if self.pack_cache is not None:
            with self.pack_cache:
                out, ok = fwd_bwd()
        else:
            out, ok = fwd_bwd()
        grads = None
        if dp:
            # the collectives (RCCL) and the stream joins are part of the capture; the optimizer reads the averaged gradients in place
            self.reducer.finish(scatter=False)
            fg = self.reducer.flat_grads()
            grads = [fg[id(p)] if p.grad is not None else None for p in self.opt.params]
            ok = (self.reducer.flag_result() == 0).float()         # 1 <=> no rank reported a non-finite loss
        out["grad_norm"] = self.opt.step(self.max_norm, grads, ok)
        self._images_refresh()
        if getattr(self.model, "defer_codebook_hooks", False):
            self.model.commit_codebook_hooks(ok)
        if hasattr(self.model, "_quantizers"):
            for qz in self.model._quantizers():
                if qz._prepared is not None:
                    qz.prepared(qz._prepared[0][2], 1 << 20)
        return out

    def step(self, tile: torch.Tensor, mask: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        lr_now = cosine_lr(self.step_idx, self.total_steps, self.lr, self.min_lr)
        for g in self.opt.param_groups:
            g["lr"] = lr_now
        if self.opt_lr_dev_sync():
            pass
        self._set_lambda_vq()
        self.model.train()
        self.opt.zero_grad(set_to_none=True)
        self._images_current()
        if self.pack_cache is not None:
            self.pack_cache.__enter__()
        try:
            return self._step_eager(tile, mask, lr_now)
        finally:
            if self.pack_cache is not None:
                self.pack_cache.__exit__(None, None, None)

    def _step_eager(self, tile, mask, lr_now):
        out = self.model.forward_tiles(tile, mask)
        loss = out["loss"]
        ok = None
        if self.check_finite:
            if self.hip_opt:
                # step.py:1057-1074 (skip the batch on a non-finite loss) evaluated on the device: the flag gates the optimizer
                # kernels, so the host never waits for the loss and keeps queueing the next step
                # x * 0 == 0 holds exactly for finite x and fails for NaN / +-inf: isfinite in three tiny kernels instead of six
                ok = self._finite_flag(out)
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
        self._backward(loss)
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
            self._images_refresh()
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
        if self.hip_opt and hasattr(self.model, "_quantizers"):
            # rebuild the codebooks' fragment images right behind the update (optimizer, EMA, revival have all run): the next
            # step's assignment is then a single kernel launch
            for qz in self.model._quantizers():
                if qz._prepared is not None:
                    qz.prepared(qz._prepared[0][2], 1 << 20)
        self.step_idx += 1
        out["lr"] = lr_now
        return out
