"""CodebookManager: usage tracking and dead-code revival for the VQ codebook.

The legacy trainer constructs `CodebookManager(num_codes=model.quant.codebook_size, code_dim=model.quant.emb_dim)` and hands it to
`model.attach_codebook_manager(manager)` (scripts/train_vqvae.py:92,196-198); the `vqvae.codebook_manager` module itself is not in
the reference tree, so the behaviour below is the build's definition ("dead-code reset"):

  * every training forward the model reports the per-code assignment counts of the batch (`update`); they are summed into a window
    on the device (no host synchronisation);
  * every `reset_every` optimizer steps (`after_step`, called by `VQVAETrainer`), codes used fewer than `min_count` times in the
    window are re-seeded with encoder outputs of the current batch -- `frl_vq_revive_dead_codes`, row = splitmix64(seed + k) mod N --
    and their AdamW moments are cleared; the window restarts.
  * data parallel: the window is all-reduced (SUM) so that every rank revives the same codes, and rank 0's new vectors are
    broadcast (the ranks hold different batches).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist

from .. import ops


class CodebookManager:
    def __init__(self, num_codes: int, code_dim: int, reset_every: int = 100, min_count: int = 1, seed: int = 0):
        if reset_every <= 0 or min_count <= 0:
            raise ValueError("reset_every and min_count must be positive")
        self.num_codes, self.code_dim = int(num_codes), int(code_dim)
        self.reset_every, self.min_count, self.seed = int(reset_every), int(min_count), int(seed)
        self.window: Optional[torch.Tensor] = None          # int64 [K] on the codebook's device
        self.revived: Optional[torch.Tensor] = None         # int32 [1] running total (device)
        self.steps = 0
        self._z: Optional[torch.Tensor] = None
        self._z_ok: Optional[torch.Tensor] = None            # device flag of the batch whose rows are kept (None: unguarded)

    def update(self, counts: Optional[torch.Tensor], z_rows: Optional[torch.Tensor] = None, ok: Optional[torch.Tensor] = None) -> None:
        """counts: int32 [K] assignments of this batch; z_rows: the encoder outputs [N, d] that were quantized (kept for revival);
        ok: optional device float [1] of the train step's isfinite guard -- a batch with ok <= 0 adds nothing to the window and its
        rows never seed a code (all on the device, no host synchronisation)."""
        if counts is None:
            return
        if counts.numel() != self.num_codes:
            raise ValueError(f"expected {self.num_codes} counts, got {counts.numel()}")
        if self.window is None:
            self.window = torch.zeros(self.num_codes, dtype=torch.int64, device=counts.device)
            self.revived = torch.zeros(1, dtype=torch.int32, device=counts.device)
        if ok is not None:
            counts = counts.to(torch.int64) * (ok.reshape(-1)[:1] > 0).to(torch.int64)
        self.window += counts
        if z_rows is not None:
            self._z, self._z_ok = z_rows.detach(), ok

    def usage(self) -> torch.Tensor:
        """Fraction of the window's assignments per code (device tensor)."""
        w = self.window.double()
        return w / w.sum().clamp_min(1.0)

    def after_step(self, quantizer, optimizer=None) -> bool:
        """Call once per optimizer step; returns True when a revival pass was launched."""
        self.steps += 1
        if self.window is None or self.steps % self.reset_every:
            return False
        cb = quantizer.codebook
        if self._z is None or self._z.shape[1] != cb.shape[1]:
            raise RuntimeError("CodebookManager.after_step: no encoder rows recorded (the model passes them to update())")
        distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        if distributed:
            dist.all_reduce(self.window)
        m = v = None
        if optimizer is not None and hasattr(optimizer, "exp_avg"):          # HipAdamW keeps flat lists
            for i, p in enumerate(optimizer.params):
                if p is cb:
                    m, v = optimizer.exp_avg[i], optimizer.exp_avg_sq[i]
        elif optimizer is not None and cb in getattr(optimizer, "state", {}):
            st = optimizer.state[cb]
            m, v = st.get("exp_avg"), st.get("exp_avg_sq")
        z = self._z if self._z.is_contiguous() else self._z.contiguous()
        window = self.window
        if self._z_ok is not None:                           # candidate rows of a skipped (non-finite) batch: revive nothing this time
            window = torch.where(self._z_ok.reshape(-1)[:1] > 0, window, torch.full_like(window, self.min_count))
        ops.vq_revive_dead_codes(cb.data, window, self.min_count, z, self.seed + self.steps, m, v, self.revived)
        torch.autograd.graph.increment_version(cb)
        if distributed:
            dist.broadcast(cb.data, src=0)
        if hasattr(quantizer, "ema_sum"):                                     # EMA quantizer: keep its running sums consistent
            dead = window < self.min_count
            quantizer.ema_sum[dead] = cb.data[dead]
            quantizer.ema_count[dead] = 1.0
        self.window.zero_()
        self._z = self._z_ok = None
        return True
