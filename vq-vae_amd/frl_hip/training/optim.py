"""Clip + AdamW for the HIP path: two launches per step over a device table of parameter tensors
(csrc/optim.hip: frl_adamw_clip_step) instead of the ~45 foreach / elementwise kernels of
torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW.  Same update rule and step order as the reference trainer
(frl/training/representation/step.py:1081-1087; AdamW groups of scripts/train_vqvae.py:221-228).

The object mimics the small part of the torch optimizer interface the trainer uses (param_groups[i]["lr"], zero_grad,
state_dict / load_state_dict with the torch key names "exp_avg" / "exp_avg_sq" / "step").
"""
from __future__ import annotations

import ctypes
import struct
from typing import Dict, List, Optional, Sequence

import torch

from .. import _lib
from ..ops import _stream, check, workspace

CHUNK = 4096


def chunk_table(numels: Sequence[int]) -> torch.Tensor:
    """int32 [nchunks, 2] = (tensor index, 4096-element window) -- the fixed work decomposition of the multi-tensor kernels."""
    rows = []
    for ti, n in enumerate(numels):
        rows.extend((ti, w) for w in range((n + CHUNK - 1) // CHUNK))
    return torch.tensor(rows, dtype=torch.int32).reshape(-1, 2)


class ChunkTable:
    """Device copy of the chunk table + the host copy of its tensor column (the C side splits launches by tensor range)."""

    def __init__(self, numels: Sequence[int], device):
        t = chunk_table(numels)
        self.n = int(t.shape[0])
        self.dev = t.to(device)
        self.host_tensor_col = (ctypes.c_int * self.n)(*t[:, 0].tolist())


class HipAdamW:
    def __init__(self, groups: List[dict], lr: float, betas=(0.9, 0.95), eps: float = 1e-8):
        self.param_groups = [dict(params=list(g["params"]), lr=lr, weight_decay=float(g.get("weight_decay", 0.0)), betas=tuple(betas), eps=eps)
                             for g in groups]
        self.params: List[torch.nn.Parameter] = [p for g in self.param_groups for p in g["params"]]
        for p in self.params:
            if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous():
                raise ValueError("HipAdamW needs contiguous float32 CUDA parameters")
        self.wd = [g["weight_decay"] for g in self.param_groups for _ in g["params"]]
        self.state: Dict[int, dict] = {}
        self.step_count = 0
        self.device = self.params[0].device
        self.exp_avg = [torch.zeros_like(p) for p in self.params]
        self.exp_avg_sq = [torch.zeros_like(p) for p in self.params]
        self.lag = [0] * len(self.params)                  # updates each tensor skipped (torch counts steps per parameter)
        self._key = None
        self._desc = None
        self._chunks = None
        self._live = None
        self._keep = None
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=self.device)
        self.lr_dev: Optional[torch.Tensor] = None         # device float[1]: when set, the kernels read the learning rate from it
        self.counters = torch.zeros(2, dtype=torch.int32, device=self.device)   # {updates applied, updates skipped}, device-side

    def zero_grad(self, set_to_none: bool = True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def _table(self, grads: List[Optional[torch.Tensor]]):
        """Device tables over the parameters that HAVE a gradient (torch.optim.AdamW skips the others entirely)."""
        key = tuple(g.data_ptr() if g is not None else 0 for g in grads) + tuple(self.lag)
        if key != self._key:                               # gradient buffers moved (or first step): rebuild the HOST record table
            live = [i for i, g in enumerate(grads) if g is not None]
            raw = b"".join(struct.pack("<QQQQqfi", self.params[i].data_ptr(), grads[i].data_ptr(), self.exp_avg[i].data_ptr(),
                                       self.exp_avg_sq[i].data_ptr(), self.params[i].numel(), self.wd[i], self.lag[i]) for i in live)
            self._desc = ctypes.create_string_buffer(raw, len(raw)) if live else None
            if tuple(live) != self._live:                  # the chunk table only depends on WHICH tensors take part
                self._chunks = ChunkTable([self.params[i].numel() for i in live], self.device) if live else None
                self._live = tuple(live)
            self._nlive = len(live)
            self._key = key
        return self._desc

    def step(self, max_norm: float = 0.0, grads: Optional[List[torch.Tensor]] = None, ok: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One update; `grads` (same order as the parameters) defaults to p.grad.  `ok` (device float[1]) makes the update
        conditional on the device: ok <= 0 skips it (isfinite guard without a host sync).  Returns the pre-clip global gradient
        norm (device tensor, no host sync)."""
        if grads is None:
            grads = [p.grad for p in self.params]
        grads = [g if (g is None or (g.dtype == torch.float32 and g.is_contiguous())) else g.float().contiguous() for g in grads]
        self._keep = grads                                 # keep converted copies alive until the kernels ran
        desc = self._table(grads)
        for i, g in enumerate(grads):
            if g is None:
                self.lag[i] += 1                           # takes effect from the next table build on
        if desc is None:
            self.step_count += 1
            self.grad_norm.zero_()
            return self.grad_norm
        lr = float(self.param_groups[0]["lr"])
        b1, b2 = self.param_groups[0]["betas"]
        self.step_count += 1
        lib = _lib.load()
        ws = workspace(lib.frl_adamw_workspace_bytes(), self.device)
        check(lib.frl_adamw_clip_step(ctypes.cast(desc, ctypes.c_void_p), self._nlive, ctypes.c_void_p(self._chunks.dev.data_ptr()),
                                      ctypes.cast(self._chunks.host_tensor_col, ctypes.c_void_p), self._chunks.n, float(max_norm), lr,
                                      float(b1), float(b2),
                                      float(self.param_groups[0]["eps"]), self.step_count, ctypes.c_void_p(self.grad_norm.data_ptr()),
                                      ctypes.c_void_p(ok.data_ptr()) if ok is not None else None, ctypes.c_void_p(self.counters.data_ptr()),
                                      ctypes.c_void_p(self.lr_dev.data_ptr()) if self.lr_dev is not None else None,
                                      ctypes.c_void_p(ws.data_ptr()), ws.numel(), _stream()), "frl_adamw_clip_step")
        torch.autograd.graph.increment_version(self.params)     # the kernel wrote through raw pointers: tell torch (version-keyed caches)
        return self.grad_norm

    @property
    def applied_and_skipped(self):
        """(updates applied, updates skipped) -- reads the device counters (host sync)."""
        a, b = self.counters.tolist()
        return int(a), int(b)

    # torch-compatible checkpoint payload (frl/training/representation/checkpointing.py stores optimizer.state_dict())
    def state_dict(self) -> dict:
        self.step_count = int(self.counters[0].item()) + 0   # the device counter is authoritative (skipped batches)
        st = {i: {"step": torch.tensor(float(self.step_count - self.lag[i])), "exp_avg": m, "exp_avg_sq": v}
              for i, (m, v) in enumerate(zip(self.exp_avg, self.exp_avg_sq))}
        off, groups = 0, []
        for g in self.param_groups:
            n = len(g["params"])
            groups.append({k: v for k, v in g.items() if k != "params"} | {"params": list(range(off, off + n))})
            off += n
        return {"state": st, "param_groups": groups}

    def load_state_dict(self, sd: dict):
        steps = {int(i): int(float(s["step"])) for i, s in sd["state"].items()}
        self.step_count = max(steps.values()) if steps else 0
        for i, s in sd["state"].items():
            self.exp_avg[int(i)].copy_(s["exp_avg"])
            self.exp_avg_sq[int(i)].copy_(s["exp_avg_sq"])
            self.lag[int(i)] = self.step_count - steps[int(i)]
        self.counters[0] = self.step_count
        for g, sg in zip(self.param_groups, sd["param_groups"]):
            for k in ("lr", "weight_decay", "betas", "eps"):
                if k in sg:
                    g[k] = sg[k]
        self._key = None
