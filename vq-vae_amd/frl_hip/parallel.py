"""Data-parallel gradient exchange: one process per GPU, RCCL (torch.distributed backend "nccl") over xGMI.

The reference has no distributed code; tiles are independent, so the only exchange per step is
an all-reduce(sum)/world of the gradients.  The payload is 1-5 MB, i.e. latency-bound on the 7 x 153 GB/s xGMI
links, so gradients are flattened into at most TWO contiguous float32 buckets (decoders + quantizer + phase path first,
type encoder second) and each bucket is reduced with ONE collective on a side stream as soon as its last gradient has
been produced, overlapping the remaining backward kernels.  Works unchanged with the gloo backend on CPU tensors
(used by the world_size=2 CPU tests).
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import ctypes
import struct

import torch
import torch.distributed as dist


class BucketedGradAllReduce:
    def __init__(self, named_params: Sequence, late_prefixes: Sequence[str] = ("encoder.", "spatial_conv."),
                 process_group=None, force_hooks: bool = False):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        params = [(n, p) for n, p in named_params if p.requires_grad]
        late = [(n, p) for n, p in params if n.startswith(tuple(late_prefixes))]
        early = [(n, p) for n, p in params if not n.startswith(tuple(late_prefixes))]
        self.buckets: List[List[torch.nn.Parameter]] = [[p for _, p in b] for b in (early, late) if b]
        self._flat: List[torch.Tensor] = []
        self._pending: Dict[int, int] = {}
        self._works = []
        self._index = {}
        for bi, bucket in enumerate(self.buckets):
            n = sum(p.numel() for p in bucket)
            dev = bucket[0].device
            # the LAST bucket carries one extra float: the "non-finite loss" flag of this rank rides along with the gradients,
            # so agreeing on the isfinite guard costs no collective of its own
            extra = 1 if bi == len(self.buckets) - 1 else 0
            self._flat.append(torch.zeros(n + extra, dtype=torch.float32, device=dev))
            for p in bucket:
                self._index[id(p)] = bi
        self.comm_stream = torch.cuda.Stream() if (self.buckets and self.buckets[0][0].is_cuda) else None
        self._hooks = []
        self.extra_streams = []                             # streams besides the hook's own that produce gradients of a bucket
        self.flag_src = None                                # float32 [1] tensor (1 = this rank saw a non-finite loss), set per step
        self._tables: Dict[int, tuple] = {}                 # bucket -> (gradient pointer key, device desc table, device chunk table)
        self.on_gpu = bool(self.buckets) and self.buckets[0][0].is_cuda
        self._train_stream = None
        self._collective_always = bool(force_hooks and dist.is_available() and dist.is_initialized())
        if self.world > 1 or force_hooks:
            for bucket in self.buckets:
                for p in bucket:
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self.reset()

    @property
    def active(self) -> bool:
        """True when gradients pass through the flat buckets (world > 1, or hooks forced for single-rank testing)."""
        return bool(self._hooks)

    def reset(self):
        # called from the training thread at step boundaries: remember its stream -- a bucket's last gradient hook may fire on the
        # model's side stream while other gradients of the same bucket were produced on this one
        self._train_stream = torch.cuda.current_stream() if self.on_gpu else None
        self._pending = {bi: len(b) for bi, b in enumerate(self.buckets)}
        self._works = []

    # called by autograd right after p.grad has been accumulated
    def _on_grad(self, p: torch.nn.Parameter):
        bi = self._index[id(p)]
        self._pending[bi] -= 1
        if self._pending[bi] == 0:
            self._launch(bi)

    def _launch(self, bi: int):
        bucket, flat = self.buckets[bi], self._flat[bi]
        if self.comm_stream is not None:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            for st in [self._train_stream] + list(self.extra_streams):   # training stream + e.g. the model's phase-branch stream
                if st is not None:
                    self.comm_stream.wait_stream(st)
            ctx = torch.cuda.stream(self.comm_stream)
        else:
            ctx = _Null()
        with ctx:
            if self.on_gpu:
                self._pack_gpu(bi)                         # ONE launch: flat = grads / world  (csrc/optim.hip)
            else:
                off = 0
                for p in bucket:
                    n = p.numel()
                    g = p.grad if p.grad is not None else torch.zeros_like(p)
                    flat[off:off + n].copy_(g.reshape(-1))
                    off += n
                if bi == len(self.buckets) - 1:
                    flat[off] = self.flag_src.reshape(()) if self.flag_src is not None else 0.0
                flat.div_(self.world)
            # (a single-rank group still issues the collective when one exists: the RCCL call is then part of what a captured step records)
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True) if (self.world > 1 or self._collective_always) else None
        self._works.append((bi, work))

    def _pack_gpu(self, bi: int):
        from . import _lib
        from .ops import check
        from .training.optim import ChunkTable
        bucket, flat = self.buckets[bi], self._flat[bi]
        grads = []
        for p in bucket:
            g = p.grad
            if g is not None and not (g.dtype == torch.float32 and g.is_contiguous()):
                g = g.float().contiguous()
            grads.append(g)
        self._keep = getattr(self, "_keep", {})
        self._keep[bi] = grads
        last = bi == len(self.buckets) - 1
        flag_ptr = self.flag_src.data_ptr() if (last and self.flag_src is not None) else 0
        key = tuple(g.data_ptr() if g is not None else 0 for g in grads) + (flag_ptr,)
        tab = self._tables.get(bi)
        if tab is None or tab[0] != key:
            off, recs, numels = 0, [], []
            for p, g in zip(bucket, grads):
                recs.append(struct.pack("<QQq", g.data_ptr() if g is not None else 0, flat.data_ptr() + 4 * off, p.numel()))
                numels.append(p.numel())
                off += p.numel()
            if last:                                        # flag slot (source NULL -> 0 when no flag was set)
                recs.append(struct.pack("<QQq", flag_ptr, flat.data_ptr() + 4 * off, 1))
                numels.append(1)
            raw = b"".join(recs)
            chunks = tab[2] if tab is not None else ChunkTable(numels, flat.device)
            tab = (key, ctypes.create_string_buffer(raw, len(raw)), chunks, len(numels))
            self._tables[bi] = tab
        check(_lib.load().frl_multi_tensor_scale_copy(ctypes.cast(tab[1], ctypes.c_void_p), tab[3], ctypes.c_void_p(tab[2].dev.data_ptr()),
                                                      ctypes.cast(tab[2].host_tensor_col, ctypes.c_void_p), tab[2].n, 1.0 / self.world,
                                                      ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "frl_multi_tensor_scale_copy")

    def flag_result(self) -> torch.Tensor:
        """Mean over ranks of the flags handed in through `flag_src` (float32 [1] view, valid after finish()): 0 <=> every rank was fine."""
        return self._flat[-1][-1:]

    def flat_grads(self) -> Dict[int, torch.Tensor]:
        """id(param) -> view of its averaged gradient inside the flat bucket (valid after finish(scatter=False))."""
        out = {}
        for bi, bucket in enumerate(self.buckets):
            off = 0
            for p in bucket:
                n = p.numel()
                out[id(p)] = self._flat[bi][off:off + n].view_as(p)
                off += n
        return out

    def finish(self, scatter: bool = True):
        """Waits for the collectives; with scatter=True the averaged gradients are copied back into p.grad, with scatter=False
        they stay in the flat buckets (flat_grads()) for an optimizer that reads them in place.  Call after backward()."""
        if self.world <= 1 and not self._hooks:
            return
        for bi, pend in self._pending.items():
            if pend != 0:          # parameters that received no gradient this step (unused branch)
                self._launch(bi)
        for bi, work in self._works:
            if work is not None:
                work.wait()
            if self.comm_stream is not None:
                torch.cuda.current_stream().wait_stream(self.comm_stream)
            if not scatter:
                continue
            off = 0
            for p in self.buckets[bi]:
                n = p.numel()
                if p.grad is None:
                    p.grad = torch.empty_like(p)
                p.grad.copy_(self._flat[bi][off:off + n].view_as(p))
                off += n
        self.reset()


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
