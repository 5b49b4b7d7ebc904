"""Raw (non-autograd) Python bindings over the C ABI: torch tensors in, torch tensors out.

torch is used for device memory, the current stream and dtype bookkeeping only; every computation
below is a HIP kernel of libfrlhip.so.  All activation tensors are NHWC "rows": [..., C] contiguous.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import ACT_NONE, ACT_RELU, ACT_SIGMOID, BF16, F32, check  # noqa: F401

_WS = {}

# ----------------------------------------------------------------------------------------------
# optional per-op HIP-event timing (bench.py roofline): events are recorded on the stream the kernels are launched on
# ----------------------------------------------------------------------------------------------
_TIMING = {"on": False, "events": {}}


def set_timing(on: bool) -> None:
    _TIMING["on"] = bool(on)
    _TIMING["events"] = {}


def timing_summary():
    """name -> (calls, total_ms); synchronises the device."""
    torch.cuda.synchronize()
    return {k: (len(v), sum(a.elapsed_time(b) for a, b in v)) for k, v in _TIMING["events"].items()}


def kernel_timing(on: bool) -> None:
    """Library-side HIP-event pair around every kernel launch (see include/frl_hip.h: frl_kernel_timing_enable)."""
    _lib.load().frl_kernel_timing_enable(1 if on else 0)


def kernel_timing_report():
    """kernel expression -> (calls, total_ms) since the last report; synchronises the recorded events."""
    buf = ctypes.create_string_buffer(1 << 16)
    _lib.load().frl_kernel_timing_report(buf, len(buf))
    out = {}
    for line in buf.value.decode().splitlines():
        name, calls, ms = line.rsplit("\t", 2)
        out[name] = (int(calls), float(ms))
    return out


# ----------------------------------------------------------------------------------------------
# Index validation of the sparse ops (gathers, pair lists).  The reference's advanced indexing raises on an out-of-range index; the
# kernels take device pointers and would read out of bounds.  The Python wrappers therefore wrap negative indices, clamp the rest into
# range and OR "something was out of range" into a device flag without a host synchronisation; `index_errors()` reads it, and with
# FRL_HIP_CHECK_INDICES=1 the wrappers raise IndexError on the spot (one sync per call: debugging).
# ----------------------------------------------------------------------------------------------
_INDEX_FLAG = {}


def sanitize_indices(idx: torch.Tensor, n: int, what: str) -> torch.Tensor:
    """idx int64 (any shape) addressing n rows -> indices in [0, n): negatives wrapped as torch indexing does, out-of-range clamped and flagged."""
    import os
    bad = ((idx < -n) | (idx >= n)).any()
    flag = _INDEX_FLAG.get(idx.device)
    if flag is None:
        flag = _INDEX_FLAG[idx.device] = torch.zeros(1, dtype=torch.int32, device=idx.device)
    flag |= bad.to(torch.int32)
    if os.environ.get("FRL_HIP_CHECK_INDICES", "0") == "1" and bool(bad.item()):
        raise IndexError(f"{what}: index out of range for {n} rows")
    return torch.where(idx < 0, idx + n, idx).clamp_(0, max(n - 1, 0))


def index_errors(device=None, reset: bool = True) -> bool:
    """True when any sparse op since the last call saw an out-of-range index on `device` (all devices when None); synchronises."""
    hit = False
    for dev, flag in _INDEX_FLAG.items():
        if device is None or torch.device(device) == dev:
            hit |= bool(flag.item())
            if reset:
                flag.zero_()
    return hit


class PackCache:
    """Weight-image cache of a trainer (include/frl_hip.h: frl_pack_cache_*): the packed MFMA fragment images of the model's weights live
    in an arena owned by this object; `with cache:` makes the conv-like calls use them, `refresh()` rewrites all of them in one launch."""

    def __init__(self, device, nbytes: int = 64 << 20):
        lib = _lib.load()
        self.arena = torch.empty(int(nbytes) + lib.frl_pack_cache_table_bytes(), dtype=torch.uint8, device=device)
        self.handle = lib.frl_pack_cache_create(ctypes.c_void_p(self.arena.data_ptr()), self.arena.numel())
        if self.handle <= 0:
            check(self.handle, "frl_pack_cache_create")
        self._prev = 0

    def __enter__(self):
        self._prev = _lib.load().frl_pack_cache_activate(self.handle)
        return self

    def __exit__(self, *a):
        _lib.load().frl_pack_cache_activate(self._prev)
        return False

    def refresh(self) -> None:
        check(_lib.load().frl_pack_cache_refresh(self.handle, _stream()), "frl_pack_cache_refresh")

    @property
    def images(self) -> int:
        return _lib.load().frl_pack_cache_images(self.handle)

    def close(self) -> None:
        if self.handle > 0:
            _lib.load().frl_pack_cache_destroy(self.handle)
            self.handle = 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class span:
    """with ops.span("name"): ...  -- HIP-event timing of a sub-range (used for single-kernel roofline numbers)."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if _TIMING["on"]:
            self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *a):
        if _TIMING["on"]:
            self.e1.record()
            _TIMING["events"].setdefault(self.name, []).append((self.e0, self.e1))
        return False


def _timed(name):
    def deco(fn):
        def wrapper(*a, **kw):
            if not _TIMING["on"]:
                return fn(*a, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **kw)
            e1.record()
            _TIMING["events"].setdefault(name, []).append((e0, e1))
            return out
        wrapper.__name__ = fn.__name__
        wrapper.__doc__ = fn.__doc__
        return wrapper
    return deco


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"frl_hip supports float32 and bfloat16 activations, got {t.dtype}")


def _p(t: Optional[torch.Tensor]):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk_rows(t: torch.Tensor, c: int, name: str):
    if not t.is_cuda:
        raise _lib.FrlHipError(f"{name}: tensor must live on the GPU (no CPU fallback)")
    if not t.is_contiguous() or t.shape[-1] != c:
        raise ValueError(f"{name}: expected contiguous [..., {c}] rows, got {tuple(t.shape)} stride {t.stride()}")


def _chk_like(t: torch.Tensor, ref: torch.Tensor, name: str):
    """An epilogue operand read at the addresses the kernel writes: same shape, dtype and device, contiguous, not the output itself."""
    if t.shape != ref.shape or t.dtype != ref.dtype or t.device != ref.device or not t.is_contiguous():
        raise ValueError(f"{name}: expected a contiguous {tuple(ref.shape)} {ref.dtype} tensor on {ref.device}, got "
                         f"{tuple(t.shape)} {t.dtype} stride {t.stride()}")


def _f32(t: Optional[torch.Tensor], name: str):
    if t is None:
        return None
    if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
        raise ValueError(f"{name}: parameters are passed as contiguous float32 CUDA tensors")
    return t


_DEFER = {"on": False, "keep": []}


class deferred_reductions:
    """`with deferred_reductions(params): loss.backward()` -- the slab reductions behind the weight-gradient kernels of the backward pass
    (csrc/defer.hip, include/frl_hip.h) are parked and run in ONE launch when the block ends, on the current stream.  While it is open every
    kernel call gets a workspace of its own (the parked slabs live in them), kept until the flush.  Before the flush the block checks that
    every parked destination lies inside the .grad of one of `params`: autograd must have adopted the tensors the kernels' wrappers
    returned, not copies of them (a parameter used twice, or a hook that clones gradients, would read a tensor nothing has written yet)."""

    def __init__(self, params):
        self.params = list(params)

    def __enter__(self):
        check(_lib.load().frl_defer_begin(), "frl_defer_begin")
        _DEFER["keep"] = []
        _DEFER["on"] = True
        return self

    def __exit__(self, et, ev, tb):
        lib = _lib.load()
        _DEFER["on"] = False
        keep, _DEFER["keep"] = _DEFER["keep"], []
        if et is not None:
            lib.frl_defer_abort()
            return False
        try:
            buf = (ctypes.c_void_p * 512)()
            n = lib.frl_defer_destinations(ctypes.cast(buf, ctypes.c_void_p), 512)
            spans = sorted((g.data_ptr(), g.data_ptr() + g.numel() * g.element_size())
                           for g in (p.grad for p in self.params) if g is not None)
            starts = [a for a, _ in spans]
            import bisect
            for i in range(n):
                d = int(buf[i])
                k = bisect.bisect_right(starts, d) - 1
                if k < 0 or not (spans[k][0] <= d < spans[k][1]):
                    raise RuntimeError("deferred_reductions: a parked gradient is not the .grad of any parameter (autograd copied or summed "
                                       "it: a parameter used twice, or a gradient hook); run this backward without deferral")
        except Exception:
            lib.frl_defer_abort()
            raise
        rc = lib.frl_defer_flush(_stream())
        if rc < 0:
            check(rc, "frl_defer_flush")
        cur = torch.cuda.current_stream()
        for t in keep:                                          # slabs written on a side stream, read by the flush on this one
            t.record_stream(cur)
        return False


def workspace(nbytes: int, device) -> torch.Tensor:
    """Grow-only per-(device, stream) scratch buffer handed to kernels that need one (inside `deferred_reductions`: a buffer per call)."""
    if _DEFER["on"]:
        ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device)
        _DEFER["keep"].append(ws)
        return ws
    key = (str(device), torch.cuda.current_stream().cuda_stream)
    ws = _WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _WS[key] = ws
    return ws


# ----------------------------------------------------------------------------------------------
# pointwise convolution (reference: nn.Conv2d(.,.,1) call sites, see csrc/pw_conv.hip)
# ----------------------------------------------------------------------------------------------
@_timed("conv1x1_fwd")
def conv1x1_fwd(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], act: int = ACT_NONE) -> torch.Tensor:
    cout, cin = w.shape[0], w.shape[1]
    _chk_rows(x, cin, "conv1x1_fwd.x")
    w = _f32(w.reshape(cout, cin), "w")
    y = torch.empty(x.shape[:-1] + (cout,), dtype=x.dtype, device=x.device)
    p = x.numel() // cin
    lib = _lib.load()
    ws = workspace(lib.frl_conv_workspace_bytes(cin, cout, 1), x.device)
    check(lib.frl_conv1x1_fwd(_p(x), _p(w), _p(_f32(bias, "bias")), _p(y), p, cin, cout, act, _dt(x), _p(ws), ws.numel(),
                              _stream()), "frl_conv1x1_fwd")
    return y


@_timed("conv1x1_bwd_data")
def conv1x1_bwd_data(dy: torch.Tensor, w: torch.Tensor, y: Optional[torch.Tensor] = None, act: int = ACT_NONE,
                     add: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dx = (dy .* act'(y)) W (+ add: a second gradient stream of x, accumulated in the kernel's epilogue)."""
    cout, cin = w.shape[0], w.shape[1]
    _chk_rows(dy, cout, "conv1x1_bwd_data.dy")
    w = _f32(w.reshape(cout, cin), "w")
    dx = torch.empty(dy.shape[:-1] + (cin,), dtype=dy.dtype, device=dy.device)
    p = dy.numel() // cout
    lib = _lib.load()
    ws = workspace(lib.frl_conv_workspace_bytes(cin, cout, 1), dy.device)
    if add is not None:
        _chk_like(add, dx, "conv1x1_bwd_data.add")
        check(lib.frl_conv1x1_bwd_data_add(_p(dy), _p(y), act, _p(w), _p(dx), _p(add), p, cin, cout, _dt(dy), _p(ws), ws.numel(),
                                           _stream()), "frl_conv1x1_bwd_data_add")
        return dx
    check(lib.frl_conv1x1_bwd_data(_p(dy), _p(y), act, _p(w), _p(dx), p, cin, cout, _dt(dy), _p(ws), ws.numel(), _stream()),
          "frl_conv1x1_bwd_data")
    return dx


def _conv1x1_bwd_weight_impl(dy: torch.Tensor, x: torch.Tensor, y: Optional[torch.Tensor] = None, act: int = ACT_NONE,
                       want_bias: bool = True, scalar_frags: bool = False) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    cout, cin = dy.shape[-1], x.shape[-1]
    _chk_rows(dy, cout, "bwd_weight.dy")
    _chk_rows(x, cin, "bwd_weight.x")
    p = dy.numel() // cout
    lib = _lib.load()
    nbytes = lib.frl_conv1x1_bwd_weight_workspace_bytes(p, cin, cout)
    ws = workspace(nbytes, dy.device)
    dw = torch.empty(cout, cin, dtype=torch.float32, device=dy.device)
    db = torch.empty(cout, dtype=torch.float32, device=dy.device) if want_bias else None
    check(lib.frl_conv_tap_bwd_weight(_p(dy), _p(y), act, _p(x), _p(dw), cin, 1, _p(db), p, cin, cout, 1, 1, 0,
                                      _dt(dy), _p(ws), ws.numel(), 1 if scalar_frags else 0, _stream()),
          "frl_conv_tap_bwd_weight")
    return dw, db


conv1x1_bwd_weight = _timed("conv1x1_bwd_weight")(_conv1x1_bwd_weight_impl)


# ----------------------------------------------------------------------------------------------
# vector quantizer (csrc/vq.hip)
# ----------------------------------------------------------------------------------------------
def vq_stream_tiles(nt: int) -> int:
    """A/B hook (include/frl_hip.h: frl_vq_stream_tiles): tiles per wave and batch of the streaming assignment kernel, 0 = the resident
    kernel, -1 = default.  Returns the previous setting."""
    return int(_lib.load().frl_vq_stream_tiles(int(nt)))


@_timed("vq_prepare")
def vq_prepare(codebook: torch.Tensor, n_rows: int, dtype: torch.dtype, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Prepared image of the codebook (norms + packed MFMA fragments) for `vq_assign(..., prep=)`; valid until the codebook changes."""
    k, d = codebook.shape
    cb = _f32(codebook, "codebook")
    lib = _lib.load()
    nbytes = lib.frl_vq_prepared_bytes(k, d)
    if out is None or out.numel() < nbytes or out.device != cb.device:
        out = torch.empty(nbytes, dtype=torch.uint8, device=cb.device)
    check(lib.frl_vq_prepare(_p(cb), int(n_rows), k, d, BF16 if dtype == torch.bfloat16 else F32, _p(out), out.numel(), _stream()),
          "frl_vq_prepare")
    return out


@_timed("vq_assign")
def vq_assign(z: torch.Tensor, codebook: torch.Tensor, prep: Optional[torch.Tensor] = None):
    """z [N,d] rows, codebook [K,d] f32 -> (idx int32 [N], z_q [N,d], stats f32 [4], counts int32 [K]).
    prep: image from `vq_prepare` for this codebook content and z.dtype (None: built inside the call)."""
    k, d = codebook.shape
    _chk_rows(z, d, "vq_assign.z")
    cb = _f32(codebook, "codebook")
    n = z.numel() // d
    lib = _lib.load()
    ws = workspace(lib.frl_vq_workspace_bytes(n, k, d), z.device)
    idx = torch.empty(n, dtype=torch.int32, device=z.device)
    zq = torch.empty_like(z)
    stats = torch.empty(4, dtype=torch.float32, device=z.device)
    counts = torch.empty(k, dtype=torch.int32, device=z.device)
    if prep is not None and (prep.dtype != torch.uint8 or prep.device != z.device or prep.numel() < lib.frl_vq_prepared_bytes(k, d)):
        raise ValueError("vq_assign: prep is not a prepared-codebook image for this codebook")
    check(lib.frl_vq_assign_fwd_prepared(_p(z), _p(cb), _p(prep), n, k, d, _p(idx), _p(zq), _p(stats), _p(counts), _dt(z), _p(ws),
                                         ws.numel(), _stream()), "frl_vq_assign_fwd")
    return idx, zq, stats, counts


@_timed("vq_bwd")
def vq_bwd(g_out: Optional[torch.Tensor], z: torch.Tensor, codebook: torch.Tensor, idx: torch.Tensor,
           counts: torch.Tensor, gscale: Optional[torch.Tensor], beta: float, want_gz: bool = True,
           want_ge: bool = True, want_sums: bool = False, zq: Optional[torch.Tensor] = None):
    k, d = codebook.shape
    n = z.numel() // d
    lib = _lib.load()
    ws = workspace(lib.frl_vq_workspace_bytes(n, k, d), z.device)
    gz = torch.empty_like(z) if want_gz else None
    ge = torch.empty(k, d, dtype=torch.float32, device=z.device) if want_ge else None
    sums = torch.empty(k, d, dtype=torch.float32, device=z.device) if want_sums else None
    cb32 = _f32(codebook, "codebook")
    if _DEFER["on"]:                                            # the parked codebook-gradient reduction reads these when the flush runs
        _DEFER["keep"] += [t for t in (counts, gscale, cb32) if t is not None]
    check(lib.frl_vq_bwd(_p(g_out), _p(z), _p(zq), _p(cb32), _p(idx), _p(counts), _p(gscale), float(beta),
                         n, k, d, _p(gz), _p(ge), _p(sums), _dt(z), _p(ws), ws.numel(), _stream()), "frl_vq_bwd")
    return gz, ge, sums


def vq_ema_update(sums, counts, ema_count, ema_sum, codebook, decay: float, eps: float, ok: Optional[torch.Tensor] = None):
    """EMA codebook update in place; `ok` (device float [1]) <= 0 skips it on the device (isfinite guard, no host sync)."""
    k, d = codebook.shape
    if ok is not None and (ok.dtype != torch.float32 or not ok.is_cuda or ok.numel() < 1):
        raise ValueError("vq_ema_update: ok must be a float32 CUDA tensor")
    check(_lib.load().frl_vq_ema_update(_p(sums), _p(counts), k, d, float(decay), float(eps), _p(ema_count),
                                        _p(ema_sum), _p(codebook), _p(ok), _stream()), "frl_vq_ema_update")


# ----------------------------------------------------------------------------------------------
# GroupNorm over NHWC rows (csrc/norm.hip)
# ----------------------------------------------------------------------------------------------
def _mult_ptrs(mults, n):
    if mults is None or all(m is None for m in mults):
        return None
    for m in mults:
        if m is not None and (m.dtype != torch.float32 or m.numel() != 1 or not m.is_cuda):
            raise ValueError("scalar_combine: multipliers are float32 CUDA scalars")
    return (ctypes.c_void_p * n)(*[0 if m is None else m.data_ptr() for m in mults])


def scalar_combine(terms, coefs, mults=None, aux_coefs=None):
    """[device float scalars], [host floats](, [device float scalars | None]) -> (value [] f32 = sum coef_i * mult_i * term_i, ok [1] f32 = 1.0 if
    the value is finite else 0.0[, aux [] f32 = sum aux_coef_i * term_i]).  A multiplier is read on the device at run time (a scheduled loss
    weight inside a captured graph); aux_coefs adds a second combination of the same terms to the launch (a reported sub-total)."""
    n = len(terms)
    dev = terms[0].device
    for t in terms:
        if t.dtype != torch.float32 or t.numel() != 1 or not t.is_cuda:
            raise ValueError("scalar_combine: terms are float32 CUDA scalars")
    ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in terms])
    cf = (ctypes.c_float * n)(*[float(c) for c in coefs])
    out = torch.empty((), dtype=torch.float32, device=dev)
    ok = torch.empty(1, dtype=torch.float32, device=dev)
    if aux_coefs is None:
        check(_lib.load().frl_scalar_combine_dev(ptrs, cf, _mult_ptrs(mults, n), n, _p(out), _p(ok), _stream()), "frl_scalar_combine")
        return out, ok
    ac = (ctypes.c_float * n)(*[float(c) for c in aux_coefs])
    aux = torch.empty((), dtype=torch.float32, device=dev)
    check(_lib.load().frl_scalar_combine_aux(ptrs, cf, _mult_ptrs(mults, n), ac, n, _p(out), _p(ok), _p(aux), _stream()), "frl_scalar_combine_aux")
    return out, ok, aux


def scalar_fanout(g, coefs, mults=None):
    """g device float scalar -> out [n] f32 with out[i] = g * coef_i (* mult_i)."""
    n = len(coefs)
    cf = (ctypes.c_float * n)(*[float(c) for c in coefs])
    out = torch.empty(n, dtype=torch.float32, device=g.device)
    check(_lib.load().frl_scalar_fanout_dev(_p(g), cf, _mult_ptrs(mults, n), n, _p(out), _stream()), "frl_scalar_fanout")
    return out


def encoder2_supported(c0: int, c1: int, c2: int, g1: int, g2: int, hw: int, dtype) -> bool:
    return bool(_lib.load().frl_encoder2_supported(c0, c1, c2, g1, g2, hw, 1 if dtype == torch.bfloat16 else 0))


@_timed("encoder2_fwd")
def encoder2_fwd(x, w1, g1, b1, w2, g2, b2, eps: float = 1e-5):
    """Fused conv1x1 -> GroupNorm -> ReLU -> conv1x1 -> GroupNorm (csrc/enc_fused.hip).  x [B, ..., 64] bf16 -> (z, stats [B, 32])."""
    b, c = x.shape[0], x.shape[-1]
    hw = x.numel() // (b * c)
    _chk_rows(x, c, "encoder2.x")
    lib = _lib.load()
    z = torch.empty(x.shape[:-1] + (w2.shape[0],), dtype=x.dtype, device=x.device)
    stats = torch.empty(b, 32, dtype=torch.float32, device=x.device)
    ws = workspace(lib.frl_encoder2_workspace_bytes(b), x.device)
    check(lib.frl_encoder2_fwd(_p(x), _p(_f32(w1, "w1")), _p(_f32(g1, "g1")), _p(_f32(b1, "b1")), _p(_f32(w2, "w2")), _p(_f32(g2, "g2")),
                               _p(_f32(b2, "b2")), _p(z), _p(stats), b, hw, float(eps), _p(ws), ws.numel(), _stream()), "frl_encoder2_fwd")
    return z, stats


@_timed("encoder2_bwd")
def encoder2_bwd(x, dz, w1, g1, b1, w2, g2, b2, stats):
    """-> (dw1, dg1, db1, dw2, dg2, db2): parameter gradients of the fused encoder (its input is data: no dx)."""
    b, c = x.shape[0], x.shape[-1]
    hw = x.numel() // (b * c)
    lib = _lib.load()
    dw1 = torch.empty(w1.shape, dtype=torch.float32, device=x.device)
    dw2 = torch.empty(w2.shape, dtype=torch.float32, device=x.device)
    dg1, db1 = torch.empty_like(g1), torch.empty_like(b1)
    dg2, db2 = torch.empty_like(g2), torch.empty_like(b2)
    ws = workspace(lib.frl_encoder2_workspace_bytes(b), x.device)
    check(lib.frl_encoder2_bwd(_p(x), _p(dz), _p(_f32(w1, "w1")), _p(g1), _p(b1), _p(_f32(w2, "w2")), _p(g2), _p(b2), _p(stats), _p(dw1), _p(dg1),
                               _p(db1), _p(dw2), _p(dg2), _p(db2), b, hw, _p(ws), ws.numel(), _stream()), "frl_encoder2_bwd")
    return dw1, dg1, db1, dw2, dg2, db2


@_timed("groupnorm_fwd")
def groupnorm_fwd(x: torch.Tensor, gamma, beta, groups: int, eps: float = 1e-5, relu: bool = False):
    """x [B, ..., C] -> (y, mean [B,G], rstd [B,G])."""
    b, c = x.shape[0], x.shape[-1]
    hw = x.numel() // (b * c)
    _chk_rows(x, c, "groupnorm.x")
    y = torch.empty_like(x)
    mean = torch.empty(b, groups, dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    check(_lib.load().frl_groupnorm_fwd(_p(x), _p(_f32(gamma, "gamma")), _p(_f32(beta, "beta")), _p(y), _p(mean), _p(rstd),
                                        b, hw, c, groups, float(eps), int(relu), _dt(x), _stream()), "frl_groupnorm_fwd")
    return y, mean, rstd


@_timed("groupnorm_bwd")
def groupnorm_bwd(dy, x, gamma, beta, mean, rstd, groups: int, relu: bool = False):
    b, c = x.shape[0], x.shape[-1]
    hw = x.numel() // (b * c)
    lib = _lib.load()
    ws = workspace(lib.frl_groupnorm_bwd_workspace_bytes(b, c, groups), x.device)
    dx = torch.empty_like(x)
    dg = torch.empty(c, dtype=torch.float32, device=x.device)
    db = torch.empty(c, dtype=torch.float32, device=x.device)
    check(lib.frl_groupnorm_bwd(_p(dy), _p(x), _p(gamma), _p(beta), _p(mean), _p(rstd), _p(dx), _p(dg), _p(db), b, hw, c,
                                groups, int(relu), _dt(x), _p(ws), ws.numel(), _stream()), "frl_groupnorm_bwd")
    return dx, dg, db


# ----------------------------------------------------------------------------------------------
# streaming elementwise ops (csrc/elementwise.hip)
# ----------------------------------------------------------------------------------------------
@_timed("mse_fwd")
def mse_fwd(pred: torch.Tensor, target: torch.Tensor, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Returns stats f32 [2] = {mean squared error over valid elements, number of valid elements}."""
    c = pred.shape[-1]
    _chk_rows(pred, c, "mse.pred")
    _chk_rows(target, c, "mse.target")
    lib = _lib.load()
    ws = workspace(lib.frl_mse_workspace_bytes(), pred.device)
    out = torch.empty(2, dtype=torch.float32, device=pred.device)
    check(lib.frl_mse_fwd(_p(pred), _p(target), _p(mask), pred.numel() // c, c, _p(out), _dt(pred), _p(ws), ws.numel(),
                          _stream()), "frl_mse_fwd")
    return out


@_timed("mse_bwd")
def mse_bwd(pred, target, mask, gscale: Optional[torch.Tensor], stats) -> torch.Tensor:
    c = pred.shape[-1]
    d = torch.empty_like(pred)
    check(_lib.load().frl_mse_bwd(_p(pred), _p(target), _p(mask), _p(gscale), _p(stats), pred.numel() // c, c, _p(d),
                                  _dt(pred), _stream()), "frl_mse_bwd")
    return d


@_timed("film_modulate_fwd")
def film_modulate_fwd(h: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor) -> torch.Tensor:
    """h [B,T,HW..,C], gamma/beta [B,HW..,C]."""
    b, t, c = h.shape[0], h.shape[1], h.shape[-1]
    hw = h.numel() // (b * t * c)
    out = torch.empty_like(h)
    check(_lib.load().frl_film_modulate_fwd(_p(h), _p(gamma), _p(beta), _p(out), b, t, hw, c, _dt(h), _stream()),
          "frl_film_modulate_fwd")
    return out


@_timed("film_modulate_bwd")
def film_modulate_bwd(dout, h, gamma):
    b, t, c = h.shape[0], h.shape[1], h.shape[-1]
    hw = h.numel() // (b * t * c)
    dh = torch.empty_like(h)
    dg = torch.empty_like(gamma)
    db = torch.empty_like(gamma)
    check(_lib.load().frl_film_modulate_bwd(_p(dout), _p(h), _p(gamma), _p(dh), _p(dg), _p(db), b, t, hw, c, _dt(h),
                                            _stream()), "frl_film_modulate_bwd")
    return dh, dg, db


def film_fused_supported(z_type: torch.Tensor, h: torch.Tensor, hidden: int) -> bool:
    """True when the FiLM nets and the modulation run as one launch per direction (csrc/film_fused.hip): bf16, 64 -> 32 -> 12."""
    return bool(z_type.is_cuda and h.dim() >= 3 and fusion_enabled("film") and _lib.load().frl_film_fused_supported(z_type.shape[-1], hidden, h.shape[-1], _dt(h))
                and z_type.dtype == h.dtype)


def _film_params(params):
    """(w1g, b1g, w2g, b2g, w1b, b1b, w2b, b2b) float32 parameters of FiLMLayer.gamma_network / beta_network"""
    if len(params) != 8:
        raise ValueError("film_fused: expected (w1g, b1g, w2g, b2g, w1b, b1b, w2b, b2b)")
    return [_p(_f32(t, "film parameter")) for t in params]


@_timed("film_fused_fwd")
def film_fused_fwd(z_type: torch.Tensor, h: torch.Tensor, params):
    """z_type [B,HW..,64], h [B,T,HW..,12] -> (z = gamma * h + beta, gamma [B,HW..,12], beta)."""
    b, t, c = h.shape[0], h.shape[1], h.shape[-1]
    hw = h.numel() // (b * t * c)
    _chk_rows(z_type, z_type.shape[-1], "film_fused_fwd.z_type")
    _chk_rows(h, c, "film_fused_fwd.h")
    if z_type.numel() // z_type.shape[-1] != b * hw:
        raise ValueError("film_fused_fwd: z_type must hold one row per (sample, pixel) of h")
    lib = _lib.load()
    ws = workspace(lib.frl_film_fused_workspace_bytes(), h.device)
    z = torch.empty_like(h)
    gamma = torch.empty(tuple(z_type.shape[:-1]) + (c,), dtype=h.dtype, device=h.device)
    beta = torch.empty_like(gamma)
    check(lib.frl_film_fused_fwd(_p(z_type), _p(h), *_film_params(params), _p(z), _p(gamma), _p(beta), b, t, hw, _p(ws), ws.numel(), _stream()),
          "frl_film_fused_fwd")
    return z, gamma, beta


@_timed("film_fused_bwd")
def film_fused_bwd(z_type: torch.Tensor, h: torch.Tensor, dz: torch.Tensor, params):
    """-> (dh, [dw1g, db1g, dw2g, db2g, dw1b, db1b, dw2b, db2b])"""
    b, t, c = h.shape[0], h.shape[1], h.shape[-1]
    hw = h.numel() // (b * t * c)
    _chk_like(dz, h, "film_fused_bwd.dz")
    lib = _lib.load()
    ws = workspace(lib.frl_film_fused_workspace_bytes(), h.device)
    dh = torch.empty_like(h)
    grads = [torch.empty(p.shape, dtype=torch.float32, device=h.device) for p in params]
    check(lib.frl_film_fused_bwd(_p(z_type), _p(h), _p(dz), *_film_params(params), _p(dh), *[_p(g) for g in grads], b, t, hw, _p(ws), ws.numel(),
                                 _stream()), "frl_film_fused_bwd")
    return dh, grads


@_timed("gate_blend_fwd")
def gate_blend_fwd(smoothed, residual, gate_raw, min_gate: float):
    out = torch.empty_like(smoothed)
    gate = torch.empty_like(smoothed)
    check(_lib.load().frl_gate_blend_fwd(_p(smoothed), _p(residual), _p(gate_raw), float(min_gate), _p(out), _p(gate),
                                         smoothed.numel(), _dt(smoothed), _stream()), "frl_gate_blend_fwd")
    return out, gate


@_timed("gate_blend_bwd")
def gate_blend_bwd(dout, dgate_ext, residual, gate_raw, min_gate: float, sigmoid_mask: bool = False):
    """-> (d residual, d gate_raw); sigmoid_mask: d gate_raw comes back multiplied by gate_raw (1 - gate_raw) (gradient w.r.t. the pre-activation)."""
    dres = torch.empty_like(dout)
    dgraw = torch.empty_like(dout)
    check(_lib.load().frl_gate_blend_bwd_masked(_p(dout), _p(dgate_ext), _p(residual), _p(gate_raw), float(min_gate), _p(dres),
                                                _p(dgraw), dout.numel(), _dt(dout), int(sigmoid_mask), _stream()), "frl_gate_blend_bwd_masked")
    return dres, dgraw


@_timed("mean_time")
def mean_time(tile: torch.Tensor) -> torch.Tensor:
    """tile [B,T,H,W,C] -> [B,H,W,C]."""
    b, t = tile.shape[0], tile.shape[1]
    out = torch.empty((b,) + tuple(tile.shape[2:]), dtype=tile.dtype, device=tile.device)
    check(_lib.load().frl_mean_time_fwd(_p(tile), _p(out), b, t, out.numel() // b, _dt(tile), _stream()), "frl_mean_time_fwd")
    return out


@_timed("vq_revive_dead_codes")
def vq_revive_dead_codes(codebook: torch.Tensor, window_counts: torch.Tensor, min_count: int, z_rows: torch.Tensor, seed: int,
                         exp_avg: Optional[torch.Tensor] = None, exp_avg_sq: Optional[torch.Tensor] = None,
                         revived: Optional[torch.Tensor] = None) -> torch.Tensor:
    """In place: codebook[k] = z_rows[splitmix64(seed + k) % N] for every code with window_counts[k] < min_count; clears the AdamW
    moment rows of those codes; returns the device int32 counter `revived` (incremented).  No host synchronisation."""
    k, d = codebook.shape
    if codebook.dtype != torch.float32 or not codebook.is_contiguous():
        raise ValueError("vq_revive_dead_codes: codebook must be contiguous float32 [K, d]")
    if window_counts.dtype != torch.int64 or window_counts.numel() != k or window_counts.device != codebook.device:
        raise ValueError("vq_revive_dead_codes: window_counts must be int64 [K] on the codebook's device")
    if z_rows.dim() != 2 or z_rows.shape[1] != d or not z_rows.is_contiguous() or z_rows.device != codebook.device:
        raise ValueError("vq_revive_dead_codes: z_rows must be contiguous [N, d] on the codebook's device")
    for mom in (exp_avg, exp_avg_sq):
        if mom is not None and (mom.dtype != torch.float32 or mom.shape != codebook.shape or not mom.is_contiguous()):
            raise ValueError("vq_revive_dead_codes: AdamW moments must match the codebook")
    if revived is None:
        revived = torch.zeros(1, dtype=torch.int32, device=codebook.device)
    check(_lib.load().frl_vq_revive_dead_codes(_p(codebook), _p(window_counts), int(min_count), _p(z_rows), z_rows.shape[0], k, d,
                                               int(seed) & 0xFFFFFFFFFFFFFFFF, _p(exp_avg), _p(exp_avg_sq), _p(revived), _dt(z_rows),
                                               _stream()), "frl_vq_revive_dead_codes")
    return revived


@_timed("mutual_knn")
def mutual_knn(features: torch.Tensor, patch_id: torch.Tensor, coords: torch.Tensor, k: int, pos_min_spatial: float):
    """features [N, D] float32, patch_id [N] int32, coords [N, 2] float32 -> (knn_idx [N, k] int32, mutual [N, k] uint8)."""
    n, d = features.shape
    if features.dtype != torch.float32 or not features.is_contiguous():
        raise ValueError("mutual_knn: features must be contiguous float32 [N, D]")
    if patch_id.dtype != torch.int32 or patch_id.numel() != n or coords.dtype != torch.float32 or tuple(coords.shape) != (n, 2) \
            or not coords.is_contiguous() or not patch_id.is_contiguous():
        raise ValueError("mutual_knn: patch_id must be int32 [N], coords float32 [N, 2]")
    if d % 16 or d > 256:
        if d > 256:
            raise ValueError("mutual_knn: at most 256 feature channels")
        # zero columns add exactly nothing to a squared distance (and leave the summation order of the real ones untouched)
        features = torch.nn.functional.pad(features, (0, (-d) % 16)).contiguous()
        d = features.shape[1]
    knn = torch.empty((n, k), dtype=torch.int32, device=features.device)
    mutual = torch.empty((n, k), dtype=torch.uint8, device=features.device)
    check(_lib.load().frl_mutual_knn(_p(features), n, d, _p(patch_id), _p(coords), float(pos_min_spatial), int(k), _p(knn), _p(mutual),
                                     _stream()), "frl_mutual_knn")
    return knn, mutual


@_timed("normalize_tiles")
def normalize_tiles(raw: torch.Tensor, table: torch.Tensor, valid: Optional[torch.Tensor] = None, out_dtype: torch.dtype = torch.bfloat16,
                    out: Optional[torch.Tensor] = None, mask_out: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Chunk-store rows [..., F] (float16 | float32, NaN = no data) -> normalised rows (bf16 | f32) + validity bytes [...].

    `table` is the device image of the per-feature records (`data.normalization.norm_table`, 8 x 4 bytes per feature);
    `valid` an optional uint8/bool tensor over the leading dims.  See include/frl_hip.h (frl_normalize_tiles)."""
    if raw.dtype not in (torch.float16, torch.float32) or not raw.is_contiguous():
        raise ValueError("normalize_tiles: raw rows must be contiguous float16 or float32")
    if out_dtype not in (torch.bfloat16, torch.float32):
        raise ValueError("normalize_tiles: output must be bfloat16 or float32")
    f = raw.shape[-1]
    rows = raw.numel() // f if f else 0
    if table.dtype != torch.uint8 or table.numel() != 32 * f or table.device != raw.device:
        raise ValueError("normalize_tiles: table must be the 32-byte-per-feature record image on the device of the rows")
    if valid is not None:
        if valid.dtype == torch.bool:
            valid = valid.view(torch.uint8)
        if valid.dtype != torch.uint8 or valid.numel() != rows or not valid.is_contiguous() or valid.device != raw.device:
            raise ValueError("normalize_tiles: valid must be one contiguous byte per row")
    if out is None:
        out = torch.empty(raw.shape, dtype=out_dtype, device=raw.device)
    if mask_out is None:
        mask_out = torch.empty(raw.shape[:-1], dtype=torch.uint8, device=raw.device)
    if out.shape != raw.shape or out.dtype != out_dtype or not out.is_contiguous() or mask_out.numel() != rows:
        raise ValueError("normalize_tiles: output buffers do not match the rows")
    check(_lib.load().frl_normalize_tiles(_p(raw), _lib.F16 if raw.dtype == torch.float16 else F32, _p(valid) if valid is not None else None,
                                          _p(table), _p(out), _dt(out), _p(mask_out), rows, f, _stream()), "frl_normalize_tiles")
    return out, mask_out


@_timed("normalize_chunk_tiles")
def normalize_chunk_tiles(chunk: torch.Tensor, desc: torch.Tensor, tile: int, table: torch.Tensor, out_dtype: torch.dtype = torch.bfloat16,
                          out: Optional[torch.Tensor] = None, mask_out: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """One stored chunk [T, CY, CX, F] (float16 | float32) + tile descriptors [B, 4] int32 {y0, x0, h, w} ->
    normalised tiles [B, T, tile, tile, F] + validity bytes [B, T, tile, tile]; the tile cut and the zero padding of partial
    patches happen on the device (frl_normalize_chunk_tiles)."""
    if chunk.dim() != 4 or chunk.dtype not in (torch.float16, torch.float32) or not chunk.is_contiguous():
        raise ValueError("normalize_chunk_tiles: chunk must be a contiguous float16/float32 [T, CY, CX, F] tensor")
    t_, cy, cx, f = chunk.shape
    if desc.dtype != torch.int32 or desc.dim() != 2 or desc.shape[1] != 4 or not desc.is_contiguous() or desc.device != chunk.device:
        raise ValueError("normalize_chunk_tiles: desc must be a contiguous int32 [B, 4] tensor on the chunk's device")
    if table.dtype != torch.uint8 or table.numel() != 32 * f or table.device != chunk.device:
        raise ValueError("normalize_chunk_tiles: table must be the 32-byte-per-feature record image on the chunk's device")
    if out_dtype not in (torch.bfloat16, torch.float32):
        raise ValueError("normalize_chunk_tiles: output must be bfloat16 or float32")
    b = desc.shape[0]
    shape = (b, t_, tile, tile, f)
    if out is None:
        out = torch.empty(shape, dtype=out_dtype, device=chunk.device)
    if mask_out is None:
        mask_out = torch.empty(shape[:-1], dtype=torch.uint8, device=chunk.device)
    if tuple(out.shape) != shape or out.dtype != out_dtype or not out.is_contiguous() or mask_out.numel() != b * t_ * tile * tile \
            or not mask_out.is_contiguous():
        raise ValueError("normalize_chunk_tiles: output buffers do not match the tiles")
    check(_lib.load().frl_normalize_chunk_tiles(_p(chunk), _lib.F16 if chunk.dtype == torch.float16 else F32, t_, cy, cx, f, _p(desc), b, int(tile),
                                                _p(table), _p(out), _dt(out), _p(mask_out), _stream()), "frl_normalize_chunk_tiles")
    return out, mask_out


@_timed("add")
def add(a: torch.Tensor, b: torch.Tensor, scale_b: float = 1.0) -> torch.Tensor:
    """out = a + scale_b * b"""
    out = torch.empty_like(a)
    check(_lib.load().frl_add(_p(a), _p(b), float(scale_b), _p(out), a.numel(), _dt(a), _stream()), "frl_add")
    return out


# ----------------------------------------------------------------------------------------------
# 3x3 convolution (csrc/conv3x3.hip); x [B,H,W,Cin], w [Cout,Cin,3,3]
# ----------------------------------------------------------------------------------------------
@_timed("conv3x3_fwd")
def conv3x3_fwd(x, w, bias, act: int = ACT_NONE):
    b, h, wd, cin = x.shape
    cout = w.shape[0]
    _chk_rows(x, cin, "conv3x3.x")
    y = torch.empty(b, h, wd, cout, dtype=x.dtype, device=x.device)
    lib = _lib.load()
    ws = workspace(lib.frl_conv_workspace_bytes(cin, cout, 9), x.device)
    check(lib.frl_conv3x3_fwd(_p(x), _p(_f32(w, "w")), _p(_f32(bias, "bias")), _p(y), b, h, wd, cin, cout, act, _dt(x), _p(ws),
                              ws.numel(), _stream()), "frl_conv3x3_fwd")
    return y


@_timed("conv3x3_fwd_gate_blend")
def conv3x3_fwd_gate_blend(x, w, bias, smoothed, residual):
    """-> (out = smoothed + gate * residual, gate = sigmoid(conv3x3(x, w) + bias)): the blend of spatial.py:332-335 (min_gate = 0) in the
    convolution's epilogue."""
    b, h, wd, cin = x.shape
    cout = w.shape[0]
    _chk_rows(x, cin, "conv3x3.x")
    gate = torch.empty(b, h, wd, cout, dtype=x.dtype, device=x.device)
    _chk_like(smoothed, gate, "conv3x3_fwd_gate_blend.smoothed")
    _chk_like(residual, gate, "conv3x3_fwd_gate_blend.residual")
    out = torch.empty_like(gate)
    lib = _lib.load()
    ws = workspace(lib.frl_conv_workspace_bytes(cin, cout, 9), x.device)
    check(lib.frl_conv3x3_fwd_gate_blend(_p(x), _p(_f32(w, "w")), _p(_f32(bias, "bias")), _p(smoothed), _p(residual), _p(gate), _p(out), b, h, wd,
                                         cin, cout, _dt(x), _p(ws), ws.numel(), _stream()), "frl_conv3x3_fwd_gate_blend")
    return out, gate


@_timed("conv3x3_bwd_data")
def conv3x3_bwd_data(dy, w, y=None, act: int = ACT_NONE, add=None, sub_from=None, out_y=None, out_act: int = ACT_NONE):
    """dx = conv3x3^T(dy .* act'(y)) (+ add).  With sub_from the call returns (dx, sub_from - dx): both extras ride in the epilogue.
    out_y / out_act (instead of add / sub_from): dx .* out_act'(out_y) -- the gradient arrives at the next backward step already masked."""
    b, h, wd, cout = dy.shape
    cin = w.shape[1]
    dx = torch.empty(b, h, wd, cin, dtype=dy.dtype, device=dy.device)
    lib = _lib.load()
    ws = workspace(lib.frl_conv_workspace_bytes(cin, cout, 9), dy.device)
    if out_y is not None:
        if add is not None or sub_from is not None:
            raise ValueError("conv3x3_bwd_data: out_y does not combine with add / sub_from")
        _chk_like(out_y, dx, "conv3x3_bwd_data.out_y")
        check(lib.frl_conv3x3_bwd_data_outmask(_p(dy), _p(y), act, _p(_f32(w, "w")), _p(dx), _p(out_y), out_act, b, h, wd, cin, cout, _dt(dy),
                                               _p(ws), ws.numel(), _stream()), "frl_conv3x3_bwd_data_outmask")
        return dx
    if add is not None or sub_from is not None:
        out2 = None
        if add is not None:
            _chk_like(add, dx, "conv3x3_bwd_data.add")
        if sub_from is not None:
            _chk_like(sub_from, dx, "conv3x3_bwd_data.sub_from")
            out2 = torch.empty_like(dx)
        check(lib.frl_conv3x3_bwd_data_fused(_p(dy), _p(y), act, _p(_f32(w, "w")), _p(dx), _p(add), _p(sub_from), _p(out2), b, h, wd,
                                             cin, cout, _dt(dy), _p(ws), ws.numel(), _stream()), "frl_conv3x3_bwd_data_fused")
        return dx if sub_from is None else (dx, out2)
    check(lib.frl_conv3x3_bwd_data(_p(dy), _p(y), act, _p(_f32(w, "w")), _p(dx), b, h, wd, cin, cout, _dt(dy), _p(ws),
                                   ws.numel(), _stream()), "frl_conv3x3_bwd_data")
    return dx


@_timed("conv3x3_bwd_weight")
def conv3x3_bwd_weight(dy, x, y=None, act: int = ACT_NONE, scalar_frags: bool = False):
    b, h, wd, cout = dy.shape
    cin = x.shape[-1]
    lib = _lib.load()
    ws = workspace(lib.frl_conv3x3_bwd_weight_workspace_bytes(b, h, wd, cin, cout), dy.device)
    dw = torch.empty(cout, cin, 3, 3, dtype=torch.float32, device=dy.device)
    db = torch.empty(cout, dtype=torch.float32, device=dy.device)
    check(lib.frl_conv3x3_bwd_weight(_p(dy), _p(y), act, _p(x), _p(dw), _p(db), b, h, wd, cin, cout, _dt(dy), _p(ws),
                                     ws.numel(), 1 if scalar_frags else 0, _stream()), "frl_conv3x3_bwd_weight")
    return dw, db


# ----------------------------------------------------------------------------------------------
# fixed stencils of EdgeAwareSmoothingConv2D (csrc/stencil.hip)
# ----------------------------------------------------------------------------------------------
@_timed("sobel_fwd")
def sobel_fwd(x):
    b, h, w, c = x.shape
    g = torch.empty(b, h, w, 2 * c, dtype=x.dtype, device=x.device)
    check(_lib.load().frl_sobel_fwd(_p(x), _p(g), b, h, w, c, _dt(x), _stream()), "frl_sobel_fwd")
    return g


@_timed("sobel_bwd")
def sobel_bwd(dg, add=None):
    b, h, w, c2 = dg.shape
    dx = torch.empty(b, h, w, c2 // 2, dtype=dg.dtype, device=dg.device)
    if add is not None:
        _chk_like(add, dx, "sobel_bwd.add")
        check(_lib.load().frl_sobel_bwd_add(_p(dg), _p(dx), _p(add), b, h, w, c2 // 2, _dt(dg), _stream()), "frl_sobel_bwd_add")
        return dx
    check(_lib.load().frl_sobel_bwd(_p(dg), _p(dx), b, h, w, c2 // 2, _dt(dg), _stream()), "frl_sobel_bwd")
    return dx


@_timed("edge_smooth_fwd")
def edge_smooth_fwd(x, a_logit, b_logit, rank: int, coarse_dilation: int):
    b, h, w, c = x.shape
    sm, res = torch.empty_like(x), torch.empty_like(x)
    a_soft, b_soft = torch.empty_like(a_logit), torch.empty_like(b_logit)
    check(_lib.load().frl_edge_smooth_stencil_fwd(_p(x), _p(a_logit), _p(b_logit), _p(sm), _p(res), _p(a_soft), _p(b_soft),
                                                  b, h, w, c, rank, coarse_dilation, _dt(x), _stream()),
          "frl_edge_smooth_stencil_fwd")
    return sm, res, a_soft, b_soft


@_timed("edge_smooth_bwd")
def edge_smooth_bwd(d_smoothed, x, a_soft, b_soft, rank: int, coarse_dilation: int, dx_add: Optional[torch.Tensor] = None):
    """dx_add (optional, same shape and dtype as x): added to dx inside the kernel's store (the residual branch's gradient)."""
    b, h, w, c = x.shape
    dx = torch.empty_like(x)
    da, db = torch.empty_like(a_soft), torch.empty_like(b_soft)
    if dx_add is not None and (dx_add.shape != x.shape or dx_add.dtype != x.dtype or not dx_add.is_contiguous()):
        raise ValueError("edge_smooth_bwd: dx_add must match x")
    check(_lib.load().frl_edge_smooth_stencil_bwd(_p(d_smoothed), _p(x), _p(a_soft), _p(b_soft), _p(dx), _p(da), _p(db), _p(dx_add),
                                                  b, h, w, c, rank, coarse_dilation, _dt(x), _stream()),
          "frl_edge_smooth_stencil_bwd")
    return dx, da, db


def fusion_enabled(name: str) -> bool:
    """A/B switch for measurements: FRL_HIP_DISABLE=chain,film,heads routes the named fusions through their modular kernels."""
    import os
    return name not in os.environ.get("FRL_HIP_DISABLE", "").split(",")


def smooth_heads_supported(x: torch.Tensor, hidden: int, rank: int, wa, ba, wb, bb) -> bool:
    """True when the two mixing heads, their softmaxes and the directional bank run as ONE kernel per direction (csrc/smooth_fused.hip):
    bf16 rows of 64 channels, 64 hidden features, rank 4, both heads with a bias."""
    if ba is None or bb is None or x.dim() != 4 or not x.is_cuda or not fusion_enabled("heads"):
        return False
    if tuple(wa.shape[:2]) != (8 * rank, hidden) or tuple(wb.shape[:2]) != (x.shape[-1] * rank, hidden):
        return False
    return bool(_lib.load().frl_smooth_heads_supported(x.shape[-1], hidden, rank, _dt(x)))


@_timed("smooth_heads_fwd")
def smooth_heads_fwd(x, feat, wa, ba, wb, bb, coarse_dilation: int):
    """x, feat [B,H,W,64] bf16; wa [32,64(,1,1)], wb [256,64(,1,1)] f32 -> (smoothed, residual)."""
    b, h, w, c = x.shape
    _chk_rows(x, c, "smooth_heads_fwd.x")
    _chk_like(feat, x, "smooth_heads_fwd.feat")
    lib = _lib.load()
    ws = workspace(lib.frl_smooth_heads_workspace_bytes(b * h * w), x.device)
    sm, res = torch.empty_like(x), torch.empty_like(x)
    check(lib.frl_smooth_heads_fwd(_p(x), _p(feat), _p(_f32(wa, "wa")), _p(_f32(ba, "ba")), _p(_f32(wb, "wb")), _p(_f32(bb, "bb")), _p(sm), _p(res),
                                   b, h, w, coarse_dilation, _p(ws), ws.numel(), _stream()), "frl_smooth_heads_fwd")
    return sm, res


@_timed("smooth_heads_bwd")
def smooth_heads_bwd(d_smoothed, x, feat, wa, ba, wb, bb, coarse_dilation: int, dx_add: Optional[torch.Tensor] = None, dfeat_relu: bool = False):
    """-> (dx, dfeat [B,H,W,64] bf16, dwa, dba, dwb, dbb f32).  d_smoothed: gradient w.r.t. `smoothed` with the residual path folded in
    (d smoothed - d residual); dx_add: added to dx inside the kernel's store; dfeat_relu: dfeat comes back multiplied by [feat > 0]."""
    b, h, w, c = x.shape
    for t, n in ((d_smoothed, "d_smoothed"), (feat, "feat")):
        _chk_like(t, x, f"smooth_heads_bwd.{n}")
    if dx_add is not None:
        _chk_like(dx_add, x, "smooth_heads_bwd.dx_add")
    lib = _lib.load()
    npix = b * h * w
    ws = workspace(lib.frl_smooth_heads_workspace_bytes(npix), x.device)
    scratch = torch.empty(lib.frl_smooth_heads_bwd_scratch_bytes(npix), dtype=torch.uint8, device=x.device)
    dx, dfeat = torch.empty_like(x), torch.empty_like(feat)
    dwa, dwb = torch.empty(wa.shape, dtype=torch.float32, device=x.device), torch.empty(wb.shape, dtype=torch.float32, device=x.device)
    dba, dbb = torch.empty_like(ba), torch.empty_like(bb)
    check(lib.frl_smooth_heads_bwd_masked(_p(d_smoothed), _p(x), _p(feat), _p(_f32(wa, "wa")), _p(_f32(ba, "ba")), _p(_f32(wb, "wb")),
                                          _p(_f32(bb, "bb")), _p(dx_add), _p(dx), _p(dfeat), _p(dwa), _p(dba), _p(dwb), _p(dbb), _p(scratch),
                                          scratch.numel(), b, h, w, coarse_dilation, int(dfeat_relu), _p(ws), ws.numel(), _stream()),
          "frl_smooth_heads_bwd_masked")
    return dx, dfeat, dwa, dba, dwb, dbb


# ----------------------------------------------------------------------------------------------
# fused TCN block (csrc/tcn_fwd.hip, tcn_bwd.hip); x [B,T,HW..,Cin]
# ----------------------------------------------------------------------------------------------
def tcn_hot_supported(x: torch.Tensor, cin: int, cout: int, groups: int, dilation: int, has_projection: bool) -> bool:
    """True when the block on x [B,T,..,Cin] runs on the specialised `tcn_hot_*` kernels (the only ones that take a Dropout1d mask)."""
    return bool(_lib.load().frl_tcn_hot_supported(x.shape[1], cin, cout, groups, dilation, int(has_projection), _dt(x)))


def _check_drop_mask(drop_mask, x, hot: bool):
    if drop_mask is None:
        return
    if not hot:
        raise NotImplementedError("the generic TCN kernels take no Dropout1d mask (only the hot configuration: bf16, 64 channels, T = 5, "
                                  "8 groups, dilation 1/2/4); GatedResidualBlock routes other shapes through its two-input formulation")
    b, t, c = x.shape[0], x.shape[1], x.shape[-1]
    if drop_mask.dtype != x.dtype or not drop_mask.is_contiguous() or drop_mask.numel() != x.numel() // t:
        raise ValueError("drop_mask must be a contiguous [B, HW.., C] tensor of x's dtype")


@_timed("tcn_block_fwd")
def tcn_block_fwd(x, conv_w, conv_b, gn_w, gn_b, gate_w, gate_b, proj_w, proj_b, dilation: int, groups: int,
                  eps: float = 1e-5, allow_hot: bool = True, drop_mask=None):
    b, t, cin = x.shape[0], x.shape[1], x.shape[-1]
    cout = conv_w.shape[0]
    hw = x.numel() // (b * t * cin)
    y = torch.empty(x.shape[:-1] + (cout,), dtype=x.dtype, device=x.device)
    lib = _lib.load()
    hot = bool(allow_hot and lib.frl_tcn_hot_supported(t, cin, cout, groups, dilation, int(proj_w is not None), _dt(x)))
    _check_drop_mask(drop_mask, x, hot)
    if hot:
        ws = workspace(lib.frl_tcn_hot_fwd_workspace_bytes(), x.device)
        check(lib.frl_tcn_hot_fwd(_p(x), _p(drop_mask), _p(_f32(conv_w, "conv_w")), _p(conv_b), _p(gn_w), _p(gn_b),
                                  _p(_f32(gate_w.reshape(cout, cout), "gate_w")), _p(gate_b), _p(y), b * hw, hw, dilation, float(eps),
                                  _p(ws), ws.numel(), _stream()), "frl_tcn_hot_fwd")
        return y
    ws = workspace(lib.frl_conv_workspace_bytes(max(cin, cout), cout, 6), x.device)
    check(lib.frl_tcn_block_fwd(_p(x), _p(_f32(conv_w, "conv_w")), _p(conv_b), _p(gn_w), _p(gn_b),
                                _p(_f32(gate_w.reshape(cout, cout), "gate_w")), _p(gate_b), _p(proj_w), _p(proj_b),
                                _p(y), b * hw, hw, t, cin, cout, dilation, groups, float(eps), _dt(x), _p(ws), ws.numel(),
                                _stream()), "frl_tcn_block_fwd")
    return y


def tcn_chain_supported(x: torch.Tensor, blocks, head_w: torch.Tensor) -> bool:
    """True when three blocks (dilation 1, 2, 4; hot configuration each) and the 1x1 head run as ONE forward launch (tcn_chain_fwd_kernel).
    blocks: three tuples (conv_w, conv_b, gn_w, gn_b, gate_w, gate_b, dilation, groups, has_projection)."""
    if len(blocks) != 3 or x.dim() < 3 or not x.is_cuda or not fusion_enabled("chain"):
        return False
    lib = _lib.load()
    t, c = x.shape[1], x.shape[-1]
    for blk, dil in zip(blocks, (1, 2, 4)):
        if blk[6] != dil or not lib.frl_tcn_hot_supported(t, c, blk[0].shape[0], blk[7], dil, int(blk[8]), _dt(x)):
            return False
    ch = head_w.shape[0]
    return head_w.shape[1] == 64 and ch in (4, 8, 12, 16)


@_timed("tcn_chain_fwd")
def tcn_chain_fwd(x, blocks, head_w, head_b, eps: float = 1e-5):
    """x [B,5,HW..,64] -> (y1, y2, y3 [same shape], h [B,5,HW..,Ch])."""
    b, t, c = x.shape[0], x.shape[1], x.shape[-1]
    hw = x.numel() // (b * t * c)
    _chk_rows(x, c, "tcn_chain_fwd.x")
    ch = head_w.shape[0]
    lib = _lib.load()
    ws = workspace(lib.frl_tcn_chain_fwd_workspace_bytes(), x.device)
    ys = [torch.empty_like(x) for _ in range(3)]
    h = torch.empty(x.shape[:-1] + (ch,), dtype=x.dtype, device=x.device)
    arr = ctypes.c_void_p * 3
    cols = [arr(*[_f32(blk[i] if i != 4 else blk[i].reshape(c, c), "tcn parameter").data_ptr() for blk in blocks]) for i in range(6)]
    check(lib.frl_tcn_chain_fwd(_p(x), *[ctypes.cast(a, ctypes.c_void_p) for a in cols], _p(_f32(head_w.reshape(ch, c), "head_w")), _p(_f32(head_b, "head_b")),
                                _p(ys[0]), _p(ys[1]), _p(ys[2]), _p(h), b * hw, hw, ch, float(eps), _p(ws), ws.numel(), _stream()),
          "frl_tcn_chain_fwd")
    return ys[0], ys[1], ys[2], h


@_timed("tcn_block_bwd")
def tcn_block_bwd_head_supported(x, dh, head_w, dilation: int) -> bool:
    """True when the last phase block's backward can take the head's output gradient dh directly (csrc/tcn_hot_bwd4.hip, HEAD variant)."""
    b, t, cin = x.shape[0], x.shape[1], x.shape[-1]
    hw = x.numel() // (b * t * cin)
    lib = _lib.load()
    return bool(fusion_enabled("headbwd") and dilation == 4 and x.dtype == torch.bfloat16 and dh.dtype == torch.bfloat16 and
                lib.frl_tcn_hot_supported(t, cin, cin, 8, dilation, 0, _dt(x)) and
                lib.frl_tcn_hot_bwd_head_supported(b * hw, hw, int(head_w.shape[0])))


@_timed("tcn_block_bwd")
def tcn_block_bwd_head(x, dh, head_w, conv_w, conv_b, gn_w, gn_b, gate_w, gate_b, dilation: int, eps: float = 1e-5):
    """Backward of the last hot block fed with dh [B,T,HW..,Ch] (gradient of the 1x1 phase head's output) and head_w [Ch,64]: dy = dh head_w is
    formed inside the kernel.  Returns the dict of tcn_block_bwd."""
    b, t, cin = x.shape[0], x.shape[1], x.shape[-1]
    cout = conv_w.shape[0]
    hw = x.numel() // (b * t * cin)
    npix = b * hw
    ch = int(head_w.shape[0])
    _chk_rows(dh, ch, "tcn_block_bwd_head.dh")
    lib = _lib.load()
    dx = torch.empty_like(x)
    g = {k: torch.empty_like(v, dtype=torch.float32) for k, v in
         dict(conv_w=conv_w, conv_b=conv_b, gn_w=gn_w, gn_b=gn_b, gate_w=gate_w, gate_b=gate_b).items()}
    ws = workspace(lib.frl_tcn_hot_bwd_head_workspace_bytes(npix), x.device)
    with span("tcn_block_bwd.main"):
        check(lib.frl_tcn_hot_bwd_head(_p(x), _p(dh), _p(_f32(head_w.reshape(ch, cin), "head_w")), ch, _p(conv_w), _p(conv_b), _p(gn_w), _p(gn_b),
                                       _p(gate_w.reshape(cout, cout)), _p(gate_b), _p(dx), _p(g["conv_w"]), _p(g["conv_b"]), _p(g["gn_w"]),
                                       _p(g["gn_b"]), _p(g["gate_w"]), _p(g["gate_b"]), npix, hw, dilation, float(eps), _p(ws), ws.numel(),
                                       _stream()), "frl_tcn_hot_bwd_head")
    g["dx"] = dx
    return g


def tcn_block_bwd(x, dy, conv_w, conv_b, gn_w, gn_b, gate_w, gate_b, proj_w, proj_b, dilation: int, groups: int,
                  eps: float = 1e-5, allow_fused: bool = True, allow_hot: bool = True, drop_mask=None, want_dx: bool = True):
    """Returns dict(dx, conv_w, conv_b, gn_w, gn_b, gate_w, gate_b[, proj_w, proj_b]) gradients.  want_dx=False (the block's input
    needs no gradient): dx is None where the kernel can skip it (hot configuration without mask), otherwise computed as usual."""
    b, t, cin = x.shape[0], x.shape[1], x.shape[-1]
    cout = conv_w.shape[0]
    hw = x.numel() // (b * t * cin)
    npix = b * hw
    lib = _lib.load()
    dev = x.device
    hot = bool(allow_fused and allow_hot and lib.frl_tcn_hot_supported(t, cin, cout, groups, dilation, int(proj_w is not None), _dt(x)))
    _check_drop_mask(drop_mask, x, hot)
    if hot:
        skip_dx = (not want_dx) and drop_mask is None and bool(lib.frl_tcn_hot_bwd_nodx_supported(npix, hw))
        dx = None if skip_dx else torch.empty_like(x)
        g = {k: torch.empty_like(v, dtype=torch.float32) for k, v in
             dict(conv_w=conv_w, conv_b=conv_b, gn_w=gn_w, gn_b=gn_b, gate_w=gate_w, gate_b=gate_b).items()}
        ws = workspace(lib.frl_tcn_hot_bwd_workspace_bytes(npix), dev)
        with span("tcn_block_bwd.main"):
            check(lib.frl_tcn_hot_bwd(_p(x), _p(drop_mask), _p(dy), _p(conv_w), _p(conv_b), _p(gn_w), _p(gn_b), _p(gate_w.reshape(cout, cout)), _p(gate_b),
                                      _p(dx), _p(g["conv_w"]), _p(g["conv_b"]), _p(g["gn_w"]), _p(g["gn_b"]), _p(g["gate_w"]),
                                      _p(g["gate_b"]), npix, hw, dilation, float(eps), _p(ws), ws.numel(), _stream()), "frl_tcn_hot_bwd")
        g["dx"] = dx
        return g
    if allow_fused and lib.frl_tcn_block_bwd_fused_supported(t, cin, cout, groups, int(proj_w is not None), _dt(x)):
        dx = torch.empty_like(x)
        g = {k: torch.empty_like(v, dtype=torch.float32) for k, v in
             dict(conv_w=conv_w, conv_b=conv_b, gn_w=gn_w, gn_b=gn_b, gate_w=gate_w, gate_b=gate_b).items()}
        ws = workspace(lib.frl_tcn_block_bwd_fused_workspace_bytes(npix), dev)
        with span("tcn_block_bwd.main"):
            check(lib.frl_tcn_block_bwd_fused(_p(x), _p(dy), _p(conv_w), _p(conv_b), _p(gn_w), _p(gn_b),
                                              _p(gate_w.reshape(cout, cout)), _p(gate_b), _p(dx), _p(g["conv_w"]), _p(g["conv_b"]),
                                              _p(g["gn_w"]), _p(g["gn_b"]), _p(g["gate_w"]), _p(g["gate_b"]), npix, hw, t, dilation,
                                              groups, float(eps), _p(ws), ws.numel(), _stream()), "frl_tcn_block_bwd_fused")
        g["dx"] = dx
        return g
    oshape = x.shape[:-1] + (cout,)
    dconv = torch.empty(oshape, dtype=x.dtype, device=dev)
    dgpre, normed, dres = torch.empty_like(dconv), torch.empty_like(dconv), torch.empty_like(dconv)
    dgam = torch.empty(cout, dtype=torch.float32, device=dev)
    dbet = torch.empty(cout, dtype=torch.float32, device=dev)
    gate_w2 = gate_w.reshape(cout, cout)
    ws = workspace(lib.frl_tcn_block_bwd_workspace_bytes(npix, cout), dev)
    with span("tcn_block_bwd.main"):
        check(lib.frl_tcn_block_bwd(_p(x), _p(dy), _p(conv_w), _p(conv_b), _p(gn_w), _p(gn_b), _p(gate_w2), _p(gate_b), _p(proj_w),
                                    _p(proj_b), _p(dconv), _p(dgpre), _p(normed), _p(dres), _p(dgam), _p(dbet), npix, hw, t, cin,
                                    cout, dilation, groups, float(eps), _dt(x), _p(ws), ws.numel(), _stream()), "frl_tcn_block_bwd")
    dx = torch.empty_like(x)
    ws1 = workspace(lib.frl_conv_workspace_bytes(max(cin, cout), max(cin, cout), 6), dev)
    check(lib.frl_tcn_block_bwd_data(_p(dconv), _p(dres), _p(conv_w), _p(proj_w), _p(dx), npix, hw, t, cin, cout, dilation,
                                     _dt(x), _p(ws1), ws1.numel(), _stream()), "frl_tcn_block_bwd_data")
    p = b * t * hw
    dw = torch.empty(cout, cin, 3, dtype=torch.float32, device=dev)
    dcb = torch.empty(cout, dtype=torch.float32, device=dev)
    ws2 = workspace(lib.frl_conv1x1_bwd_weight_workspace_bytes(p, max(cin, cout), cout), dev)
    for k in range(3):
        check(lib.frl_conv_tap_bwd_weight(_p(dconv), None, 0, _p(x), ctypes.c_void_p(dw.data_ptr() + 4 * k), cin * 3, 3,
                                          _p(dcb) if k == 1 else None, p, cin, cout, hw, t, (k - 1) * dilation, _dt(x),
                                          _p(ws2), ws2.numel(), 0, _stream()), "frl_conv_tap_bwd_weight")
    dgw, dgb = _conv1x1_bwd_weight_impl(dgpre, normed)
    out = dict(dx=dx, conv_w=dw, conv_b=dcb, gn_w=dgam, gn_b=dbet, gate_w=dgw.reshape(gate_w.shape), gate_b=dgb)
    if proj_w is not None:
        dpw, dpb = _conv1x1_bwd_weight_impl(dres, x)
        out.update(proj_w=dpw.reshape(proj_w.shape), proj_b=dpb)
    return out


# ----------------------------------------------------------------------------------------------
# fused decoder + reconstruction loss (csrc/dec_fused.hip)
# ----------------------------------------------------------------------------------------------
def decoder_mse_supported(cz: int, hidden: int, features: int, t: torch.Tensor) -> bool:
    return bool(_lib.load().frl_decoder_mse_fused_supported(cz, hidden, features, _dt(t)))


@_timed("decoder_mse_fwd")
def decoder_mse_fwd(z, w1, b1, w2, b2, target, mask=None, want_xhat: bool = False):
    """z [..., Cz], target [..., 64] -> (stats f32 [2] = {mse, n_valid}, xhat or None)."""
    cz = z.shape[-1]
    p = z.numel() // cz
    lib = _lib.load()
    ws = workspace(lib.frl_decoder_mse_workspace_bytes(p, cz), z.device)
    out = torch.empty(2, dtype=torch.float32, device=z.device)
    xhat = torch.empty_like(target) if want_xhat else None
    check(lib.frl_decoder_mse_fwd(_p(z), _p(w1), _p(b1), _p(w2), _p(b2), _p(target), _p(mask), _p(xhat), _p(out), p, cz, _p(ws),
                                  ws.numel(), _stream()), "frl_decoder_mse_fwd")
    return out, xhat


@_timed("decoder_mse_bwd")
def decoder_mse_bwd(z, w1, b1, w2, b2, target, mask, gscale, stats):
    cz = z.shape[-1]
    p = z.numel() // cz
    lib = _lib.load()
    ws = workspace(lib.frl_decoder_mse_workspace_bytes(p, cz), z.device)
    dz = torch.empty_like(z)
    dw1, db1 = torch.empty_like(w1, dtype=torch.float32), torch.empty_like(b1, dtype=torch.float32)
    dw2, db2 = torch.empty_like(w2, dtype=torch.float32), torch.empty_like(b2, dtype=torch.float32)
    check(lib.frl_decoder_mse_bwd(_p(z), _p(w1), _p(b1), _p(w2), _p(b2), _p(target), _p(mask), _p(gscale), _p(stats), _p(dz), _p(dw1),
                                  _p(db1), _p(dw2), _p(db2), p, cz, _p(ws), ws.numel(), _stream()), "frl_decoder_mse_bwd")
    return dz, dw1, db1, dw2, db2


# ----------------------------------------------------------------------------------------------
# sparse-location gather + InfoNCE over mined pairs (csrc/contrastive.hip)
# ----------------------------------------------------------------------------------------------
@_timed("gather_locations")
def gather_locations(feature: torch.Tensor, coords: torch.Tensor) -> torch.Tensor:
    """feature [C, H, W] with any strides (float32 | bfloat16), coords int64 [N, 2] (row, col) -> [N, C] contiguous."""
    if not feature.is_cuda:
        raise _lib.FrlHipError("gather_locations: tensor must live on the GPU (no CPU fallback)")
    c, h, w = feature.shape
    if coords.dtype != torch.int64 or not coords.is_contiguous() or coords.device != feature.device:
        raise ValueError("gather_locations: coords must be a contiguous int64 [N, 2] tensor on the feature's device")
    n = coords.shape[0]
    out = torch.empty(n, c, dtype=feature.dtype, device=feature.device)
    sc, sh, sw = feature.stride()
    check(_lib.load().frl_gather_locations_fwd(_p(feature), sc, sh, sw, c, h, w, _p(coords), n, _p(out), _dt(feature), _stream()),
          "frl_gather_locations_fwd")
    return out


@_timed("segment_sum_rows")
def segment_sum_rows(vals: torch.Tensor, order: Optional[torch.Tensor], keys_sorted: torch.Tensor, out: torch.Tensor, accumulate: bool = False):
    """out[key] (+)= sum of vals[order[i]] over each run of equal keys_sorted[i]; rows are summed in list order (reproducible)."""
    m, d = vals.shape
    if vals.dtype != torch.float32 or not vals.is_contiguous() or out.dtype != torch.float32 or not out.is_contiguous() or out.shape[-1] != d:
        raise ValueError("segment_sum_rows: vals [M, D] and out [.., D] must be contiguous float32")
    for t in (order, keys_sorted):
        if t is not None and (t.dtype != torch.int64 or not t.is_contiguous() or t.numel() != m):
            raise ValueError("segment_sum_rows: order / keys must be contiguous int64 [M]")
    check(_lib.load().frl_segment_sum_rows(_p(vals), _p(order), _p(keys_sorted), m, d, _p(out), d, int(accumulate), _stream()),
          "frl_segment_sum_rows")
    return out


@_timed("infonce_fwd")
def infonce_fwd(emb, pairs, weights, is_pos, seg, temperature: float, sim: int, want_coef: bool = True):
    """Pairs sorted by anchor with segment bounds `seg` -> (loss f32 [1], sims [T], coef [T] or None)."""
    t, d = pairs.shape[0], emb.shape[1]
    sims = torch.empty(t, dtype=torch.float32, device=emb.device)
    logits = torch.empty_like(sims)
    nseg = seg.numel() - 1
    loss_a = torch.empty(nseg, dtype=torch.float32, device=emb.device)
    coef = torch.empty_like(sims) if want_coef else None
    loss = torch.empty(1, dtype=torch.float32, device=emb.device)
    check(_lib.load().frl_infonce_fwd(_p(emb), d, _p(pairs), _p(weights), _p(is_pos), t, _p(seg), nseg, float(temperature), int(sim),
                                      _p(sims), _p(logits), _p(loss_a), _p(coef), _p(loss), _stream()), "frl_infonce_fwd")
    return loss, sims, coef


@_timed("infonce_bwd")
def infonce_pair_grads(emb, pairs, sims, coef, gscale, nseg: int, temperature: float, sim: int):
    t, d = pairs.shape[0], emb.shape[1]
    ga = torch.empty(t, d, dtype=torch.float32, device=emb.device)
    gb = torch.empty_like(ga)
    check(_lib.load().frl_infonce_pair_grads(_p(emb), d, _p(pairs), _p(sims), _p(coef), _p(gscale), t, nseg, float(temperature), int(sim),
                                             _p(ga), _p(gb), _stream()), "frl_infonce_pair_grads")
    return ga, gb
