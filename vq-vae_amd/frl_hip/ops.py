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


def _f32(t: Optional[torch.Tensor], name: str):
    if t is None:
        return None
    if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
        raise ValueError(f"{name}: parameters are passed as contiguous float32 CUDA tensors")
    return t


def workspace(nbytes: int, device) -> torch.Tensor:
    """Grow-only per-(device, stream) scratch buffer handed to kernels that need one."""
    key = (str(device), torch.cuda.current_stream().cuda_stream)
    ws = _WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _WS[key] = ws
    return ws


# ----------------------------------------------------------------------------------------------
# pointwise convolution (reference: nn.Conv2d(.,.,1) call sites, see csrc/pw_conv.hip)
# ----------------------------------------------------------------------------------------------
def conv1x1_fwd(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], act: int = ACT_NONE) -> torch.Tensor:
    cout, cin = w.shape[0], w.shape[1]
    _chk_rows(x, cin, "conv1x1_fwd.x")
    w = _f32(w.reshape(cout, cin), "w")
    y = torch.empty(x.shape[:-1] + (cout,), dtype=x.dtype, device=x.device)
    p = x.numel() // cin
    check(_lib.load().frl_conv1x1_fwd(_p(x), _p(w), _p(_f32(bias, "bias")), _p(y), p, cin, cout, act, _dt(x), _stream()),
          "frl_conv1x1_fwd")
    return y


def conv1x1_bwd_data(dy: torch.Tensor, w: torch.Tensor, y: Optional[torch.Tensor] = None, act: int = ACT_NONE) -> torch.Tensor:
    cout, cin = w.shape[0], w.shape[1]
    _chk_rows(dy, cout, "conv1x1_bwd_data.dy")
    w = _f32(w.reshape(cout, cin), "w")
    dx = torch.empty(dy.shape[:-1] + (cin,), dtype=dy.dtype, device=dy.device)
    p = dy.numel() // cout
    check(_lib.load().frl_conv1x1_bwd_data(_p(dy), _p(y), act, _p(w), _p(dx), p, cin, cout, _dt(dy), _stream()),
          "frl_conv1x1_bwd_data")
    return dx


def conv1x1_bwd_weight(dy: torch.Tensor, x: torch.Tensor, y: Optional[torch.Tensor] = None, act: int = ACT_NONE,
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


# ----------------------------------------------------------------------------------------------
# vector quantizer (csrc/vq.hip)
# ----------------------------------------------------------------------------------------------
def vq_assign(z: torch.Tensor, codebook: torch.Tensor):
    """z [N,d] rows, codebook [K,d] f32 -> (idx int32 [N], z_q [N,d], stats f32 [4], counts int32 [K])."""
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
    check(lib.frl_vq_assign_fwd(_p(z), _p(cb), n, k, d, _p(idx), _p(zq), _p(stats), _p(counts), _dt(z), _p(ws),
                                ws.numel(), _stream()), "frl_vq_assign_fwd")
    return idx, zq, stats, counts


def vq_bwd(g_out: Optional[torch.Tensor], z: torch.Tensor, codebook: torch.Tensor, idx: torch.Tensor,
           counts: torch.Tensor, gscale: Optional[torch.Tensor], beta: float, want_gz: bool = True,
           want_ge: bool = True, want_sums: bool = False):
    k, d = codebook.shape
    n = z.numel() // d
    lib = _lib.load()
    ws = workspace(lib.frl_vq_workspace_bytes(n, k, d), z.device)
    gz = torch.empty_like(z) if want_gz else None
    ge = torch.empty(k, d, dtype=torch.float32, device=z.device) if want_ge else None
    sums = torch.empty(k, d, dtype=torch.float32, device=z.device) if want_sums else None
    check(lib.frl_vq_bwd(_p(g_out), _p(z), _p(_f32(codebook, "codebook")), _p(idx), _p(counts), _p(gscale), float(beta),
                         n, k, d, _p(gz), _p(ge), _p(sums), _dt(z), _p(ws), ws.numel(), _stream()), "frl_vq_bwd")
    return gz, ge, sums


def vq_ema_update(sums, counts, ema_count, ema_sum, codebook, decay: float, eps: float):
    k, d = codebook.shape
    check(_lib.load().frl_vq_ema_update(_p(sums), _p(counts), k, d, float(decay), float(eps), _p(ema_count),
                                        _p(ema_sum), _p(codebook), _stream()), "frl_vq_ema_update")
