"""CPU oracle for the VQ-VAE training hot path (TEST INFRASTRUCTURE ONLY).

This file is a plain CPU PyTorch/numpy restatement of the reference algorithm
on the hot path.  It is imported only by ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg -- never by the product package
(``vq-vae_amd/frl_hip``), which must fail loudly when its HIP library is missing.

Pinning status
--------------
* Encoder (type path, phase path): **pinned** -- checked tensor-for-tensor
  against the imported reference model by ``oracle/make_golden.py`` (run in the
  build container; the fixtures it wrote live in ``tests/golden/``).
* VQ step, decoder, train-step skeleton: **parity unpinned by the reference** --
  the reference no longer ships a quantizer or decoder (SURVEY.md section 8a rows
  a11/a12).  They follow the standard VQ-VAE definition with the constants the
  reference's configs keep (frl/config/frl_model_v0.yaml:29-35,
  frl/config/frl_bindings_v0.yaml:887-891, scripts/train_vqvae.py:410-436) and
  are pinned by self-consistency tests only (fp64 brute force, gradcheck).
* Tile ingest (per-channel normalisation + masking, ``normalize_tiles_np``): **parity
  unpinned by the reference** -- FeatureBuilder / Normalizer import ``zarr`` (absent
  in this image) and the reference keeps no fixture for them; restated from
  frl/data/loaders/builders/feature_builder.py:487-548,709-737.

All functions take a flat ``state`` dict with the reference's state-dict key
names (frl/models/representation.py) and tensors in the reference's NCHW / NCT
layouts, so the restatement can be diffed against the reference line by line.
Every function is dtype generic: pass float64 tensors for golden values.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

State = Dict[str, torch.Tensor]


# ----------------------------------------------------------------------------
# Building blocks
# ----------------------------------------------------------------------------
def group_norm(x: torch.Tensor, groups: int, weight, bias, eps: float = 1e-5):
    """nn.GroupNorm: biased variance over (C/G, *spatial) per sample.

    Reference: frl/models/conv2d_encoder.py:117, frl/models/tcn.py:65,92.
    """
    n, c = x.shape[0], x.shape[1]
    xg = x.reshape(n, groups, -1)
    mean = xg.mean(dim=2, keepdim=True)
    var = ((xg - mean) ** 2).mean(dim=2, keepdim=True)
    xh = ((xg - mean) / torch.sqrt(var + eps)).reshape(x.shape)
    shape = [1, c] + [1] * (x.dim() - 2)
    return xh * weight.reshape(shape) + bias.reshape(shape)


def conv2d_encoder_forward(state: State, x: torch.Tensor, num_groups: int,
                           prefix: str = "encoder.") -> torch.Tensor:
    """Conv2DEncoder.forward in eval / dropout-free mode.

    Reference: frl/models/conv2d_encoder.py:100-127,150-159.  Layer stack
    ``[conv1x1(bias=False) -> GroupNorm -> ReLU (-> Dropout2d)] x (L-1) ->
    conv1x1 -> GroupNorm`` -- the last layer stops after GroupNorm.  The
    Sequential indices depend on whether Dropout2d modules exist, so the conv /
    norm layers are discovered from the keys present.
    """
    idx = sorted({int(k[len(prefix) + 7:].split(".")[0]) for k in state
                  if k.startswith(prefix + "layers.")})
    convs = [i for i in idx if state[f"{prefix}layers.{i}.weight"].dim() == 4]
    h = x
    for n, ci in enumerate(convs):
        w = state[f"{prefix}layers.{ci}.weight"]
        h = F.conv2d(h, w)
        h = group_norm(h, num_groups, state[f"{prefix}layers.{ci + 1}.weight"],
                       state[f"{prefix}layers.{ci + 1}.bias"])
        if n != len(convs) - 1:
            h = F.relu(h)
    return h


def edge_smooth_forward(state: State, x: torch.Tensor, rank: int = 4,
                        num_directions: int = 4, coarse_dilation: int = 3,
                        min_gate: float = 0.0, prefix: str = "spatial_conv."
                        ) -> Tuple[torch.Tensor, torch.Tensor, Dict[str, torch.Tensor]]:
    """EdgeAwareSmoothingConv2D.forward -> (output, gate, intermediates).

    Reference: frl/models/spatial.py:278-339 (pipeline), :224-249 (fixed bank /
    Sobel buffers), :258-272 (learned nets), :333-335 (gate floor).
    """
    b, c, hh, ww = x.shape
    k = num_directions * 2
    r = rank
    sobel_x = state[prefix + "sobel_x"].to(x.dtype)
    sobel_y = state[prefix + "sobel_y"].to(x.dtype)
    bank = state[prefix + "bank"].to(x.dtype)
    dx = F.conv2d(x, sobel_x, padding=1, groups=c)
    dy = F.conv2d(x, sobel_y, padding=1, groups=c)
    feat = F.relu(F.conv2d(torch.cat([dx, dy], 1),
                           state[prefix + "mix_backbone.0.weight"],
                           state[prefix + "mix_backbone.0.bias"], padding=1))
    a_logit = F.conv2d(feat, state[prefix + "mix_head_A.weight"], state[prefix + "mix_head_A.bias"])
    b_logit = F.conv2d(feat, state[prefix + "mix_head_B.weight"], state[prefix + "mix_head_B.bias"])
    a = torch.softmax(a_logit.reshape(b, k, r, hh, ww), dim=1)
    bw = torch.softmax(b_logit.reshape(b, c, r, hh, ww), dim=2)
    slot = torch.zeros(b, c, r, hh, ww, dtype=x.dtype)
    for i in range(num_directions):
        filt = bank[i]
        fine = F.conv2d(x, filt, padding=1, groups=c)
        coarse = F.conv2d(x, filt, padding=coarse_dilation, dilation=coarse_dilation, groups=c)
        slot = slot + fine.unsqueeze(2) * a[:, 2 * i].unsqueeze(1)
        slot = slot + coarse.unsqueeze(2) * a[:, 2 * i + 1].unsqueeze(1)
    smoothed = (bw * slot).sum(dim=2)
    residual = x - smoothed
    g1 = F.relu(F.conv2d(residual, state[prefix + "gate_net.0.weight"],
                         state[prefix + "gate_net.0.bias"], padding=1))
    gate = torch.sigmoid(F.conv2d(g1, state[prefix + "gate_net.2.weight"],
                                  state[prefix + "gate_net.2.bias"], padding=1))
    if min_gate > 0.0:
        gate = gate.clamp(min=min_gate)
    out = smoothed + gate * residual
    inter = dict(dx=dx, dy=dy, feat=feat, a_logit=a_logit, b_logit=b_logit,
                 smoothed=smoothed, residual=residual, g1=g1)
    return out, gate, inter


def tcn_block_forward(state: State, x: torch.Tensor, dilation: int, num_groups: int,
                      prefix: str, drop_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """GatedResidualBlock.forward on [N, C, T]; ``drop_mask`` [N, C] (0 or 1/(1-p)) is a given
    Dropout1d realisation applied to the conv input only (tcn.py:53,89-90).

    Reference: frl/models/tcn.py:78-111.  ``gate`` is computed from the
    pre-ReLU normalised features; the residual projection exists only when
    C_in != C_out (:71-76).
    """
    w = state[prefix + "conv.weight"]
    ksz = w.shape[2]
    pad = (ksz - 1) * dilation // 2
    if (prefix + "projection.weight") in state:
        res = F.conv1d(x, state[prefix + "projection.weight"], state[prefix + "projection.bias"])
    else:
        res = x
    xin = x if drop_mask is None else x * drop_mask.unsqueeze(-1)
    out = F.conv1d(xin, w, state[prefix + "conv.bias"], padding=pad, dilation=dilation)
    out = group_norm(out, num_groups, state[prefix + "norm.weight"], state[prefix + "norm.bias"])
    gate = torch.sigmoid(F.conv1d(out, state[prefix + "gate.weight"], state[prefix + "gate.bias"]))
    out = F.relu(out)
    return gate * out + (1 - gate) * res


def tcn_forward(state: State, x: torch.Tensor, dilations, num_groups: int,
                prefix: str = "phase_tcn.") -> torch.Tensor:
    """TCNEncoder.forward with pooling='none' for [N,C,T] or [B,C,T,H,W].

    Reference: frl/models/tcn.py:220-302 (5-D path flattens pixels :242-248 and
    restores :289-290).
    """
    spatial = x.dim() == 5
    if spatial:
        b, c, t, hh, ww = x.shape
        x = x.permute(0, 3, 4, 1, 2).reshape(b * hh * ww, c, t)
    for i, d in enumerate(dilations):
        x = tcn_block_forward(state, x, d, num_groups, f"{prefix}layers.{i}.")
    if spatial:
        x = x.reshape(b, hh, ww, x.shape[1], t).permute(0, 3, 4, 1, 2)
    return x


def film_forward(state: State, cond: torch.Tensor, prefix: str = "phase_film."):
    """FiLMLayer.forward: two independent conv1x1 -> ReLU -> conv1x1 nets.

    Reference: frl/models/conditioning.py:55-67,82-102.
    """
    def net(name):
        h = F.relu(F.conv2d(cond, state[f"{prefix}{name}.0.weight"], state[f"{prefix}{name}.0.bias"]))
        return F.conv2d(h, state[f"{prefix}{name}.2.weight"], state[f"{prefix}{name}.2.bias"])
    return net("gamma_network"), net("beta_network")


# ----------------------------------------------------------------------------
# RepresentationModel restatement
# ----------------------------------------------------------------------------
def model_forward(state: State, x: torch.Tensor, hp: dict):
    """RepresentationModel.forward(x, return_gate=True) -> (z_type, gate, h).

    Reference: frl/models/representation.py:317-334.
    """
    h = conv2d_encoder_forward(state, x, hp.get("type_encoder_num_groups", 8))
    z, gate, _ = edge_smooth_forward(state, h, rank=hp.get("spatial_conv_rank", 4),
                                     num_directions=hp.get("spatial_conv_num_directions", 4),
                                     coarse_dilation=hp.get("spatial_conv_coarse_dilation", 3),
                                     min_gate=hp.get("min_gate", 0.0))
    return z, gate, h


def model_forward_phase(state: State, x_phase: torch.Tensor, z_type: torch.Tensor, hp: dict):
    """RepresentationModel.forward_phase (dense).  Caller stop-grads z_type.

    Reference: frl/models/representation.py:336-374.
    """
    b, c, t, hh, ww = x_phase.shape
    h = tcn_forward(state, x_phase, hp.get("phase_tcn_dilations", (1, 2, 4)),
                    hp.get("phase_tcn_num_groups", 8))
    co = h.shape[1]
    h = h.permute(0, 2, 1, 3, 4).reshape(b * t, co, hh, ww)
    h = F.conv2d(h, state["phase_head.weight"], state["phase_head.bias"])
    zp = h.shape[1]
    h = h.reshape(b, t, zp, hh, ww).permute(0, 2, 1, 3, 4)
    gamma, beta = film_forward(state, z_type)
    return gamma.unsqueeze(2) * h + beta.unsqueeze(2)


def model_forward_phase_at_locations(state: State, x_px: torch.Tensor, z_px: torch.Tensor, hp: dict):
    """RepresentationModel.forward_phase_at_locations -> (z [N,T,zp], gamma, beta, h).

    Reference: frl/models/representation.py:376-436.
    """
    n, c, t = x_px.shape
    h = tcn_forward(state, x_px, hp.get("phase_tcn_dilations", (1, 2, 4)),
                    hp.get("phase_tcn_num_groups", 8))
    co = h.shape[1]
    h = h.permute(0, 2, 1).reshape(n * t, co, 1, 1)
    h = F.conv2d(h, state["phase_head.weight"], state["phase_head.bias"])
    zp = h.shape[1]
    h = h.reshape(n, t, zp).permute(0, 2, 1)
    gamma, beta = film_forward(state, z_px.unsqueeze(-1).unsqueeze(-1))
    gamma, beta = gamma.squeeze(-1), beta.squeeze(-1)
    z = (gamma * h + beta).permute(0, 2, 1)
    return z, gamma.squeeze(-1), beta.squeeze(-1), h


# ----------------------------------------------------------------------------
# Vector quantizer (build definition -- parity unpinned by the reference)
# ----------------------------------------------------------------------------
def vq_argmin_np(z: np.ndarray, e: np.ndarray, chunk: int = 4096) -> np.ndarray:
    """idx_n = argmin_k sum_j (z_nj - e_kj)^2 evaluated in float64, first index on ties.

    Constants/spec: frl/config/frl_model_v0.yaml:29-35 (kind vq, K=256, d=64);
    the reference's sibling pairwise-L2 idiom is torch.cdist + topk at
    frl/losses/pairs.py:570,591.  Direct differences (not the expanded form), so
    the oracle carries no cancellation error.
    """
    z = np.asarray(z, dtype=np.float64)
    e = np.asarray(e, dtype=np.float64)
    out = np.empty(z.shape[0], dtype=np.int64)
    for s in range(0, z.shape[0], chunk):
        zz = z[s:s + chunk]
        d = np.zeros((zz.shape[0], e.shape[0]), dtype=np.float64)
        for j in range(z.shape[1]):  # sequential over features: identical rows give identical sums
            diff = zz[:, j:j + 1] - e[None, :, j]
            d += diff * diff
        out[s:s + chunk] = d.argmin(axis=1)
    return out


def vq_forward(z: torch.Tensor, codebook: torch.Tensor, beta: float = 0.25,
               idx: Optional[torch.Tensor] = None):
    """Standard VQ-VAE quantizer on row vectors z [N,d], codebook [K,d].

    Returns (z_st, vq_loss, perplexity, idx, l_codebook, l_commit) with
    ``z_st = z + (z_q - z).detach()`` (straight_through: true,
    frl_model_v0.yaml:35), ``vq_loss = L_codebook + beta * L_commit``
    (commitment_cost 0.25, frl_model_v0.yaml:34; terms [commitment, codebook],
    frl_bindings_v0.yaml:887-891) and ``perplexity = exp(-sum p log(p+1e-10))``
    (model returns vq_loss, perplexity: scripts/train_vqvae.py:287).
    """
    if idx is None:
        idx = torch.from_numpy(vq_argmin_np(z.detach().cpu().double().numpy(),
                                            codebook.detach().cpu().double().numpy()))
    zq = codebook[idx]
    l_codebook = ((z.detach() - zq) ** 2).mean()
    l_commit = ((z - zq.detach()) ** 2).mean()
    vq_loss = l_codebook + beta * l_commit
    z_st = z + (zq - z).detach()
    counts = torch.bincount(idx, minlength=codebook.shape[0]).to(z.dtype)
    p = counts / counts.sum()
    perplexity = torch.exp(-(p * torch.log(p + 1e-10)).sum())
    return z_st, vq_loss, perplexity, idx, l_codebook, l_commit


def vq_ema_update(codebook, ema_count, ema_sum, z, idx, decay: float = 0.99, eps: float = 1e-5):
    """EMA codebook update (quantizer 'ema', scripts/train_vqvae.py:412-414: decay .99, eps 1e-5).

    N_k <- g N_k + (1-g) n_k ; m_k <- g m_k + (1-g) sum_{n:idx=k} z_n ;
    e_k = m_k / ((N_k + eps) / (sum N + K eps) * sum N)   (Laplace smoothing).
    """
    k = codebook.shape[0]
    onehot_cnt = torch.bincount(idx, minlength=k).to(z.dtype)
    zsum = torch.zeros_like(codebook).index_add_(0, idx, z)
    new_count = decay * ema_count + (1 - decay) * onehot_cnt
    new_sum = decay * ema_sum + (1 - decay) * zsum
    n = new_count.sum()
    smoothed = (new_count + eps) / (n + k * eps) * n
    return new_sum / smoothed.unsqueeze(1), new_count, new_sum


# ----------------------------------------------------------------------------
# Decoder + reconstruction loss (build definition -- parity unpinned)
# ----------------------------------------------------------------------------
def decoder_forward(state: State, z: torch.Tensor, prefix: str) -> torch.Tensor:
    """conv1x1(d->hidden)+b -> ReLU -> conv1x1(hidden->F)+b on NCHW input.

    Template: Conv2DHead (frl/models/heads.py:128-198) with channels=[hidden],
    kernel 1, activation 'none'; state keys ``{prefix}layers.{0,2}.{weight,bias}``.
    """
    h = F.relu(F.conv2d(z, state[prefix + "layers.0.weight"], state[prefix + "layers.0.bias"]))
    return F.conv2d(h, state[prefix + "layers.2.weight"], state[prefix + "layers.2.bias"])


def reconstruction_loss_l2(pred: torch.Tensor, target: torch.Tensor,
                           mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Masked mean squared error: frl/losses/reconstruction.py:95-139 (loss_type 'l2', mean).

    mask [B,H,W] (True = valid) broadcasts over channels (:116-123); all-False -> 0 (:131-133).
    """
    loss = (pred - target) ** 2
    if mask is None:
        return loss.mean()
    while mask.dim() < loss.dim():
        mask = mask.unsqueeze(1)
    mask = mask.expand_as(loss)
    valid = loss[mask]
    if valid.numel() == 0:
        return torch.zeros((), dtype=pred.dtype)
    return valid.mean()


# ----------------------------------------------------------------------------
# Full VQ-VAE step on (time, y, x, feature) tiles
# ----------------------------------------------------------------------------
def tile_to_inputs(tile: torch.Tensor):
    """tile [B,T,H,W,F] -> x_type [B,F,H,W] = mean over time, x_phase [B,F,T,H,W].

    Build decision recorded in SURVEY.md section 8 preamble (the reference feeds
    separate static / annual groups: frl/config/frl_binding_v1.yaml:228-268,337-352).
    """
    x_phase = tile.permute(0, 4, 1, 2, 3)
    x_type = tile.mean(dim=1).permute(0, 3, 1, 2)
    return x_type, x_phase


def vqvae_forward(state: State, tile: torch.Tensor, hp: dict):
    """Full forward: tiles -> losses.  Returns a dict of tensors.

    loss = lambda_recon * (L_type + L_phase) + lambda_vq * (L_codebook + beta * L_commit)
    (loss total per SURVEY.md section 8c; lambda_vq flag scripts/train_vqvae.py:433-456).
    Phase path is conditioned on stopgrad(z_type_cont) (frl_model_v0.yaml:255,275;
    representation.py:350-351).
    """
    x_type, x_phase = tile_to_inputs(tile)
    z_type, gate, h = model_forward(state, x_type, hp)
    b, d, hh, ww = z_type.shape
    zrows = z_type.permute(0, 2, 3, 1).reshape(-1, d)
    z_st, vq_loss, perp, idx, l_cb, l_cm = vq_forward(zrows, state["quant.codebook"], hp.get("beta", 0.25))
    zq = z_st.reshape(b, hh, ww, d).permute(0, 3, 1, 2)
    xhat_type = decoder_forward(state, zq, "decoder_type.")
    l_type = reconstruction_loss_l2(xhat_type, x_type)
    out = dict(z_type=z_type, gate=gate, h=h, idx=idx, vq_loss=vq_loss, perplexity=perp,
               l_codebook=l_cb, l_commit=l_cm, xhat_type=xhat_type, l_type=l_type)
    loss = hp.get("lambda_recon", 1.0) * l_type + hp.get("lambda_vq", 1.0) * vq_loss
    if hp.get("phase", True):
        z_phase = model_forward_phase(state, x_phase, z_type.detach(), hp)
        bb, zp, t, _, _ = z_phase.shape
        zp_in = z_phase
        if "quant_phase.codebook" in state:
            prow = z_phase.permute(0, 2, 3, 4, 1).reshape(-1, zp)
            p_st, pvq, pperp, pidx, _, _ = vq_forward(prow, state["quant_phase.codebook"], hp.get("beta", 0.25))
            zp_in = p_st.reshape(bb, t, hh, ww, zp).permute(0, 4, 1, 2, 3)
            loss = loss + hp.get("lambda_vq", 1.0) * pvq
            out.update(idx_phase=pidx, vq_loss_phase=pvq, perplexity_phase=pperp)
        flat = zp_in.permute(0, 2, 1, 3, 4).reshape(bb * t, zp, hh, ww)
        xhat_phase = decoder_forward(state, flat, "decoder_phase.")
        f = xhat_phase.shape[1]
        xhat_phase = xhat_phase.reshape(bb, t, f, hh, ww).permute(0, 2, 1, 3, 4)
        l_phase = reconstruction_loss_l2(xhat_phase, x_phase)
        loss = loss + hp.get("lambda_recon", 1.0) * l_phase
        out.update(z_phase=z_phase, xhat_phase=xhat_phase, l_phase=l_phase)
    out["loss"] = loss
    return out


def init_extra_state(state: State, in_features: int, d: int, zp: int, k: int,
                     hidden: int = 128, seed: int = 0, k_phase: int = 0,
                     dtype=torch.float32) -> State:
    """Adds quantizer + decoder parameters (torch default conv init; codebook U(-1/K, 1/K))."""
    g = torch.Generator().manual_seed(seed)

    def conv_init(co, ci):
        bound = 1.0 / math.sqrt(ci)
        w = (torch.rand(co, ci, 1, 1, generator=g, dtype=torch.float64) * 2 - 1) * bound
        bvec = (torch.rand(co, generator=g, dtype=torch.float64) * 2 - 1) * bound
        return w.to(dtype), bvec.to(dtype)

    state = dict(state)
    state["quant.codebook"] = ((torch.rand(k, d, generator=g, dtype=torch.float64) * 2 - 1) / k).to(dtype)
    for name, cin in (("decoder_type.", d), ("decoder_phase.", zp)):
        state[name + "layers.0.weight"], state[name + "layers.0.bias"] = conv_init(hidden, cin)
        state[name + "layers.2.weight"], state[name + "layers.2.bias"] = conv_init(in_features, hidden)
    if k_phase:
        state["quant_phase.codebook"] = ((torch.rand(k_phase, zp, generator=g, dtype=torch.float64) * 2 - 1)
                                         / k_phase).to(dtype)
    return state


FIXED_BUFFERS = ("spatial_conv.bank", "spatial_conv.sobel_x", "spatial_conv.sobel_y")


def trainable_keys(state: State):
    return [k for k in state if k not in FIXED_BUFFERS]


def vqvae_loss_and_grads(state: State, tile: torch.Tensor, hp: dict):
    """Runs forward + backward with CPU autograd; returns (outputs, grads by key)."""
    leaf = {k: (v.detach().clone().requires_grad_(k not in FIXED_BUFFERS)) for k, v in state.items()}
    out = vqvae_forward(leaf, tile, hp)
    out["loss"].backward()
    grads = {k: v.grad.detach().clone() for k, v in leaf.items() if v.grad is not None}
    return {k: (v.detach() if torch.is_tensor(v) else v) for k, v in out.items()}, grads


# ----------------------------------------------------------------------------
# Train-step skeleton (reference: step.py:121-125,1076-1090; loops.py:97-110)
# ----------------------------------------------------------------------------
def cosine_lr(step: int, total_steps: int, lr: float, min_lr: float) -> float:
    """lr_at(step): cosine lr -> min_lr (scripts/train_vqvae.py:250-253; vae_v0.yaml:13-19)."""
    prog = min(step / max(total_steps, 1), 1.0)
    return min_lr + (lr - min_lr) * 0.5 * (1.0 + math.cos(math.pi * prog))


def beta_schedule(epoch: int, cfg: dict) -> float:
    """Linear beta ramp (configs/vae_v0.yaml:21-27)."""
    if not cfg.get("enabled", False):
        return float(cfg.get("end_value", 1.0))
    s, e = cfg["start_epoch"], cfg["end_epoch"]
    if epoch <= s:
        return float(cfg["start_value"])
    if epoch >= e:
        return float(cfg["end_value"])
    return float(cfg["start_value"] + (cfg["end_value"] - cfg["start_value"]) * (epoch - s) / (e - s))


class OracleTrainer:
    """zero_grad -> fwd -> loss -> isfinite guard -> backward -> clip(1.0) -> AdamW -> lr step.

    AdamW groups follow scripts/train_vqvae.py:221-228 (codebook: no weight decay,
    betas (0.9, 0.95)).
    """

    def __init__(self, state: State, hp: dict, lr=1e-4, weight_decay=0.01, max_norm=1.0,
                 total_steps=1000, min_lr=1e-6):
        self.hp = hp
        self.params = {k: v.detach().clone().requires_grad_(True) for k, v in state.items()
                       if k not in FIXED_BUFFERS}
        self.buffers = {k: v.detach().clone() for k, v in state.items() if k in FIXED_BUFFERS}
        cb = [v for k, v in self.params.items() if "quant" in k and "codebook" in k]
        rest = [v for k, v in self.params.items() if not ("quant" in k and "codebook" in k)]
        self.opt = torch.optim.AdamW([
            {"params": rest, "weight_decay": weight_decay},
            {"params": cb, "weight_decay": 0.0}], lr=lr, betas=(0.9, 0.95))
        self.lr, self.min_lr, self.total_steps, self.max_norm = lr, min_lr, total_steps, max_norm
        self.step_idx = 0

    def step(self, tile: torch.Tensor):
        for g in self.opt.param_groups:
            g["lr"] = cosine_lr(self.step_idx, self.total_steps, self.lr, self.min_lr)
        self.opt.zero_grad(set_to_none=True)
        out = vqvae_forward({**self.params, **self.buffers}, tile, self.hp)
        loss = out["loss"]
        if not torch.isfinite(loss):
            return out
        loss.backward()
        torch.nn.utils.clip_grad_norm_(list(self.params.values()), self.max_norm)
        self.opt.step()
        self.step_idx += 1
        return out


# ---------------------------------------------------------------------------------------------------------------
# Tile ingest: per-channel normalisation + masking of raw (time, y, x, feature) rows.
# **Parity unpinned by the reference**: its FeatureBuilder / Normalizer modules import `zarr` (absent in this image), so they
# cannot be run here and the reference holds no fixture for them; this is a restatement from the source text, channel by
# channel in the reference's own numpy form, against which the one-pass device kernel is compared bit for bit.
# ---------------------------------------------------------------------------------------------------------------
def normalize_channel_np(data: np.ndarray, preset: dict, stats: Optional[dict]) -> np.ndarray:
    """One channel, float32 numpy.  Follows FeatureBuilder._normalize_array (frl/data/loaders/builders/feature_builder.py:487-548);
    'minmax' follows MinMaxNormalizer (frl/data/normalization/normalization.py:172-201)."""
    stats = stats or {}
    kind = (preset or {}).get("type", "identity")
    out = data.copy()
    if kind == "zscore":                                            # feature_builder.py:504-509
        mean, sd = stats.get("mean", 0.0), stats.get("sd", 1.0)
        if sd < 1e-8:
            sd = 1.0
        out = (data - np.float32(mean)) / np.float32(sd)
    elif kind == "robust_iqr":                                      # :511-518
        q25, q50, q75 = stats.get("q25", 0.0), stats.get("q50", 0.0), stats.get("q75", 1.0)
        iqr = q75 - q25
        if iqr < 1e-8:
            iqr = 1.0
        out = (data - np.float32(q50)) / np.float32(iqr)
    elif kind == "minmax":                                          # normalization.py:179-199
        if preset.get("min") is not None and preset.get("max") is not None:
            lo, hi = preset["min"], preset["max"]
        else:
            lo, hi = stats["min"], stats["max"]
        rng = hi - lo
        if not rng > 1e-8:
            rng = 1.0
        out = (data - np.float32(lo)) / np.float32(rng)
    elif kind == "linear_rescale":                                  # feature_builder.py:520-531
        in_min = preset.get("in_min") if preset.get("in_min") is not None else 0.0
        in_max = preset.get("in_max") if preset.get("in_max") is not None else 1.0
        out_min = preset.get("out_min") if preset.get("out_min") is not None else 0.0
        out_max = preset.get("out_max") if preset.get("out_max") is not None else 1.0
        in_range = in_max - in_min
        if in_range < 1e-8:
            in_range = 1.0
        out_range = out_max - out_min
        out = ((data - np.float32(in_min)) / np.float32(in_range)) * np.float32(out_range) + np.float32(out_min)
    elif kind not in ("clamp", "none", "identity"):
        raise ValueError(kind)
    clamp = (preset or {}).get("clamp")
    if clamp and clamp.get("enabled", False):                       # :539-546
        cmin, cmax = clamp.get("min"), clamp.get("max")
        if cmin is not None or cmax is not None:
            out = np.clip(out, None if cmin is None else np.float32(cmin), None if cmax is None else np.float32(cmax))
    return out.astype(np.float32)


def normalize_tiles_np(raw: np.ndarray, valid: Optional[np.ndarray], presets: list, stats: list):
    """raw [..., F] (float16 | float32; NaN = no data), valid [...] or None -> (normalised float32 [..., F], mask uint8 [...]).
    Mask = given validity AND all features finite; invalid rows are zeroed AFTER normalisation
    (FeatureBuilder.build_feature order, feature_builder.py:160-170; _apply_mask_to_data :709-737)."""
    x = raw.astype(np.float32)
    ok = np.isfinite(x).all(axis=-1)
    if valid is not None:
        ok &= valid.astype(bool)
    out = np.empty_like(x)
    with np.errstate(invalid="ignore", over="ignore"):
        for c in range(x.shape[-1]):
            out[..., c] = normalize_channel_np(x[..., c], presets[c], stats[c])
    out = np.where(ok[..., None], out, np.float32(0.0))
    return out, ok.astype(np.uint8)


# ---------------------------------------------------------------------------------------------------------------
# Dead-code revival (CodebookManager; the reference constructs it at scripts/train_vqvae.py:196-198 but does not ship the module:
# build definition, **parity unpinned**).  Bit-level mirror of frl_vq_revive_dead_codes.
# ---------------------------------------------------------------------------------------------------------------
def splitmix64(x: int) -> int:
    m = (1 << 64) - 1
    x = (x + 0x9E3779B97F4A7C15) & m
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & m
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & m
    return x ^ (x >> 31)


def revive_dead_codes_np(codebook: np.ndarray, window_counts: np.ndarray, min_count: int, z_rows: np.ndarray, seed: int):
    """-> (new codebook float32, boolean dead mask).  Code k with window_counts[k] < min_count takes z_rows[splitmix64(seed + k) % N]."""
    out = codebook.astype(np.float32).copy()
    dead = window_counts < min_count
    n = z_rows.shape[0]
    for k in np.nonzero(dead)[0]:
        out[k] = z_rows[splitmix64((seed + int(k)) & ((1 << 64) - 1)) % n].astype(np.float32)
    return out, dead


# ---------------------------------------------------------------------------------------------------------------
# Mutual kNN pair mining -- restatement of pairs_mutual_knn_chunked (frl/losses/pairs.py:531-610) in float64 numpy with the
# (distance, index) tie rule.  **Pinned**: tests/golden/mutual_knn_*.npz hold the reference function's own output
# (oracle/make_pairs_golden.py imports it); the restatement equals it on those inputs.
# ---------------------------------------------------------------------------------------------------------------
def mutual_knn_pairs_np(features: np.ndarray, coord_list: list, offsets: list, k: int, pos_min_spatial: float = 4.0):
    """-> (pairs [P, 2] int64 ordered by anchor then neighbour rank, knn_idx [N, k] int64 with -1 padding)."""
    x = features.astype(np.float64)
    n = x.shape[0]
    d2 = ((x[:, None, :] - x[None, :, :]) ** 2).sum(-1)
    np.fill_diagonal(d2, np.inf)                                                  # pairs.py:573-575
    for p, cp in enumerate(coord_list):                                           # pairs.py:577-588
        ps, pe = offsets[p], offsets[p + 1]
        c = np.asarray(cp, dtype=np.float32).reshape(-1, 2)
        sp = np.sqrt(((c[:, None, :] - c[None, :, :]) ** 2).sum(-1, dtype=np.float32))
        blk = d2[ps:pe, ps:pe]
        blk[sp < np.float32(pos_min_spatial)] = np.inf
    knn = np.full((n, k), -1, dtype=np.int64)
    kk = min(k, n - 1)
    for i in range(n):                                                            # pairs.py:590-594
        order = np.lexsort((np.arange(n), d2[i]))[:kk]
        order = np.where(np.isinf(d2[i][order]), -1, order)
        knn[i, :kk] = order
    pairs = [(i, int(j)) for i in range(n) for j in knn[i] if j >= 0 and (knn[j] == i).any()]     # pairs.py:596-610
    return np.asarray(pairs, dtype=np.int64).reshape(-1, 2), knn


# ---------------------------------------------------------------------------------------------------------------------------
# InfoNCE over mined pairs + location gather (frl/losses/contrastive.py:29-212, frl/utils/spatial.py:132-173)
# ---------------------------------------------------------------------------------------------------------------------------
def contrastive_loss_oracle(embeddings: torch.Tensor, pos_pairs: torch.Tensor, neg_pairs: torch.Tensor, pos_weights=None,
                            neg_weights=None, temperature: float = 0.07, similarity: str = "l2") -> torch.Tensor:
    """Anchor by anchor in plain loops (float64 when the embeddings are): for every anchor a with at least one positive

        L_a = -log(sum_p w_p exp(l_p - m) + 1e-8) + log(sum_{p,n} w exp(l - m) + 1e-8),   l = sim / t,  m = max logit of the anchor

    which is what the reference's scatter_reduce('amax') / scatter_add formulation evaluates (contrastive.py:171-209); loss = mean
    over those anchors; negatives of anchors without a positive are ignored (:160-170); empty pos_pairs -> 0 (:104-105)."""
    if pos_pairs.numel() == 0:
        return torch.zeros((), dtype=embeddings.dtype)
    dim = embeddings.shape[1]

    def sim(a, b):
        if similarity == "l2":
            return -((a - b) ** 2).sum() / dim
        if similarity == "dot":
            return (a * b).sum()
        if similarity == "cosine":
            return (a / a.norm().clamp_min(1e-12) * (b / b.norm().clamp_min(1e-12))).sum()
        raise ValueError(f"Unknown similarity function: {similarity}")

    pw = torch.ones(pos_pairs.shape[0], dtype=embeddings.dtype) if pos_weights is None else pos_weights.to(embeddings.dtype)
    nw = torch.ones(neg_pairs.shape[0], dtype=embeddings.dtype) if neg_weights is None else neg_weights.to(embeddings.dtype)
    losses = []
    for a in sorted(set(pos_pairs[:, 0].tolist())):
        pos = [torch.log(pw[i]) + sim(embeddings[a], embeddings[int(pos_pairs[i, 1])]) / temperature
               for i in range(pos_pairs.shape[0]) if int(pos_pairs[i, 0]) == a]
        neg = [torch.log(nw[i]) + sim(embeddings[a], embeddings[int(neg_pairs[i, 1])]) / temperature
               for i in range(neg_pairs.shape[0]) if int(neg_pairs[i, 0]) == a]
        pos_t, all_t = torch.stack(pos), torch.stack(pos + neg)
        m = all_t.max()     # NOT detached: the reference's scatter_reduce_('amax') is differentiable, and with eps inside the logarithms
        #                     the +-m terms do not cancel exactly (an O(1e-8) term on the anchor's arg-max logit)
        losses.append(-torch.log(torch.exp(pos_t - m).sum() + 1e-8) + torch.log(torch.exp(all_t - m).sum() + 1e-8))
    return torch.stack(losses).mean()


def extract_at_locations_np(feature: np.ndarray, coords: np.ndarray) -> np.ndarray:
    """feature [C, H, W], coords [N, 2] (row, col) -> [N, C]  (frl/utils/spatial.py:132-154)."""
    return np.stack([feature[:, int(r), int(c)] for r, c in coords], axis=0) if len(coords) else np.zeros((0, feature.shape[0]), feature.dtype)
