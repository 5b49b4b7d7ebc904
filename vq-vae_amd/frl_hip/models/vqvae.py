"""VQ-VAE over (time, y, x, feature) tiles: RepresentationModel encoder -> vector quantizer -> 1x1-conv decoders.

The reference tree no longer ships a quantizer or decoder; this module implements the build
definition of SURVEY.md section 8a rows a11/a12 behind the surviving legacy trainer contract
(scripts/train_vqvae.py:183-198,221-224,287):
  * `model.quant.codebook` is an nn.Parameter [K, emb_dim] whose qualified name contains "quant.codebook";
  * `model.quant.codebook_size`, `model.quant.emb_dim`, `model.attach_codebook_manager(m)`;
  * `model(batch) -> (cont_pred, cat_logits, canopy_pred, vq_loss, perplexity)`.
The class extends RepresentationModel so the encoder keeps the reference's state-dict keys (reference checkpoints load
with strict=False) and the extra keys are `quant.codebook`, `decoder_type.layers.{0,2}.*`, `decoder_phase.layers.{0,2}.*`.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.distributed as dist
import torch.nn as nn

from .. import functional as Fh
from .. import ops
from .blocks import Conv2DHead
from .representation import RepresentationModel


class VectorQuantizer(nn.Module):
    """argmin-L2 codebook lookup with straight-through estimator; 'st' (gradient) or 'ema' codebook updates."""

    def __init__(self, codebook_size: int = 256, emb_dim: int = 64, beta: float = 0.25, quantizer: str = "st",
                 ema_decay: float = 0.99, ema_eps: float = 1e-5):
        super().__init__()
        if quantizer not in ("st", "ema"):
            raise ValueError("quantizer must be 'st' or 'ema' (scripts/train_vqvae.py:412)")
        self.codebook_size, self.emb_dim, self.beta = codebook_size, emb_dim, beta
        self.quantizer, self.ema_decay, self.ema_eps = quantizer, ema_decay, ema_eps
        self.codebook = nn.Parameter(torch.empty(codebook_size, emb_dim).uniform_(-1.0 / codebook_size, 1.0 / codebook_size),
                                     requires_grad=(quantizer == "st"))
        if quantizer == "ema":
            self.register_buffer("ema_count", torch.zeros(codebook_size))
            self.register_buffer("ema_sum", self.codebook.detach().clone())
        self.last_counts: Optional[torch.Tensor] = None
        self.last_stats: Optional[torch.Tensor] = None
        # A trainer that guards the step against a non-finite loss sets defer_ema: the statistics of the batch are then kept until
        # apply_ema(ok) -- after the loss is known -- instead of being folded into the running averages inside forward().
        self.defer_ema = False
        self._pending_ema = None
        self._prepared = None                               # (key, image): see prepared()

    def prepared(self, dtype: torch.dtype, n_rows: int) -> torch.Tensor:
        """Prepared image of the codebook for rows of `dtype` (norms + packed fragments, `ops.vq_prepare`), rebuilt only when the
        codebook has changed: keyed on the parameter's storage and version counter (HipAdamW, the EMA update and the dead-code
        revival bump it; so does every in-place torch operation)."""
        cb = self.codebook
        key = (cb.data_ptr(), cb._version, dtype)
        if self._prepared is None or self._prepared[0] != key:
            buf = self._prepared[1] if self._prepared is not None else None
            self._prepared = (key, ops.vq_prepare(cb.detach(), n_rows, dtype, out=buf))
        return self._prepared[1]

    def forward(self, z_rows: torch.Tensor, raw_terms: bool = False):
        """z_rows [N, d] -> (z_q [N,d] with straight-through gradient, vq_loss, perplexity, idx int32 [N]).  raw_terms (gradient quantizer
        only): vq_loss is returned as its parts ((L_codebook, 1.0), (L_commit, beta)) for a caller that folds them into its own weighted
        sum of loss terms (VQVAE.forward_tiles: one launch for the total AND the reported vq_loss instead of two)."""
        zq, l_cb, l_cm, perp, idx, counts, stats = Fh.VQFn.apply(z_rows, self.codebook, self.prepared(z_rows.dtype, z_rows.shape[0]))
        self.last_counts = counts
        self.last_stats = stats          # f32 [4]: sum ||z - z_q||^2, perplexity, rows whose arg-min was re-evaluated exactly, mean squared error
        if self.quantizer == "ema":
            vq_loss = Fh.scalar_combine([l_cm], [self.beta])[0]
            if self.training:
                with torch.no_grad():
                    _, _, sums = ops.vq_bwd(None, z_rows.detach(), self.codebook.detach(), idx, counts, None, 0.0,
                                            want_gz=False, want_ge=False, want_sums=True)
                self._pending_ema = (sums, counts)
                if not self.defer_ema:
                    self.apply_ema()
        elif raw_terms:
            vq_loss = ((l_cb, 1.0), (l_cm, float(self.beta)))
        else:
            vq_loss = Fh.scalar_combine([l_cb, l_cm], [1.0, self.beta])[0]       # one launch (and one in the backward)
        return zq, vq_loss, perp, idx

    @torch.no_grad()
    def apply_ema(self, ok: Optional[torch.Tensor] = None) -> None:
        """Folds the batch statistics kept by forward() into the running averages and rewrites the codebook.  Data parallel: the
        per-code counts and sums are summed over the ranks first (the codebook is no parameter here, so it is in no gradient bucket:
        without this every rank would drift to its own codebook).  `ok` (device float [1], <= 0 = non-finite loss) gates the update
        on the device; pass ok=False-like host values by simply not calling this and using drop_ema()."""
        if self._pending_ema is None:
            return
        sums, counts = self._pending_ema
        self._pending_ema = None
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            counts = counts.clone()                                        # last_counts keeps this rank's own batch
            dist.all_reduce(sums)
            dist.all_reduce(counts)
        ops.vq_ema_update(sums, counts, self.ema_count, self.ema_sum, self.codebook.data, self.ema_decay, self.ema_eps, ok)
        if self._prepared is not None:                                     # the cached fragment image belongs to the old codebook
            self._prepared = ((None, None, self._prepared[0][2]), self._prepared[1])   # (buffer kept: captured graphs hold its address)
        if self.defer_ema:                                                 # (inside forward() the autograd graph still holds the codebook)
            torch.autograd.graph.increment_version(self.codebook)

    def drop_ema(self) -> None:
        self._pending_ema = None


class VQVAE(RepresentationModel):
    """Encoder -> VQ(z_type) -> type decoder ; phase path conditioned on stopgrad(z_type_cont) -> phase decoder."""

    def __init__(self, in_features: Optional[int] = None, codebook_size: int = 256, emb_dim: int = 64, beta: float = 0.25,
                 hidden: int = 128, quantizer: str = "st", ema_decay: float = 0.99, ema_eps: float = 1e-5,
                 phase: bool = True, phase_codebook_size: int = 0, lambda_recon: float = 1.0, lambda_vq: float = 1.0,
                 cont_dim: Optional[int] = None, cat_vocab_sizes=None, naip_bands: Optional[int] = None, cat_emb_dim: int = 8,
                 **repr_kwargs):
        # Legacy constructor call (scripts/train_vqvae.py:183-195): VQVAE(cont_dim=, cat_vocab_sizes=, naip_bands=, emb_dim=,
        # codebook_size=, beta=, hidden=, quantizer=, cat_emb_dim=, ema_decay=, ema_eps=).  `cont_dim` is the number of continuous
        # features = the tile's feature axis (alias of in_features).  The tile VQ-VAE has no categorical embeddings and no NAIP
        # image branch (the `vqvae` package that defined them is absent from the reference tree): an empty
        # vocabulary table and any band count are accepted and recorded, a non-empty table is refused instead of being ignored.
        if cont_dim is not None:
            if in_features is not None and int(in_features) != int(cont_dim):
                raise ValueError(f"cont_dim ({cont_dim}) and in_features ({in_features}) name the same axis and disagree")
            in_features = int(cont_dim)
        if in_features is None:
            in_features = 64
        if cat_vocab_sizes:
            raise ValueError("cat_vocab_sizes is not empty: categorical inputs are not part of the (time, y, x, feature) tile VQ-VAE "
                             f"(got {sorted(dict(cat_vocab_sizes))}); pass an empty mapping")
        self_legacy = dict(cat_vocab_sizes=dict(cat_vocab_sizes or {}), naip_bands=None if naip_bands is None else int(naip_bands),
                           cat_emb_dim=int(cat_emb_dim))
        repr_kwargs.setdefault("z_type_dim", emb_dim)
        if repr_kwargs["z_type_dim"] != emb_dim:
            raise ValueError("emb_dim must equal z_type_dim (the quantizer acts on z_type)")
        if "type_encoder_channels" not in repr_kwargs and emb_dim != 64:
            repr_kwargs["type_encoder_channels"] = (128, emb_dim)
        super().__init__(type_in_channels=in_features, phase_in_channels=in_features, **repr_kwargs)
        self.in_features, self.phase, self.lambda_recon, self.lambda_vq = in_features, phase, lambda_recon, lambda_vq
        self.cont_dim, self.legacy_inputs = in_features, self_legacy
        self.quant = VectorQuantizer(codebook_size, emb_dim, beta, quantizer, ema_decay, ema_eps)
        self.decoder_type = Conv2DHead(emb_dim, [hidden], in_features)
        if phase:
            self.decoder_phase = Conv2DHead(self.z_phase_dim, [hidden], in_features)
            if phase_codebook_size:
                self.quant_phase = VectorQuantizer(phase_codebook_size, self.z_phase_dim, beta, quantizer, ema_decay, ema_eps)
        # a trainer with a per-step lambda_vq(step) schedule keeps the current value in this device scalar (float32 [1]); the loss head then
        # reads it at run time, so a captured step follows the schedule (None: the host value self.lambda_vq is baked into the launch)
        self.lambda_vq_dev: Optional[torch.Tensor] = None
        self.codebook_manager = None
        self._manager_takes_rows = False
        self.defer_codebook_hooks = False                   # set by a trainer with an isfinite guard: see commit_codebook_hooks
        self._pending_manager = None
        self.fused_decoder = True
        # The phase path is conditioned on stopgrad(z_type): forward AND backward of the two branches are independent, so the phase
        # branch runs on a side HIP stream next to VQ + type decoder (forward) and next to the whole type-path backward.
        self.concurrent_phase = True
        self._side_stream = None

    def phase_stream(self, device):
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream(device=device)
        return self._side_stream

    def attach_codebook_manager(self, manager) -> None:
        """scripts/train_vqvae.py:197-198: the manager tracks usage / dead codes from `quant.last_counts`; a manager whose
        `update` takes a second argument (training.codebook_manager.CodebookManager) also receives the encoder rows for revival."""
        import inspect
        self.codebook_manager = manager
        upd = getattr(manager, "update", None)
        self._manager_takes_rows = upd is not None and len(inspect.signature(upd).parameters) >= 2

    def _decode_loss(self, dec: Conv2DHead, z: torch.Tensor, target: torch.Tensor, mask, want_recon: bool):
        """Decoder + masked L2.  Hot configuration (bf16, hidden 128, 64 features): one fused kernel per direction, the
        reconstruction is materialised only on request; otherwise the modular conv1x1 / MSE kernels."""
        l0, l2 = dec.layers[0], dec.layers[-1]
        if (len(dec.layers) == 3 and self.fused_decoder
                and ops.decoder_mse_supported(z.shape[-1], l0.out_channels, l2.out_channels, z)):
            w1 = l0.weight.reshape(l0.out_channels, l0.in_channels)
            w2 = l2.weight.reshape(l2.out_channels, l2.in_channels)
            return Fh.decoder_mse(z, w1, l0.bias, w2, l2.bias, target, mask, want_recon)
        xhat = dec(z)
        return Fh.mse_loss(xhat, target, mask), xhat

    def _phase_branch(self, tile, z_type_detached, mask, return_recon) -> Dict[str, torch.Tensor]:
        """Dense phase path -> (optional phase codebook) -> phase decoder + masked L2; returns its outputs and `loss_terms`."""
        b, t, hh, ww, f = tile.shape
        out: Dict[str, torch.Tensor] = {}
        z_phase = self.forward_phase_nhwc(tile, z_type_detached)            # [B,T,H,W,zp]
        zp_in = z_phase
        terms = []                                                          # (loss term, weight) pairs, summed by forward_tiles in one launch
        if hasattr(self, "quant_phase"):
            zpq, pvq, pperp, pidx = self.quant_phase(z_phase.reshape(-1, z_phase.shape[-1]))
            zp_in = zpq.reshape(z_phase.shape)
            terms.append(self._vq_term(pvq))
            out.update(idx_phase=pidx, vq_loss_phase=pvq, perplexity_phase=pperp)
        if mask is None or mask.dim() == 4:                                 # [B,T,H,W]: per-observation validity from the tile ingest
            pmask = mask
        else:
            pmask = mask.unsqueeze(1).expand(b, t, hh, ww).contiguous()
        l_phase, xhat_phase = self._decode_loss(self.decoder_phase, zp_in, tile, pmask, return_recon)
        terms.append((l_phase, self.lambda_recon))
        out.update(z_phase=z_phase, l_phase=l_phase, loss_terms=terms)
        if xhat_phase is not None:
            out["xhat_phase"] = xhat_phase
        return out

    def forward_tiles(self, tile: torch.Tensor, mask: Optional[torch.Tensor] = None,
                      return_recon: bool = False, differentiable_vq_loss: bool = False) -> Dict[str, torch.Tensor]:
        """tile [B,T,H,W,F] (any float dtype, GPU) -> dict(loss, l_type, l_phase, vq_loss, perplexity, idx, ...).
        `mask` (1 = valid) is per pixel [B,H,W] or per observation [B,T,H,W] (as `TilePrefetcher` delivers it); with the latter the
        type-path loss counts a pixel only if all its time steps are valid (its input is their mean).
        `xhat_type` / `xhat_phase` are present when return_recon=True or when the modular decoder path is taken.
        out["vq_loss"] is a reported value (no gradient flows through it; out["loss"] carries the quantizer's gradients) unless
        differentiable_vq_loss=True, which the legacy contract asks for: its caller builds the total loss itself."""
        self._require_gpu(tile)
        tile = self._rows(tile)
        b, t, hh, ww, f = tile.shape
        with torch.no_grad():
            x_type = ops.mean_time(tile)                                   # [B,H,W,F]
        z_type, gate = self.forward_nhwc(x_type, return_gate=True)          # [B,H,W,d]
        side = ph = None
        if self.phase and self.concurrent_phase and tile.is_cuda:
            main, side = torch.cuda.current_stream(), self.phase_stream(tile.device)
            side.wait_stream(main)
            zt = z_type.detach()
            zt.record_stream(side)
            tile.record_stream(side)
            if mask is not None:
                mask.record_stream(side)
            with torch.cuda.stream(side):
                ph = self._phase_branch(tile, zt, mask, return_recon)
        d = z_type.shape[-1]
        # gradient quantizer on the GPU: its two loss parts join the total below directly (one launch yields the total, the flag and vq_loss)
        split_vq = self.quant.quantizer != "ema" and z_type.is_cuda and not differentiable_vq_loss
        zq, vq_loss, perp, idx = self.quant(z_type.reshape(-1, d), raw_terms=split_vq)
        vq_parts = vq_loss if split_vq else None
        tmask = mask.amin(dim=1) if (mask is not None and mask.dim() == 4) else mask
        l_type, xhat_type = self._decode_loss(self.decoder_type, zq.reshape(b, hh, ww, d), x_type, tmask, return_recon)
        out = dict(z_type=z_type, gate=gate, idx=idx, vq_loss=None if split_vq else vq_loss, perplexity=perp, l_type=l_type,
                   vq_stats=self.quant.last_stats)
        if xhat_type is not None:
            out["xhat_type"] = xhat_type
        if split_vq:
            terms = [(l_type, self.lambda_recon)]
            for part, w in vq_parts:                                    # lambda_vq * (L_codebook + beta * L_commit), term by term
                t = self._vq_term(part)
                terms.append((t[0], t[1] * w) + tuple(t[2:]))
            aux = [0.0] + [w for _, w in vq_parts]
        else:
            terms = [(l_type, self.lambda_recon), self._vq_term(vq_loss)]
            aux = None
        if self.phase:
            if side is not None:
                main.wait_stream(side)
                for v in ph.values():
                    if torch.is_tensor(v):
                        v.record_stream(main)
            else:
                ph = self._phase_branch(tile, z_type.detach(), mask, return_recon)
            terms += ph.pop("loss_terms")
            out.update(ph)
        # weighted sum of the loss terms and its isfinite flag in ONE launch (the trainer's device-side guard reads out["loss_ok"])
        res = Fh.scalar_combine([t[0] for t in terms], [float(t[1]) for t in terms], [t[2] if len(t) > 2 else None for t in terms],
                                aux_coefs=None if aux is None else aux + [0.0] * (len(terms) - len(aux)))
        loss, ok = res[0], res[1]
        if aux is not None:
            out["vq_loss"] = res[2]                                     # L_codebook + beta L_commit (reported value: no gradient flows through it)
        out["loss"] = loss
        if ok is not None:
            out["loss_ok"] = ok
        if self.codebook_manager is not None and hasattr(self.codebook_manager, "update") and self.training:
            self._pending_manager = (self.quant.last_counts, z_type.detach().reshape(-1, d))
            if not self.defer_codebook_hooks:
                self.commit_codebook_hooks()
        return out

    @torch.no_grad()
    def init_codebook_from_tiles(self, tile: torch.Tensor, seed: int = 0) -> None:
        """Data-dependent codebook initialisation: every code starts on the encoder output of a pixel drawn (seeded, without
        replacement) from `tile` [B,T,H,W,F] -- the usual remedy for the index collapse of a codebook initialised far from the
        encoder's output distribution (the legacy trainer pairs the model with a CodebookManager for the same reason,
        scripts/train_vqvae.py:196-198; the manager re-seeds dead codes with encoder rows in exactly this way during training).
        The phase codebook, when present, is initialised from z_phase rows likewise.  Runs the HIP forward path."""
        self._require_gpu(tile)
        tile = self._rows(tile)
        was_training = self.training
        self.eval()
        try:
            z_type = self.forward_nhwc(ops.mean_time(tile))
            gen = torch.Generator(device="cpu").manual_seed(int(seed))

            def draw(rows: torch.Tensor, q: "VectorQuantizer"):
                rows = rows.reshape(-1, rows.shape[-1]).float()
                if rows.shape[0] < q.codebook_size:
                    raise ValueError(f"init_codebook_from_tiles: {rows.shape[0]} encoder rows for {q.codebook_size} codes")
                pick = torch.randperm(rows.shape[0], generator=gen)[:q.codebook_size].to(rows.device)
                q.codebook.copy_(rows[pick])
                if q.quantizer == "ema":
                    q.ema_sum.copy_(q.codebook)
                    q.ema_count.fill_(1.0)

            draw(z_type, self.quant)
            if hasattr(self, "quant_phase"):
                draw(self.forward_phase_nhwc(tile, z_type), self.quant_phase)
        finally:
            self.train(was_training)

    def _vq_term(self, vq_loss):
        """(term, host weight[, device multiplier]) of a quantizer loss in the total: lambda_vq from the device scalar when a trainer set one"""
        if self.lambda_vq_dev is not None and vq_loss.is_cuda:
            return (vq_loss, 1.0, self.lambda_vq_dev)
        return (vq_loss, self.lambda_vq)

    def _quantizers(self):
        return [q for q in (getattr(self, "quant", None), getattr(self, "quant_phase", None)) if q is not None]

    def set_defer_codebook_hooks(self, on: bool) -> None:
        """on: forward_tiles no longer mutates codebook state (EMA running averages, the manager's usage window); the caller commits it
        with commit_codebook_hooks(ok) once it knows whether the batch counts (the reference skips a batch with a non-finite loss,
        frl/training/representation/step.py:1057-1074) or discards it with drop_codebook_hooks()."""
        self.defer_codebook_hooks = bool(on)
        for qz in self._quantizers():
            qz.defer_ema = bool(on)

    def commit_codebook_hooks(self, ok: Optional[torch.Tensor] = None) -> None:
        for qz in self._quantizers():
            if qz.quantizer == "ema":
                qz.apply_ema(ok)
        pend, self._pending_manager = self._pending_manager, None
        if pend is not None and self.codebook_manager is not None:
            counts, rows = pend
            if self._manager_takes_rows:
                if ok is not None:
                    self.codebook_manager.update(counts, rows, ok=ok)
                else:
                    self.codebook_manager.update(counts, rows)
            else:                                                           # a manager with the bare update(counts) signature
                self.codebook_manager.update(counts if ok is None else counts * (ok > 0).to(counts.dtype))

    def drop_codebook_hooks(self) -> None:
        self._pending_manager = None
        for qz in self._quantizers():
            qz.drop_ema()

    def forward(self, batch, return_gate: bool = False):
        """Legacy contract (scripts/train_vqvae.py:287): dict batch -> (cont_pred, cat_logits, canopy_pred, vq_loss, perplexity).

        A tensor argument keeps the RepresentationModel.forward semantics ([B,C,H,W] -> z_type)."""
        if isinstance(batch, dict):
            out = self.forward_tiles(batch["tile"], batch.get("mask"), return_recon=True, differentiable_vq_loss=True)
            return out["xhat_type"], {}, out.get("xhat_phase"), out["vq_loss"], out["perplexity"]
        return super().forward(batch, return_gate)
