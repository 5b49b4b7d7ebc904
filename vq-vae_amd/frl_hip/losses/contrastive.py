"""InfoNCE over mined pairs on the HIP path: `contrastive_loss` with the reference's signature and semantics
(frl/losses/contrastive.py:29-212; callers frl/training/representation/step.py:563,787).

    L_a = -log( sum_p w_p exp(sim(a,p)/t) / ( sum_p w_p exp(sim(a,p)/t) + sum_n w_n exp(sim(a,n)/t) ) ),  mean over anchors with a positive

with the reference's stabilised evaluation (per-anchor maximum, eps = 1e-8 inside both logarithms), similarities l2 | cosine | dot,
negatives of anchors without a positive ignored, empty `pos_pairs` -> 0.  Where the reference groups pairs with scatter_reduce /
scatter_add (float atomics, order varies run to run), the pairs are sorted by anchor here (torch.sort, stable -- index plumbing) and
every segment is reduced by one wave in a fixed order; the gradient rows are folded into d(embeddings) with the same sorted-segment
primitive.  Loss and gradients are bit-reproducible.
"""
from __future__ import annotations

from typing import Literal, Optional

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import ops

_SIM = {"l2": 0, "cosine": 1, "dot": 2}


class _InfoNCEFn(Function):
    @staticmethod
    def forward(ctx, emb, pairs, weights, is_pos, seg, temperature, sim):
        loss, sims, coef = ops.infonce_fwd(emb, pairs, weights, is_pos, seg, temperature, sim, want_coef=emb.requires_grad)
        ctx.save_for_backward(emb, pairs, sims, coef)
        ctx.nseg, ctx.temperature, ctx.sim = seg.numel() - 1, temperature, sim
        return loss.reshape(())

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        emb, pairs, sims, coef = ctx.saved_tensors
        ga, gb = ops.infonce_pair_grads(emb, pairs, sims, coef, g.reshape(1).float().contiguous(), ctx.nseg, ctx.temperature, ctx.sim)
        de = torch.zeros_like(emb)
        ops.segment_sum_rows(ga, None, pairs[:, 0].contiguous(), de)                       # already sorted by anchor
        keys, order = torch.sort(pairs[:, 1], stable=True)
        ops.segment_sum_rows(gb, order, keys, de, accumulate=True)
        return de, None, None, None, None, None, None


def contrastive_loss(embeddings: torch.Tensor, pos_pairs: torch.Tensor, neg_pairs: torch.Tensor,
                     pos_weights: Optional[torch.Tensor] = None, neg_weights: Optional[torch.Tensor] = None,
                     temperature: float = 0.07, similarity: Literal["l2", "cosine", "dot"] = "l2") -> torch.Tensor:
    if similarity not in _SIM:
        raise ValueError(f"Unknown similarity function: {similarity}")
    dev = embeddings.device
    if pos_pairs.numel() == 0:
        return torch.tensor(0.0, device=dev, dtype=embeddings.dtype)
    emb = embeddings if embeddings.dtype == torch.float32 else embeddings.float()
    emb = emb.contiguous()
    # (the reference's emb[pairs[:, k]] raises on an out-of-range row; here such rows are wrapped / clamped and flagged: ops.index_errors)
    pos_pairs = ops.sanitize_indices(pos_pairs.to(dev, torch.int64).reshape(-1, 2), emb.shape[0], "contrastive_loss: pos_pairs")
    neg_pairs = ops.sanitize_indices(neg_pairs.to(dev, torch.int64).reshape(-1, 2), emb.shape[0], "contrastive_loss: neg_pairs")
    pw = torch.ones(pos_pairs.shape[0], device=dev) if pos_weights is None else pos_weights.to(dev, torch.float32)
    nw = torch.ones(neg_pairs.shape[0], device=dev) if neg_weights is None else neg_weights.to(dev, torch.float32)
    # negatives count only for anchors that have a positive (contrastive.py:160-170)
    has_pos = torch.zeros(emb.shape[0], dtype=torch.bool, device=dev)
    has_pos[pos_pairs[:, 0]] = True
    keep = has_pos[neg_pairs[:, 0]]
    neg_pairs, nw = neg_pairs[keep], nw[keep]
    pairs = torch.cat([pos_pairs, neg_pairs], dim=0)
    weights = torch.cat([pw, nw])
    is_pos = torch.cat([torch.ones(pos_pairs.shape[0], dtype=torch.uint8, device=dev), torch.zeros(neg_pairs.shape[0], dtype=torch.uint8, device=dev)])
    anchors, order = torch.sort(pairs[:, 0], stable=True)          # positives of an anchor stay ahead of its negatives, each in list order
    pairs, weights, is_pos = pairs[order].contiguous(), weights[order].contiguous(), is_pos[order].contiguous()
    uniq, counts = torch.unique_consecutive(anchors, return_counts=True)
    seg = torch.zeros(uniq.numel() + 1, dtype=torch.int64, device=dev)
    seg[1:] = torch.cumsum(counts, 0)
    loss = _InfoNCEFn.apply(emb, pairs, weights, is_pos, seg, float(temperature), _SIM[similarity])
    return loss.to(embeddings.dtype)
