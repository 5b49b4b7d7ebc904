"""GPU parity of the InfoNCE loss over mined pairs and of the sparse-location gathers against fixtures written
by the REFERENCE's own functions (oracle/make_contrastive_golden.py: frl/losses/contrastive.py, frl/utils/spatial.py)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _fx(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


@pytest.mark.parametrize("case", ["a", "b", "c", "d"])
def test_contrastive_loss_matches_reference(golden_dir, case):
    from frl_hip.losses import contrastive_loss
    fx = _fx(golden_dir, f"contrastive_{case}.npz")
    emb = torch.from_numpy(fx["emb"]).float().to(DEV).requires_grad_(True)
    pos, neg = torch.from_numpy(fx["pos"]).to(DEV), torch.from_numpy(fx["neg"]).to(DEV)
    pw = torch.from_numpy(fx["pw"]).float().to(DEV) if "pw" in fx.files else None
    nw = torch.from_numpy(fx["nw"]).float().to(DEV) if "nw" in fx.files else None
    loss = contrastive_loss(emb, pos, neg, pw, nw, temperature=float(fx["t"]), similarity=str(fx["sim"]))
    assert abs(loss.item() - float(fx["loss64"])) <= 2e-6 * max(1.0, abs(float(fx["loss64"])))          # f32 evaluation vs float64 reference
    loss.backward()
    g = emb.grad.cpu().numpy()
    scale = max(1e-6, np.abs(fx["grad64"]).max())
    assert np.abs(g - fx["grad64"]).max() <= 1e-5 * scale
    # bit-reproducible (sorted segments, fixed-order sums): a second evaluation gives identical bits, loss and gradient
    emb2 = torch.from_numpy(fx["emb"]).float().to(DEV).requires_grad_(True)
    loss2 = contrastive_loss(emb2, pos, neg, pw, nw, temperature=float(fx["t"]), similarity=str(fx["sim"]))
    loss2.backward()
    assert loss2.item() == loss.item() and torch.equal(emb2.grad, emb.grad)


def test_contrastive_loss_edge_cases():
    from frl_hip.losses import contrastive_loss
    emb = torch.randn(10, 8, device=DEV)
    empty = torch.empty(0, 2, dtype=torch.long, device=DEV)
    assert contrastive_loss(emb, empty, torch.tensor([[0, 1]], device=DEV)).item() == 0.0                  # no positives: 0
    # positives but no negatives at all: nothing to contrast, loss ~ 0 (eps inside the logarithms only)
    l = contrastive_loss(emb, torch.tensor([[0, 1], [2, 3]], device=DEV), empty)
    assert abs(l.item()) < 1e-6
    with pytest.raises(ValueError):
        contrastive_loss(emb, torch.tensor([[0, 1]], device=DEV), empty, similarity="manhattan")
    # a very low temperature with a well separated positive: loss -> 0
    e = torch.zeros(3, 4, device=DEV)
    e[1, 0] = 0.01
    e[2, 0] = 5.0
    l = contrastive_loss(e, torch.tensor([[0, 1]], device=DEV), torch.tensor([[0, 2]], device=DEV), temperature=0.01)
    assert l.item() < 1e-6


def test_extract_at_locations_matches_reference(golden_dir):
    from frl_hip.utils import extract_at_locations, extract_temporal_at_locations
    fx = _fx(golden_dir, "extract_locations.npz")
    coords = torch.from_numpy(fx["coords"]).to(DEV)
    feat = torch.from_numpy(fx["feat"]).to(DEV)                                   # reference layout [C, H, W], contiguous
    assert np.array_equal(extract_at_locations(feat, coords).cpu().numpy(), fx["out"])
    nhwc = feat.permute(1, 2, 0).contiguous()                                     # this library's rows; the [C, H, W] VIEW is gathered in place
    assert np.array_equal(extract_at_locations(nhwc.permute(2, 0, 1), coords).cpu().numpy(), fx["out"])
    bf = nhwc.to(torch.bfloat16).permute(2, 0, 1)
    assert torch.equal(extract_at_locations(bf, coords).float().cpu(), torch.from_numpy(fx["out"]).to(torch.bfloat16).float())
    assert np.array_equal(extract_temporal_at_locations(torch.from_numpy(fx["feat_t"]).to(DEV), coords).cpu().numpy(), fx["out_t"])
    # backward: rows that address the same pixel add up (coords[7] == coords[3])
    fa = nhwc.permute(2, 0, 1).detach().requires_grad_(True)
    w = torch.from_numpy(fx["w"]).to(DEV)
    (extract_at_locations(fa, coords) * w).sum().backward()
    assert np.abs(fa.grad.cpu().numpy() - fx["grad"]).max() <= 1e-6
    neg = coords.clone()
    neg[0] = torch.tensor([-1, -2], device=DEV)                                   # torch indexing semantics for negative indices
    assert torch.equal(extract_at_locations(feat, neg)[0], feat[:, -1, -2])


def test_out_of_range_indices_are_flagged_not_dereferenced(monkeypatch):
    """The reference's advanced indexing raises IndexError on an out-of-range coordinate / pair row; the kernels take raw pointers, so
    the wrappers clamp such indices (no out-of-bounds read), raise a sticky device flag (`ops.index_errors`, no sync on the hot path)
    and, with FRL_HIP_CHECK_INDICES=1, raise IndexError at once.  Negative indices wrap as in torch."""
    from frl_hip import ops
    from frl_hip.losses import contrastive_loss
    from frl_hip.utils import extract_at_locations
    ops.index_errors()
    feat = torch.randn(4, 6, 5, device=DEV)
    ok = torch.tensor([[0, 0], [5, 4], [-6, -5]], device=DEV)
    assert torch.equal(extract_at_locations(feat, ok), torch.stack([feat[:, 0, 0], feat[:, 5, 4], feat[:, 0, 0]]))
    assert not ops.index_errors()
    bad = torch.tensor([[0, 0], [6, 0], [0, 1 << 40]], device=DEV)
    out = extract_at_locations(feat, bad)                                          # clamped: finite values of the raster, no fault
    assert torch.isfinite(out).all() and ops.index_errors() and not ops.index_errors()
    emb = torch.randn(10, 8, device=DEV, requires_grad=True)
    pos = torch.tensor([[0, 1], [2, -1]], device=DEV)                              # -1 = row 9
    neg = torch.tensor([[0, 3], [2, 4]], device=DEV)
    l_ref = contrastive_loss(emb, torch.tensor([[0, 1], [2, 9]], device=DEV), neg)
    assert contrastive_loss(emb, pos, neg).item() == l_ref.item() and not ops.index_errors()
    l_bad = contrastive_loss(emb, torch.tensor([[0, 1], [2, 10]], device=DEV), neg)
    l_bad.backward()
    assert torch.isfinite(emb.grad).all() and ops.index_errors()
    monkeypatch.setenv("FRL_HIP_CHECK_INDICES", "1")
    with pytest.raises(IndexError):
        contrastive_loss(emb, torch.tensor([[0, 1], [2, 10]], device=DEV), neg)
    with pytest.raises(IndexError):
        extract_at_locations(feat, bad)
    ops.index_errors()
