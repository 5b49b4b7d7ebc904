"""Mutual-kNN pair mining on the GPU against the reference's own output (golden) and the float64 oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import frl_oracle as O  # noqa: E402

DEV = "cuda:0"


def _case(golden_dir, name):
    fx = np.load(os.path.join(golden_dir, f"mutual_knn_{name}.npz"))
    offsets = fx["offsets"].tolist()
    coords = [torch.from_numpy(fx["coords"][offsets[p]:offsets[p + 1]]).to(DEV) for p in range(len(offsets) - 1)]
    return fx, torch.from_numpy(fx["features"]).to(DEV), coords, offsets


@pytest.mark.parametrize("name", ["a", "b", "c", "d"])
def test_pairs_equal_the_reference_output(golden_dir, name):
    from frl_hip.losses import pairs_mutual_knn_chunked
    fx, feats, coords, offsets = _case(golden_dir, name)
    got = pairs_mutual_knn_chunked(feats, coords, offsets, int(fx["k"]), pos_min_spatial=float(fx["min_sp"]), chunk_size=128)
    assert got.dtype == torch.long and got.device.type == "cuda"
    assert np.array_equal(got.cpu().numpy(), fx["pairs"])                                  # same pairs, same order: bit-exact indices


def test_knn_table_matches_oracle_and_padding(golden_dir):
    from frl_hip import ops
    fx, feats, coords, offsets = _case(golden_dir, "c")                                    # 8 anchors, k = 16 > N - 1
    _, want = O.mutual_knn_pairs_np(fx["features"], [c.cpu().numpy() for c in coords], offsets, 16, float(fx["min_sp"]))
    pid = torch.repeat_interleave(torch.arange(2, dtype=torch.int32), torch.tensor([5, 3])).to(DEV)
    knn, mutual = ops.mutual_knn(feats, pid, torch.cat(coords).float().contiguous(), 16, float(fx["min_sp"]))
    assert np.array_equal(knn.cpu().numpy().astype(np.int64), want) and (knn[:, 7:] == -1).all().item()
    assert not mutual[knn < 0].any().item()


def test_large_problem_properties_and_limits():
    from frl_hip import _lib, ops
    from frl_hip.losses import pairs_mutual_knn_chunked
    g = torch.Generator().manual_seed(5)
    n_p, patches, d, k = 1000, 4, 64, 8
    feats = torch.randn(n_p * patches, d, generator=g).to(DEV)
    coords = [torch.randint(0, 64, (n_p, 2), generator=g).to(DEV) for _ in range(patches)]
    offsets = [n_p * p for p in range(patches + 1)]
    pairs = pairs_mutual_knn_chunked(feats, coords, offsets, k, pos_min_spatial=4.0)
    p = pairs.cpu().numpy()
    have = set(map(tuple, p.tolist()))
    assert len(have) == len(p) > 0 and all((j, i) in have for i, j in have) and all(i != j for i, j in have)
    allc = torch.cat(coords).float().cpu().numpy()
    pid = np.repeat(np.arange(patches), n_p)
    same = pid[p[:, 0]] == pid[p[:, 1]]
    sp = np.sqrt(((allc[p[:, 0]] - allc[p[:, 1]]) ** 2).sum(-1))
    assert (sp[same] >= 4.0).all()                                                       # spatial constraint inside a patch
    # every reported neighbour really is among the k nearest admissible anchors (float64 check on a sample of anchors)
    x = feats.double().cpu().numpy()
    for i in np.unique(p[:, 0])[:25]:
        d2 = ((x - x[i]) ** 2).sum(-1)
        d2[i] = np.inf
        close = (pid == pid[i]) & (np.sqrt(((allc - allc[i]) ** 2).sum(-1)) < 4.0)
        d2[close] = np.inf
        kth = np.sort(d2)[k - 1]
        assert (d2[p[p[:, 0] == i, 1]] <= kth * (1 + 1e-6)).all()
    assert pairs_mutual_knn_chunked(feats[:1], [coords[0][:1]], [0, 1], k).shape == (0, 2)   # a single anchor has no neighbour
    assert pairs_mutual_knn_chunked(feats[:0], [], [0], k).shape == (0, 2)
    with pytest.raises(_lib.FrlHipError):                                                 # the raw entry point wants padded widths
        _lib.check(_lib.load().frl_mutual_knn(feats.data_ptr(), 10, 12, None, None, 4.0, 2, None, None, None))
    nmax = _lib.load().frl_mutual_knn_max_points(d)
    big = torch.zeros(nmax + 1, d, device=DEV)
    with pytest.raises(_lib.FrlHipError):
        ops.mutual_knn(big, torch.zeros(nmax + 1, dtype=torch.int32, device=DEV), torch.zeros(nmax + 1, 2, device=DEV), k, 4.0)
