#!/usr/bin/env python3
"""Times mutual-kNN pair mining on the GPU box: the HIP path (`frl_mutual_knn` + the nonzero compaction) against the
reference's algorithm as it would run on the same GPU through PyTorch (chunked torch.cdist + topk + masking, frl/losses/pairs.py:531-610,
restated here for the comparison only).  Prints one JSON object."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vq-vae_amd"))
from frl_hip.losses import pairs_mutual_knn_chunked  # noqa: E402


def torch_style(features, coord_list, offsets, k, pos_min_spatial=4.0, chunk_size=128):
    n, dev = features.shape[0], features.device
    knn = torch.full((n, k), -1, dtype=torch.long, device=dev)
    for s in range(0, n, chunk_size):
        e = min(s + chunk_size, n)
        dist = torch.cdist(features[s:e], features)
        loc = torch.arange(e - s, device=dev)
        dist[loc, s + loc] = float("inf")
        for p, cp in enumerate(coord_list):
            ps, pe = offsets[p], offsets[p + 1]
            qs, qe = max(s, ps), min(e, pe)
            if qs >= qe:
                continue
            sp = torch.cdist(cp[qs - ps:qe - ps].float(), cp.float())
            blk = dist[qs - s:qe - s, ps:pe]
            blk[sp < pos_min_spatial] = float("inf")
        kk = min(k, n - 1)
        v, i = dist.topk(kk, dim=1, largest=False)
        i[torch.isinf(v)] = -1
        knn[s:e, :kk] = i
    ii = torch.arange(n, device=dev).repeat_interleave(k)
    jj = knn.reshape(-1)
    ok = jj >= 0
    ii, jj = ii[ok], jj[ok]
    mut = (knn[jj] == ii.unsqueeze(1)).any(1)
    return torch.stack([ii[mut], jj[mut]], 1)


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    dev = "cuda:0"
    out = []
    for n_p, patches, d, k in [(250, 8, 64, 8), (300, 16, 64, 16), (1000, 8, 64, 8)]:
        g = torch.Generator().manual_seed(0)
        feats = torch.randn(n_p * patches, d, generator=g).to(dev)
        coords = [torch.randint(0, 64, (n_p, 2), generator=g).to(dev) for _ in range(patches)]
        offsets = [n_p * p for p in range(patches + 1)]
        a = pairs_mutual_knn_chunked(feats, coords, offsets, k)
        b = torch_style(feats, coords, offsets, k)
        same = sorted(map(tuple, a.tolist())) == sorted(map(tuple, b.tolist()))
        n = n_p * patches
        ms_hip = timeit(lambda: pairs_mutual_knn_chunked(feats, coords, offsets, k), 20)
        ms_ref = timeit(lambda: torch_style(feats, coords, offsets, k), 5)
        from frl_hip import ops
        pid = torch.repeat_interleave(torch.arange(patches, dtype=torch.int32, device=dev), n_p).contiguous()
        cc = torch.cat(coords).float().contiguous()
        ms_kern = timeit(lambda: ops.mutual_knn(feats, pid, cc, k, 4.0), 20)
        out.append({"anchors": n, "D": d, "k": k, "pairs": int(a.shape[0]), "same_pairs_as_torch_algorithm": same, "hip_ms": round(ms_hip, 3), "hip_kernels_only_ms": round(ms_kern, 3),
                    "torch_cdist_topk_ms": round(ms_ref, 3), "speedup": round(ms_ref / ms_hip, 1),
                    "f32_GFLOP": round(3 * n * n * d / 1e9, 2), "hip_kernels_TFLOP_s": round(3 * n * n * d / ms_kern / 1e9, 2)})
    print(json.dumps({"mutual_knn": out}))


if __name__ == "__main__":
    main()
