"""VQ assign micro-benchmark at BASELINE configs[1] (262 144 rows, K = 512, d = 64, bf16): median / min of 30 launches (HIP events around the
C-ABI call), algorithmic bytes / time, and the share of rows re-evaluated exactly.  Codebook = rows drawn from z (a live codebook) or randn."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "vq-vae_amd"))
from frl_hip import ops  # noqa: E402

dev = "cuda:0"
N, K, d = 262144, 512, 64
g = torch.Generator().manual_seed(0)
z = torch.randn(N, d, generator=g).to(torch.bfloat16).to(dev)
variants = [int(v) for v in os.environ.get("VQ_BENCH_TILES", "0,1,2,4").split(",")]
for name, nt in [(nm, v) for nm in ("data",) for v in variants]:
    ops.vq_stream_tiles(nt)
    g.manual_seed(1)
    E = z[torch.randperm(N, generator=g)[:K].to(dev)].float().contiguous() if name == "data" else torch.randn(K, d, generator=g).to(dev)
    prep = ops.vq_prepare(E, N, torch.bfloat16)
    for _ in range(3):
        ops.vq_assign(z, E, prep)
    ts = []
    for _ in range(30):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        idx, zq, stats, counts = ops.vq_assign(z, E, prep)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    nbytes = N * (2 * d * 2 + 4) + K * d * 4
    print(json.dumps({"codebook": name, "stream_tiles": nt, "us_median": round(ts[15], 1), "us_min": round(ts[0], 1), "GB/s_at_min": round(nbytes / ts[0] / 1e3, 1),
                      "frac_of_8TB/s": round(nbytes / ts[0] / 1e3 / 8000, 3), "rows_reevaluated": float(stats[2]) / N, "perplexity": round(float(stats[1]), 1)}), flush=True)
