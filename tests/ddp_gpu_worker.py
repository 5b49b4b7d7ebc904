"""Worker of tests/test_gpu_model_parity.py::test_two_rank_data_parallel_matches_single_process (launched by torch.distributed.run).

Two ranks share cuda:0 over gloo (RCCL refuses two ranks on one device): rank r trains on tile r; rank 0 writes the parameters
after the steps.  Exercises the GPU data-parallel route: gradient hooks, multi-tensor pack into the flat buckets on the side
stream, all-reduce, isfinite flag slot, HipAdamW reading the buckets in place, phase branch on its own stream."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vq-vae_amd"))

from frl_hip.models import VQVAE  # noqa: E402
from frl_hip.training.trainer import VQVAETrainer  # noqa: E402


def main():
    out_path, fixture = sys.argv[1], sys.argv[2]
    rank = int(os.environ["RANK"])
    dist.init_process_group("gloo")
    dev = "cuda:0"
    fx = np.load(fixture)
    sd = {k[6:]: torch.from_numpy(fx[k]).float() for k in fx.files if k.startswith("state.")}
    m = VQVAE(in_features=8, codebook_size=16, emb_dim=8, beta=0.25, hidden=16, z_phase_dim=4, type_encoder_channels=(16, 8),
              type_encoder_dropout=0.0, type_encoder_num_groups=4, spatial_conv_gate_hidden=8, phase_tcn_channels=(8, 8, 8),
              phase_tcn_dropout=0.0, phase_tcn_num_groups=4, compute_dtype=torch.float32).to(dev)
    m.load_state_dict(sd, strict=True)
    tr = VQVAETrainer(m, lr=1e-3, total_steps=10)
    assert tr.reducer is not None and tr.reducer.active and tr.hip_opt
    tiles = torch.from_numpy(fx["tiles"]).float().to(dev)            # [3, B, T, H, W, F]
    losses = []
    for step in range(2):
        t = tiles[step]
        half = t.shape[0] // 2
        losses.append(float(tr.step(t[rank * half:(rank + 1) * half].contiguous())["loss"]))
    torch.cuda.synchronize()
    if rank == 0:
        np.savez(out_path, losses=np.asarray(losses), **{n: p.detach().cpu().numpy() for n, p in m.named_parameters()})
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
