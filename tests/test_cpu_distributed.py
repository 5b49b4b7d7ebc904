"""world_size=2 gloo tests of the data-parallel path (the RCCL path uses the same code with backend "nccl")."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Toy(nn.Module):
    def __init__(self):
        super().__init__()
        self.encoder = nn.Linear(6, 5)
        self.spatial_conv = nn.Linear(5, 5)
        self.decoder_type = nn.Linear(5, 6)
        self.unused = nn.Linear(3, 3)      # never receives a gradient: must not dead-lock the buckets

    def forward(self, x):
        return self.decoder_type(torch.tanh(self.spatial_conv(torch.tanh(self.encoder(x)))))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vq-vae_amd"))
    from frl_hip.parallel import BucketedGradAllReduce
    torch.manual_seed(0)
    m = Toy()
    red = BucketedGradAllReduce(list(m.named_parameters()))
    assert len(red.buckets) == 2                      # early: decoder/unused, late: encoder + spatial_conv
    g = torch.Generator().manual_seed(5)
    x = torch.randn(8, 6, generator=g)
    for step in range(2):                             # two steps: bucket state must reset
        m.zero_grad(set_to_none=True)
        xs = x[rank * 4:(rank + 1) * 4] + step
        (m(xs) - xs).pow(2).mean().backward()
        red.finish()
    grads = {n: (p.grad.clone() if p.grad is not None else None) for n, p in m.named_parameters()}
    if rank == 0:
        ref = Toy()
        ref.load_state_dict(m.state_dict())
        xs = x + 1
        (ref(xs) - xs).pow(2).mean().backward()
        ok = all((grads[n] - p.grad).abs().max().item() < 1e-6 for n, p in ref.named_parameters() if p.grad is not None)
        ok = ok and all(float(grads[n].abs().max()) == 0.0 for n in ("unused.weight", "unused.bias"))
        q.put(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_equals_single_process_on_concatenated_batch():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def _worker_finite(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vq-vae_amd"))
    from frl_hip.training.trainer import VQVAETrainer

    class M(Toy):
        def forward_tiles(self, tile, mask=None):
            y = self(tile)
            loss = (y - tile).pow(2).mean()
            if rank == 1 and float(tile[0, 0]) > 100:          # only rank 1 sees a non-finite loss
                loss = loss * float("nan")
            return {"loss": loss}

    torch.manual_seed(0)
    m = M()
    tr = VQVAETrainer(m, lr=1e-2, total_steps=10, fused_optimizer=False)
    before = [p.detach().clone() for p in m.parameters()]
    bad = torch.full((4, 6), 1000.0)
    tr.step(bad)                                               # every rank must skip (consistent guard), no dead-lock
    same = all(torch.equal(a, b) for a, b in zip(before, m.parameters()))
    tr.step(torch.randn(4, 6))
    moved = any(not torch.equal(a, b) for a, b in zip(before, m.parameters()))
    q.put((rank, same, moved, tr.skipped))
    dist.barrier()
    dist.destroy_process_group()


def test_trainer_finite_guard_is_collective():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_finite, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(2))
    assert all(same and moved and skipped == 1 for _, same, moved, skipped in res), res


def _worker_ema(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "vq-vae_amd"))
    from frl_hip import ops
    from frl_hip.models.vqvae import VectorQuantizer

    def ema_cpu(sums, counts, ema_count, ema_sum, codebook, decay, eps, ok=None):     # the kernel's arithmetic (csrc/vq.hip: vq_ema_kernel)
        ema_count.mul_(decay).add_((1 - decay) * counts.float())
        n = ema_count.double().sum().float()
        ema_sum.mul_(decay).add_((1 - decay) * sums)
        k = codebook.shape[0]
        codebook.copy_(ema_sum / ((ema_count + eps) / (n + k * eps) * n).unsqueeze(1))

    ops.vq_ema_update = ema_cpu                      # CPU stand-in for the HIP kernel: the collective logic around it is what is tested
    torch.manual_seed(0)
    vq = VectorQuantizer(codebook_size=6, emb_dim=4, quantizer="ema")
    g = torch.Generator().manual_seed(100 + rank)     # every rank sees a DIFFERENT batch
    for _ in range(3):
        counts = torch.randint(0, 9, (6,), generator=g, dtype=torch.int32)
        sums = torch.randn(6, 4, generator=g) * counts.unsqueeze(1)
        vq._pending_ema = (sums, counts)
        vq.apply_ema()
    q.put((rank, vq.codebook.detach().clone(), vq.ema_count.clone(), vq.ema_sum.clone()))
    dist.barrier()
    dist.destroy_process_group()


def test_ema_quantizer_sums_statistics_over_ranks():
    """quantizer='ema' under data parallel: the codebook is no parameter (no gradient bucket), so the per-code counts and sums must be
    all-reduced before the EMA update -- otherwise every rank drifts to its own codebook from the first step on."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_ema, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    (_, cb0, n0, s0), (_, cb1, n1, s1) = res
    assert torch.equal(cb0, cb1) and torch.equal(n0, n1) and torch.equal(s0, s1)
    # and they equal ONE process fed the summed statistics
    torch.manual_seed(0)
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vq-vae_amd"))
    from frl_hip.models.vqvae import VectorQuantizer
    vq = VectorQuantizer(codebook_size=6, emb_dim=4, quantizer="ema")
    gens = [torch.Generator().manual_seed(100 + r) for r in range(2)]
    ema_count, ema_sum = vq.ema_count.clone(), vq.ema_sum.clone()
    for _ in range(3):
        cs, ss = [], []
        for g in gens:
            c = torch.randint(0, 9, (6,), generator=g, dtype=torch.int32)
            cs.append(c)
            ss.append(torch.randn(6, 4, generator=g) * c.unsqueeze(1))
        ema_count = 0.99 * ema_count + 0.01 * (cs[0] + cs[1]).float()
        ema_sum = 0.99 * ema_sum + 0.01 * (ss[0] + ss[1])
    assert torch.allclose(n0, ema_count, atol=1e-6) and torch.allclose(s0, ema_sum, atol=1e-6)
