#!/usr/bin/env python3
"""bench.py -- tiles/sec of the full VQ-VAE train step (encoder -> VQ -> decoders, fwd + bwd + AdamW) on MI355X.

Contract: python bench.py --gpus N --steps K --warmup W.  N > 1 runs one rank per GPU over RCCL: either the caller starts the ranks
(python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...: WORLD_SIZE is set and must equal N) or, when WORLD_SIZE
is unset, this process starts them itself as a child `torch.distributed.run` BEFORE touching the GPU and relays rank 0's JSON line.
Prints ONE JSON line on rank 0: metric tiles/sec on BASELINE.json configs[1] (B=256 tiles of 5x32x32x64 per GPU, K=512,
d=64, bf16 activations, float32 master weights), plus
  * "roofline": the dominant kernel's achieved rate, timed live with HIP events on the launch stream;
  * "kernels":  per-kernel-family time shares from the same events (conv MFMA %, VQ HBM GB/s of the north star);
  * "cpu_baseline": the CPU oracle (reference encoder math + oracle VQ/decoder) timed on this node's host cores, N=1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "vq-vae_amd"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
MFMA_BF16_PEAK_TF = 2500.0   # dense bf16


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256, help="tiles per GPU (weak scaling)")
    ap.add_argument("--codebook", type=int, default=512)
    ap.add_argument("--emb-dim", type=int, default=64)
    ap.add_argument("--time", type=int, default=5)
    ap.add_argument("--size", type=int, default=32)
    ap.add_argument("--features", type=int, default=64)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-finite-check", action="store_true")
    ap.add_argument("--no-collapsed-line", action="store_true", help="skip the secondary timing on the randn (collapsing) codebook")
    ap.add_argument("--serial-streams", action="store_true",
                    help="run the phase branch on the main stream (per-kernel profiling: no overlap between the two branches)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend; nccl == RCCL (the measured configuration), gloo only to rehearse the N>1 control flow")
    ap.add_argument("--no-defer-reductions", action="store_true",
                    help="A/B: one slab-reduction launch behind every weight-gradient kernel instead of one launch for all of them")
    ap.add_argument("--torch-optimizer", action="store_true", help="A/B: clip_grad_norm_ + torch.optim.AdamW instead of the two HIP launches")
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"],
                    help="replay the train step from a captured hipGraph (VQVAETrainer.step_graphed); auto: whenever the trainer supports it")
    ap.add_argument("--phase-codebook", type=int, default=0, help="second codebook on z_phase (BASELINE configs[4])")
    ap.add_argument("--codebook-init", default="data", choices=["data", "randn"],
                    help="data: codes start on encoder outputs of the first batch (every code in use: the timed steps run a live codebook); "
                         "randn: E = randn(K, d), which collapses to a few codes on randn tiles (reported as the secondary line either way)")
    ap.add_argument("--extra", action="store_true",
                    help="N=1 only: also time BASELINE configs[3] (K=8192, d=128), configs[4] at one GPU (T=10, 64x64, two codebooks of 1024) "
                         "and the float32 parity mode of configs[1]; reported under \"extra\" (secondary lines, never `value`)")
    return ap.parse_args()


def launch_ranks(args) -> int:
    """--gpus N > 1 without a launcher: start N ranks as a child torch.distributed.run (this process has not touched the GPU and
    never will), relay rank 0's JSON line, return the child's exit code."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    if proc.returncode != 0 or not lines:
        sys.stdout.write(proc.stdout)
        sys.stderr.write(f"bench.py: the {args.gpus}-rank run failed (exit code {proc.returncode})\n")
        return proc.returncode or 1
    print(lines[-1], flush=True)
    return 0


def conv_flops_per_tile(T, S, F, d, zp, hidden=128):
    """Algorithmic dense-contraction FLOPs per tile, forward, train step = 3x."""
    px = S * S
    enc = 2 * (F * 128 + 128 * d)
    spatial = 2 * (2 * d * 64 * 9) + 2 * (64 * 32) + 2 * (64 * d * 4) + 2 * (d * 64 * 9) + 2 * (64 * d * 9)
    tcn = T * 3 * (2 * (F * 64 * 3) + 2 * 64 * 64)              # per pixel, three blocks of 64 channels
    head = T * 2 * 64 * zp + 2 * (2 * (d * 32) + 2 * (32 * zp))
    dec = 2 * (d * hidden + hidden * F) + T * 2 * (zp * hidden + hidden * F)
    return px * (enc + spatial + tcn + head + dec)


def cpu_baseline(args):
    """Oracle train step on the host cores (rank 0, N=1): bounded sample of the same workload, protocol of BASELINE.md section 4
    (batches of 1 and 8 tiles, 3 warm-up + 10 timed steps each, median), at the codebook of the measured configuration."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import frl_oracle as O
    cores = min(os.cpu_count() or 1, 16)      # the box's CPU share for one GPU; more threads only thrash on these small ops
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(0)
    from frl_hip.models import VQVAE
    m = VQVAE(in_features=args.features, codebook_size=args.codebook, emb_dim=args.emb_dim, type_encoder_dropout=0.0,
              phase_tcn_dropout=0.0)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    sd["quant.codebook"] = torch.randn(args.codebook, args.emb_dim, generator=g)
    rates, total = {}, 0.0
    for bs in (1, 8):
        tr = O.OracleTrainer(sd, dict(beta=0.25), lr=1e-4, total_steps=100)
        tiles = [torch.randn(bs, args.time, args.size, args.size, args.features, generator=torch.Generator().manual_seed(1234 + i))
                 for i in range(2)]
        for i in range(3):
            tr.step(tiles[i % 2])
        ts = []
        for i in range(10):
            t0 = time.perf_counter()
            tr.step(tiles[i % 2])
            ts.append(time.perf_counter() - t0)
        total += sum(ts)
        rates[bs] = bs / sorted(ts)[len(ts) // 2]
    return {"value": round(rates[8], 2), "unit": "tiles/s", "cores": cores, "kind": "port",
            "sample": f"BASELINE.md section 4 protocol at this run's codebook: oracle train steps (reference encoder math + oracle VQ / decoder, "
                      f"f32 torch CPU, float64 argmin, K={args.codebook}, d={args.emb_dim}), 3 warm-up + 10 timed steps per batch size, "
                      f"median; value = batches of 8 tiles, batches of 1 tile: {rates[1]:.2f} tiles/s; {total:.1f} s timed"}


def build_trainer(args, dev, dtype, codebook, emb_dim, phase_codebook=0, serial=False, init_tile=None):
    """init_tile: a batch of tiles -> the codebook(s) start on encoder outputs drawn from it (VQVAE.init_codebook_from_tiles: every
    code is in use, perplexity of the order of K -- a codebook in the regime training aims for); None -> E = randn(K, d), seed 7
    (SURVEY.md 8d's argmin-parity codebook), which on randn tiles collapses to a handful of codes within a few steps."""
    from frl_hip.models import VQVAE
    from frl_hip.training.trainer import VQVAETrainer
    torch.manual_seed(0)
    model = VQVAE(in_features=args.features, codebook_size=codebook, emb_dim=emb_dim, beta=0.25, phase_codebook_size=phase_codebook,
                  type_encoder_dropout=0.0, phase_tcn_dropout=0.0, compute_dtype=dtype).to(dev)
    with torch.no_grad():
        model.quant.codebook.copy_(torch.randn(codebook, emb_dim, generator=torch.Generator().manual_seed(7)))
        if phase_codebook:
            model.quant_phase.codebook.copy_(torch.randn(phase_codebook, model.z_phase_dim, generator=torch.Generator().manual_seed(8)))
    if init_tile is not None:
        model.init_codebook_from_tiles(init_tile, seed=7)
    model.concurrent_phase = not serial
    trainer = VQVAETrainer(model, lr=1e-4, total_steps=10000, check_finite=not args.no_finite_check, fused_optimizer=not args.torch_optimizer,
                           defer_reductions=not args.no_defer_reductions)
    return model, trainer


def timed_steps(trainer, stream, steps, warmup, barrier, graphed=False):
    """W untimed + exactly K timed steps bracketed by barrier + synchronize; returns (seconds, host seconds to queue the K steps, last out)."""
    step = trainer.step_graphed if graphed else trainer.step
    if graphed:                                       # set-up, before the W warm-up steps: one capture per buffer of the tile pool
        for _ in range(len(stream.tiles)):
            step(stream.next())
    for _ in range(warmup):
        step(stream.next())
    barrier()
    t0 = time.perf_counter()
    last = None
    for _ in range(steps):                            # timed region: exactly K steps, no instrumentation
        last = step(stream.next())
    t_host = time.perf_counter() - t0                 # the host has queued everything; the device may still be running
    barrier()
    return time.perf_counter() - t0, t_host, last


def extra_lines(args, dev, barrier):
    """Secondary single-GPU lines (never `value`): BASELINE configs[3], configs[4] at one GPU, and configs[1] in float32 parity mode."""
    from frl_hip.data import SyntheticTileStream
    out = {}
    cases = [("configs[3]: K=8192 d=128, 256 tiles 5x32x32x64, bf16", dict(dtype=torch.bfloat16, K=8192, d=128, pK=0, B=256, T=5, S=32)),
             ("configs[4] at 1 GPU: T=10, 64x64 tiles, type + phase codebooks of 1024, 32 tiles, bf16",
              dict(dtype=torch.bfloat16, K=1024, d=64, pK=1024, B=32, T=10, S=64)),
             ("configs[1] in float32 parity mode (the mode that meets 1e-5): K=512 d=64, 256 tiles 5x32x32x64",
              dict(dtype=torch.float32, K=512, d=64, pK=0, B=256, T=5, S=32))]
    for name, c in cases:
        model, trainer = build_trainer(args, dev, c["dtype"], c["K"], c["d"], c["pK"])
        stream = SyntheticTileStream(c["B"], c["T"], c["S"], args.features, device=dev, dtype=c["dtype"], seed=1234)
        dt, th, last = timed_steps(trainer, stream, 10, 3, barrier, graphed=args.graph != "off" and trainer.graph_supported())
        out[name] = {"tiles/s": round(c["B"] * 10 / dt, 1), "ms_per_step": round(1e2 * dt, 3), "host_ms_per_step": round(1e2 * th, 3),
                     "steps": 10, "warmup": 3, "loss": round(float(last["loss"].detach()), 5)}
        del model, trainer, stream, last
        torch.cuda.empty_cache()
    return out


def main():
    args = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:            # nobody started the ranks: do it here, before anything touches the GPU
        raise SystemExit(launch_ranks(args))
    world = int(env_world or "1")
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start exactly --gpus ranks "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        if world > 1:
            import time
            time.sleep(3.0)        # the launcher ends the other ranks as soon as one fails: let every rank reach this line and say so
        raise SystemExit("bench.py needs an MI355X: the hot path is HIP-only (no CPU fallback)")
    local_rank %= max(torch.cuda.device_count(), 1)      # (identity on an N-GPU node; lets a 1-GPU box rehearse N ranks with gloo)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)     # "nccl" == RCCL on ROCm
        else:
            dist.init_process_group("gloo")
    from frl_hip import ops
    from frl_hip.data import SyntheticTileStream

    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    stream = SyntheticTileStream(args.batch, args.time, args.size, args.features, device=dev, dtype=dtype, seed=1234 + rank)
    init_tile = None if args.codebook_init == "randn" else stream.tiles[0]
    model, trainer = build_trainer(args, dev, dtype, args.codebook, args.emb_dim, args.phase_codebook, args.serial_streams, init_tile)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    timing = (not args.no_kernel_timing) and rank == 0
    # auto: the captured graph on one GPU; N > 1 runs the eager data-parallel step (GPU-bound as well: ~1.5 ms of host time against a 2.5 ms
    # step) -- the capture of the RCCL all-reduces is verified bit for bit at one rank only (RCCL refuses two ranks on the single GPU of the
    # development box), so it stays opt-in (--graph on) until it has run on a multi-GPU node
    graphed = args.graph != "off" and trainer.graph_supported() and (world == 1 or args.graph == "on")
    if args.graph == "on" and not graphed:
        raise SystemExit("bench.py: --graph on, but this trainer configuration cannot be captured (data-parallel reducer or a per-step lambda_vq)")
    dt, t_host, last = timed_steps(trainer, stream, args.steps, args.warmup, barrier, graphed)
    ksum, kern, ksteps = {}, {}, 0
    if not args.no_kernel_timing:                     # separate instrumented steps: HIP events around every C-ABI call (rank 0)
        ksteps = max(2, min(5, args.steps))           # EVERY rank runs them: a step contains collectives
        model.concurrent_phase = False                # per-kernel spans are only meaningful without the two branches overlapping
        if timing:
            ops.set_timing(True)                      # HIP events around every C-ABI call (op families) ...
            ops.kernel_timing(True)                   # ... and, inside the library, around every single kernel launch
        for _ in range(ksteps):
            trainer.step(stream.next())
        if timing:
            ksum = ops.timing_summary()
            kern = ops.kernel_timing_report()
            ops.set_timing(False)
            ops.kernel_timing(False)
    if world > 1:
        barrier()
    rank_ms = [1e3 * dt / args.steps]
    rccl_ranks = 1
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        if args.backend == "gloo":
            t = t.cpu()
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        rank_ms = [1e3 * float(x.item()) / args.steps for x in allt]
        dt = max(float(x.item()) for x in allt)       # MAX over ranks
        rccl_ranks = dist.get_world_size()            # as seen inside the process group the gradients are reduced in
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    tiles = args.batch * world * args.steps
    out = {
        "metric": "tiles/sec", "value": round(tiles / dt, 1), "unit": "tiles/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"BASELINE configs[1]: full VQ-VAE train step (type path + VQ + decoders + dense phase path, "
                               f"fwd+bwd+clip+AdamW), {args.batch} tiles/GPU of {args.time}x{args.size}x{args.size}x{args.features}, "
                               f"K={args.codebook}, d={args.emb_dim}, dropout 0.0",
                   "global_batch": args.batch * world, "parallelism": f"dp{world}", "finite_check": not args.no_finite_check,
                   "launch": "hipGraph replay of the whole step (one graph per tile-pool buffer)" if graphed else "eager (one launch per kernel)",
                   "deferred_reductions": bool(trainer.defer_reductions and not (trainer.reducer is not None and trainer.reducer.active)),
                   "loss": round(float(last["loss"].detach()), 5), "perplexity": round(float(last["perplexity"]), 2),
                   "codebook_init": "encoder outputs of the first batch (VQVAE.init_codebook_from_tiles)" if args.codebook_init == "data"
                   else "randn(K, d), seed 7",
                   "vq_rows_reevaluated": round(float(last["vq_stats"][2]) / (args.batch * args.size * args.size), 5)},
        "host_ms_per_step": round(1e3 * t_host / args.steps, 3),
        "rank_ms_per_step": {"max": round(max(rank_ms), 3), "min": round(min(rank_ms), 3)},
        "backend": ("rccl" if args.backend == "nccl" else "gloo") if world > 1 else None, "rccl_ranks": rccl_ranks,
    }
    if ksum:
        args.steps_timed, args.steps = args.steps, ksteps           # per-step normalisation of the instrumented steps
        n = args.batch * args.size * args.size                      # vectors / pixels per step
        s = 2 if args.dtype == "bf16" else 4
        out["kernels"] = {k: {"calls_per_step": c / args.steps, "ms_per_step": round(ms / args.steps, 4)} for k, (c, ms) in ksum.items()}
        # every kernel the library launched, event pair recorded inside the library right around the launch (the live counterpart of
        # the rocprofv3 kernel trace under profiles/)
        out["gpu_kernels"] = {k: {"launches_per_step": c / args.steps, "avg_us": round(1e3 * ms / c, 2), "ms_per_step": round(ms / args.steps, 4)}
                              for k, (c, ms) in sorted(kern.items(), key=lambda kv: -kv[1][1])}
        out["gpu_kernel_ms_per_step"] = round(sum(ms for _, ms in kern.values()) / args.steps, 3)
        # --- VQ assign: HBM roofline, algorithmic bytes = 2*d*s + 4 per vector (+ K*d*4 codebook once)
        vq_ms = ksum["vq_assign"][1] / ksum["vq_assign"][0]
        vq_bytes = n * (2 * args.emb_dim * s + 4) + args.codebook * args.emb_dim * 4
        out["vq_hbm"] = {"GB/s": round(vq_bytes / vq_ms / 1e6, 1), "frac": round(vq_bytes / vq_ms / 1e6 / HBM_PEAK_GBS, 4),
                         "ms": round(vq_ms, 4), "bytes": vq_bytes,
                         "note": "live HIP-event span of the whole frl_vq_assign_fwd call (every kernel the call launches)"}
        out["vq_hbm"]["rows_reevaluated"] = out["config"]["vq_rows_reevaluated"]     # stats[2] / N of the last timed step
        out["vq_hbm"]["perplexity"] = out["config"]["perplexity"]
        kv = [v for k, v in kern.items() if "vq_assign" in k]
        if kv:                                                # the L2/argmin kernel alone (library-side event pair around its launch)
            us = 1e3 * sum(ms for _, ms in kv) / sum(c for c, _ in kv)
            out["vq_hbm"]["assign_kernel"] = {"us": round(us, 2), "GB/s": round(vq_bytes / us / 1e3, 1),
                                              "frac": round(vq_bytes / us / 1e3 / HBM_PEAK_GBS, 4)}
        ku = profiled_kernel_us("vq_assign_kernel")
        if ku is not None:                                    # ... and from the committed rocprofv3 kernel trace
            out["vq_hbm"]["assign_kernel_rocprof"] = {"us": ku[0], "GB/s": round(vq_bytes / ku[0] / 1e3, 1),
                                                      "frac": round(vq_bytes / ku[0] / 1e3 / HBM_PEAK_GBS, 4), "source": ku[1]}
        # --- conv MFMA: algorithmic dense FLOPs of the step (3 x forward) over the event time of every conv / TCN op
        conv_keys = [k for k in ksum if k.startswith(("conv", "tcn", "film_fused", "smooth_heads", "decoder_mse", "encoder2")) and k != "tcn_block_bwd.main"]
        conv_ms = sum(ksum[k][1] for k in conv_keys) / args.steps
        flops = 3 * conv_flops_per_tile(args.time, args.size, args.features, args.emb_dim, model.z_phase_dim) * args.batch
        out["conv_mfma"] = {"TFLOP/s": round(flops / conv_ms / 1e9, 1), "frac": round(flops / conv_ms / 1e9 / MFMA_BF16_PEAK_TF, 4),
                            "ms_per_step": round(conv_ms, 3), "GFLOP_per_step": round(flops / 1e9, 1),
                            "note": "algorithmic 3x-forward dense FLOPs / HIP-event time of all conv + TCN ops (C-ABI call spans: "
                                    "slab reductions and launch gaps of each call included)"}
        # the same FLOPs over the contraction kernels alone (library-side event pairs around each launch)
        mm = ("tcn_hot", "tcn_chain", "pw_conv", "pw_wgrad", "conv3x3", "dec_mse_fwd", "dec_mse_bwd", "enc2_", "film_fused", "smooth_heads")
        mm_ms = sum(ms for k, (_, ms) in kern.items() if any(t in k for t in mm)) / args.steps
        if mm_ms > 0:
            out["conv_mfma"]["kernels_only"] = {"TFLOP/s": round(flops / mm_ms / 1e9, 1), "ms_per_step": round(mm_ms, 3),
                                                "frac": round(flops / mm_ms / 1e9 / MFMA_BF16_PEAK_TF, 4)}
        mu = mfma_util_from_profiles()
        if mu:
            out["conv_mfma"]["mfma_util"] = mu
        # --- dominant kernel -> headline roofline object
        # dominant kernel FAMILY among those with a per-launch work model below (the fused TCN kernels at the measured configuration)
        # (kernel families with a per-launch work model; time = the library-side event pairs around the kernel launches themselves)
        fam = {"tcn_block_bwd.main": "tcn_hot_bwd4_kernel", "tcn_block_bwd.nodx": "tcn_hot_bwd4_nodx_kernel", "tcn_block_bwd.head": "tcn_hot_bwd4_head_kernel", "tcn_block_fwd": "tcn_hot_fwd",
               "tcn_chain_fwd": "tcn_chain_fwd_kernel", "vq_assign": "vq_assign_kernel", "edge_smooth_fwd": "smooth_fwd_bf16",
               "edge_smooth_bwd": "smooth_bwd_bf16", "smooth_heads_fwd": "smooth_heads_fwd_kernel", "smooth_heads_bwd": "smooth_heads_bwd_kernel",
               "smooth_dx": "smooth_dx"}
        ftime = {}
        for name, sub in fam.items():
            hit = [v for k, v in kern.items() if sub in k]
            if hit:
                ftime[name] = (sum(c for c, _ in hit), sum(ms for _, ms in hit))
        if not ftime:                                               # (generic kernels only, e.g. --dtype f32): fall back to the op spans
            ftime = {k: ksum[k] for k in fam if k in ksum}
        dom = max(ftime, key=lambda k: ftime[k][1])
        out["roofline"] = roofline_for(dom, ftime, args, model, n, s)
        args.steps = args.steps_timed
    if world == 1 and args.codebook_init == "data" and not args.no_collapsed_line:
        # secondary line (never `value`): the same step on the randn codebook, whose assignment collapses to a few codes -- the regime
        # the earlier rounds' lines were measured in (the VQ kernel's re-evaluation share and histogram contention depend on it)
        m2, t2 = build_trainer(args, dev, dtype, args.codebook, args.emb_dim, args.phase_codebook, args.serial_streams, None)
        dt2, _, last2 = timed_steps(t2, stream, 20, 25, barrier, graphed)
        out["collapsed_codebook"] = {"codebook_init": "randn(K, d), seed 7", "ms_per_step": round(1e3 * dt2 / 20, 3),
                                     "tiles/s": round(args.batch * 20 / dt2, 1), "steps": 20, "warmup": 25,
                                     "perplexity": round(float(last2["perplexity"]), 2),
                                     "vq_rows_reevaluated": round(float(last2["vq_stats"][2]) / (args.batch * args.size * args.size), 5)}
        del m2, t2, last2
    if world == 1 and args.extra:
        del trainer, stream, last
        out["extra"] = extra_lines(args, dev, barrier)
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args)
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def mfma_util_from_profiles():
    """MFMA busy fraction per conv kernel family from the newest committed SQ counter pass (profiles/*_mfma_util.json), or None."""
    try:
        import glob
        f = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_mfma_util.json")))[-1]
        d = json.load(open(f))
        return {"source": os.path.relpath(f, ROOT), **d.get("kernels", {})}
    except Exception:
        return None


def profiled_kernel_us(substr):
    """(average duration in us, file) of a kernel in the newest committed rocprofv3 kernel-trace summary, or None."""
    try:
        import csv
        import glob
        f = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_kernel_stats.csv")))[-1]
        for r in csv.DictReader(open(f)):
            if substr in r["name"]:
                return round(float(r["avg_us"]), 1), os.path.relpath(f, ROOT)
    except Exception:
        pass
    return None


def pmc_traffic(name):
    """HBM bytes per launch from the newest committed rocprofv3 --pmc passes (profiles/*_pmc.json), or None."""
    key = {"tcn_block_bwd.main": ", true, false>", "tcn_block_bwd.nodx": ", false, false>", "tcn_block_bwd.head": ", true, true>", "tcn_block_fwd": "tcn_hot_fwd_kernel",
           "tcn_chain_fwd": "tcn_chain_fwd_kernel", "vq_assign": "vq_assign", "edge_smooth_bwd": "smooth_bwd_bf16r4_kernel",
           "smooth_heads_fwd": "smooth_heads_fwd_kernel", "smooth_heads_bwd": "smooth_heads_bwd_kernel", "smooth_dx": "smooth_dx",
           "conv1x1_bwd_weight": "pw_wgrad_kernel"}.get(name)
    try:
        import glob
        f = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")))[-1]
        hit = [v["hbm_bytes_per_launch"] for k, v in json.load(open(f))["kernels"].items() if key and key in k]
        if hit:                                                     # the template instances of one family (dilation 1 / 2 / 4) averaged
            return int(sum(hit) / len(hit))
    except Exception:
        pass
    return None


def roofline_for(name, ksum, args, model, n, s):
    """Roofline entry for the kernel with the largest share of the step: per-launch algorithmic work / average launch time
    (HIP events recorded on the launch stream around the C-ABI call).  The fused TCN kernels are dense contractions wrapped
    in a long per-element chain (GroupNorm, sigmoid gate, their backward): both ceilings are reported, `bound` is the one
    the kernel sits closer to, and the note says what actually limits it (vector-ALU issue)."""
    calls, ms = ksum[name]
    avg = ms / calls
    T, d = args.time, args.emb_dim
    base = {"kernel": name, "traffic": pmc_traffic(name), "avg_ms": round(avg, 4), "launches_per_step": calls / args.steps}
    if name.startswith("tcn_"):
        rows = n * T
        # valid temporal taps averaged over the three dilations (1, 2, 4) at this T
        taps = sum(sum(1 for t in range(T) for k in (-1, 0, 1) if 0 <= t + k * dl < T) for dl in (1, 2, 4)) / (3.0 * T)
        if name == "tcn_block_fwd":
            flops = rows * 2 * 64 * 64 * (taps + 1)                 # conv + gate GEMM
            nbytes = rows * 64 * s * 2                              # x read, y written
        elif name == "tcn_chain_fwd":                               # three blocks + the 1x1 head in one launch
            zp = model.z_phase_dim
            flops = rows * (3 * 2 * 64 * 64 * (taps + 1) + 2 * 64 * zp)
            nbytes = rows * s * (64 * 4 + zp)                       # x read; y1, y2, y3 and the head output written
        elif name == "tcn_block_bwd.nodx":                          # the block whose input is the tile itself: no conv^T, no dx
            flops = rows * 2 * 64 * 64 * (2 * taps + 3)
            nbytes = rows * 64 * s * 2
        elif name == "tcn_block_bwd.head":                          # the last block fed with the head's output gradient dh [rows][zp]
            zp = model.z_phase_dim
            flops = rows * (2 * 64 * 64 * (3 * taps + 3) + 2 * 2 * 64 * zp)
            nbytes = rows * s * (64 * 2 + zp)
        else:
            flops = rows * 2 * 64 * 64 * (3 * taps + 3)             # conv recompute, gate, gate^T, conv^T, two weight-gradient GEMMs
            nbytes = rows * 64 * s * 3                              # x, dy read; dx written
        mf = flops / avg / 1e9 / MFMA_BF16_PEAK_TF
        hf = nbytes / avg / 1e6 / HBM_PEAK_GBS
        hbm = {"bound": "hbm", "achieved": round(nbytes / avg / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hf, 4),
               "bytes_per_launch": nbytes}
        mfma = {"bound": "mfma", "achieved": round(flops / avg / 1e9, 1), "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s", "frac": round(mf, 4),
                "flops_per_launch": flops}
        first, second = (hbm, mfma) if hf >= mf else (mfma, hbm)
        base.update(first)
        base["other_ceiling"] = second
        if name == "tcn_chain_fwd":
            why = ("issue-bound: three blocks + head per 16-pixel wave tile (~450 MFMAs and the GroupNorm / sigmoid chains of 240 elements per lane), "
                   "two waves per SIMD, 42 % of the wave time in waits (SQ counters, profiles/r03_mfma_util.json)")
        else:
            why = ("issue-bound: ~1770 vector-ALU instructions and 226 MFMAs per wave and 32-pixel tile, two waves per SIMD (one of each subgroup)")
        base["note"] = ("HIP-event pair inside the library around the kernel launch itself (weight pack and slab reduction of the C-ABI call "
                        "excluded); " + why + ": DESIGN.md section 4")
        return base
    if name == "vq_assign":
        b = n * (2 * d * s + 4) + args.codebook * d * 4
        base.update({"bound": "hbm", "achieved": round(b / avg / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(b / avg / 1e6 / HBM_PEAK_GBS, 4), "bytes_per_launch": b})
        return base
    if name in ("smooth_heads_fwd", "smooth_heads_bwd", "smooth_dx"):
        c = args.features
        per_px = {"smooth_heads_fwd": 4 * c,                        # x, feat read; smoothed, residual written
                  "smooth_heads_bwd": 4 * c + 4 * c + 32,           # ds, x, feat read, dfeat written; u [4C] and soft-maxed A [32] written
                  "smooth_dx": 4 * c + 32 + 3 * c}[name]           # u, A read; ds, dx_add read, dx written
        b = n * s * per_px
        base.update({"bound": "hbm", "achieved": round(b / avg / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(b / avg / 1e6 / HBM_PEAK_GBS, 4), "bytes_per_launch": b})
        return base
    if name.startswith("edge_smooth"):
        c = args.features                                           # x, A (8x4), B (4C) in / out plus smoothed, residual (fwd) or ds, dx (bwd)
        b = n * s * (11 * c + 64)
        base.update({"bound": "hbm", "achieved": round(b / avg / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(b / avg / 1e6 / HBM_PEAK_GBS, 4), "bytes_per_launch": b})
        return base
    base.update({"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None})
    return base


if __name__ == "__main__":
    main()
