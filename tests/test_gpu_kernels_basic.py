"""GPU parity of the first kernel families against the CPU oracle (through the C ABI)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import frl_oracle as O  # noqa: E402


def _dev():
    assert torch.cuda.is_available(), "GPU test run without a GPU"
    return torch.device("cuda:0")


def test_library_reports_gfx950():
    import ctypes
    from frl_hip import _lib
    lib = _lib.load()
    _dev()
    buf = ctypes.create_string_buffer(128)
    _lib.check(lib.frl_device_arch(buf, 128))
    assert buf.value.decode().startswith("gfx950"), buf.value


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-6), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("P,cin,cout", [(1024, 64, 128), (1000, 128, 64), (37, 64, 12), (256, 12, 128),
                                        (64, 8, 16), (2048, 64, 256), (130, 32, 12), (512, 64, 32)])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_conv1x1_fwd(dtype, tol, P, cin, cout, act):
    from frl_hip import ops
    dev = _dev()
    g = torch.Generator().manual_seed(P + cin + cout)
    x = torch.randn(P, cin, generator=g)
    w = torch.randn(cout, cin, generator=g) / cin ** 0.5
    b = torch.randn(cout, generator=g)
    xd = x.to(dtype)
    wr = w.to(dtype).double() if dtype == torch.bfloat16 else w.double()
    ref = xd.double() @ wr.t() + b.double()
    ref = torch.relu(ref) if act == 1 else torch.sigmoid(ref) if act == 2 else ref
    y = ops.conv1x1_fwd(xd.to(dev), w.to(dev), b.to(dev), act).float().cpu()
    scale = ref.abs().max().item()
    assert (y.double() - ref).abs().max().item() <= tol * max(scale, 1.0)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-6), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("P,cin,cout", [(1024, 64, 128), (1000, 128, 64), (37, 64, 12), (256, 12, 128), (2048, 64, 256),
                                        (1536, 64, 512),    # 512 outputs: mix_head_B at d = 128 (BASELINE configs[3])
                                        # 12-channel heads on whole 64-row tiles (flat staging of the gradient rows), and narrow
                                        # layers whose INPUT rows are not 16-byte multiples (must stay on the generic staging)
                                        (8192, 64, 12), (640, 32, 12), (128, 4, 4), (192, 12, 12)])
@pytest.mark.parametrize("act", [0, 1])
def test_conv1x1_bwd(dtype, tol, P, cin, cout, act):
    from frl_hip import ops
    dev = _dev()
    g = torch.Generator().manual_seed(P * 3 + cin + cout)
    x = torch.randn(P, cin, generator=g).to(dtype)
    w = torch.randn(cout, cin, generator=g) / cin ** 0.5
    dy = torch.randn(P, cout, generator=g).to(dtype)
    y = torch.randn(P, cout, generator=g).to(dtype)  # stand-in activation output for the relu mask
    dyd = dy.double() * ((y.double() > 0).double() if act == 1 else 1.0)
    wr = w.to(dtype).double() if dtype == torch.bfloat16 else w.double()
    ref_dx = dyd @ wr
    ref_dw = dyd.t() @ x.double()
    ref_db = dyd.sum(0)
    yd = y.to(dev) if act else None
    dx = ops.conv1x1_bwd_data(dy.to(dev), w.to(dev), yd, act).float().cpu()
    assert (dx.double() - ref_dx).abs().max().item() <= tol * max(ref_dx.abs().max().item(), 1.0)
    for scalar in (False, True):
        dw, db = ops.conv1x1_bwd_weight(dy.to(dev), x.to(dev), yd, act, scalar_frags=scalar)
        wtol = 2e-5 if dtype == torch.float32 else 2e-5  # inputs exact in both modes, f32 accumulation over P
        assert (dw.cpu().double() - ref_dw).abs().max().item() <= wtol * max(ref_dw.abs().max().item(), 1.0) * 4
        assert (db.cpu().double() - ref_db).abs().max().item() <= wtol * max(ref_db.abs().max().item(), 1.0) * 4


def _vq_case(N, K, d, dtype, seed, ties=False):
    g = torch.Generator().manual_seed(seed)
    z = torch.randn(N, d, generator=g)
    e = torch.randn(K, d, generator=g)
    if ties and K > 8 and N > 8:
        e[K - 1] = e[3]
        z[5] = 0.0
        e[7] = -e[2]
        z[6] = e[4]  # exact hit
    zt, et = z.to(dtype), e
    e_eff = e.to(dtype).float() if dtype == torch.bfloat16 else e
    idx = torch.from_numpy(O.vq_argmin_np(zt.float().numpy(), e_eff.numpy()))
    return zt, et, e_eff, idx


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,K,d,ties", [(4096, 256, 64, True), (1000, 16, 8, True), (8192, 512, 64, False),
                                        (2048, 1024, 64, False), (2048, 640, 128, False), (777, 100, 12, True),
                                        (64, 16, 4, False),
                                        # >= 65536 rows with d <= 64: 16-wave workgroups (one codebook copy per CU), chunks up to 1024 codes
                                        (65536 + 123, 512, 64, True), (66000, 1024, 64, False), (65600, 1100, 64, False), (65536, 256, 32, True),
                                        # BASELINE configs[3] codebook (K = 8192, d = 128: MFMA-bound, 32 LDS chunks) at a reduced row count
                                        (4096, 8192, 128, False)])
def test_vq_assign_bit_exact(dtype, N, K, d, ties):
    from frl_hip import ops
    dev = _dev()
    zt, et, e_eff, idx_ref = _vq_case(N, K, d, dtype, seed=N + K + d, ties=ties)
    idx, zq, stats, counts = ops.vq_assign(zt.to(dev), et.to(dev))
    idx = idx.cpu().long()
    assert torch.equal(idx, idx_ref), f"{(idx != idx_ref).sum().item()} mismatching indices"
    zq_ref = e_eff[idx_ref].to(dtype)
    assert torch.equal(zq.cpu(), zq_ref)
    sq_ref = ((zt.double() - zq_ref.double()) ** 2).sum().item()
    assert abs(stats[0].item() - sq_ref) <= 1e-5 * sq_ref
    cnt_ref = torch.bincount(idx_ref, minlength=K)
    assert torch.equal(counts.cpu().long(), cnt_ref)
    p = cnt_ref.double() / N
    perp = torch.exp(-(p * torch.log(p + 1e-10)).sum()).item()
    assert abs(stats[1].item() - perp) <= 1e-4 * perp
    assert stats[2].item() < 0.05 * N + 8  # few rows needed the float64 path
    prep = ops.vq_prepare(et.to(dev), N, dtype)                           # prepared codebook image: identical results, bit for bit
    for _ in range(2):                                                   # (twice: the workspace header is re-armed every call)
        idx2, zq2, stats2, counts2 = ops.vq_assign(zt.to(dev), et.to(dev), prep)
        assert torch.equal(idx2.cpu().long(), idx_ref) and torch.equal(zq2, zq) and torch.equal(counts2, counts)
        assert torch.equal(stats2, stats)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,K,d,dup", [(66000, 512, 64, 2), (4096, 512, 64, 2), (3000, 64, 64, 64), (70000, 256, 32, 4), (2048, 1024, 64, 2)])
def test_vq_assign_every_row_is_a_tie(dtype, N, K, d, dup):
    """Worst case of the exact re-evaluation: the codebook holds every vector `dup` times, so EVERY row ties between `dup` codes
    (dup = K: all codes identical) -- the first index must win everywhere.  Exercises the in-kernel list drain (far more parked rows
    than the list holds), the pending-candidate flush (more than two candidates per lane) and, for K = 1024, the multi-chunk path.
    Also checks that a prepared codebook image gives the same result as the one-call entry point."""
    from frl_hip import ops
    dev = _dev()
    g = torch.Generator().manual_seed(N + K + dup)
    base = torch.randn(K // dup, d, generator=g)
    e = base.repeat_interleave(dup, dim=0).contiguous()                 # codes j*dup .. j*dup+dup-1 are identical
    z = torch.randn(N, d, generator=g).to(dtype)
    e_eff = e.to(dtype).float() if dtype == torch.bfloat16 else e
    ref = torch.from_numpy(O.vq_argmin_np(z.float().numpy(), e_eff[::dup].contiguous().numpy())) * dup
    idx, zq, stats, counts = ops.vq_assign(z.to(dev), e.to(dev))
    assert torch.equal(idx.cpu().long(), ref), f"{(idx.cpu().long() != ref).sum().item()} mismatching indices"
    assert torch.equal(zq.cpu(), e_eff[ref].to(dtype))
    assert torch.equal(counts.cpu().long(), torch.bincount(ref, minlength=K))
    assert stats[2].item() == N                                          # every row went through the float64 path
    sq_ref = ((z.double() - e_eff[ref].double()) ** 2).sum().item()
    assert abs(stats[0].item() - sq_ref) <= 1e-5 * sq_ref
    prep = ops.vq_prepare(e.to(dev), N, dtype)
    idx2, zq2, stats2, counts2 = ops.vq_assign(z.to(dev), e.to(dev), prep)
    assert torch.equal(idx2, idx) and torch.equal(zq2, zq) and torch.equal(counts2, counts) and torch.equal(stats2, stats)


@pytest.mark.parametrize("N,K,case", [(4096, 512, "randn"), (2048, 256, "ties"), (1024, 200, "randn"), (256, 16, "ties"), (1536, 512, "dup2"),
                                      (512, 64, "dup64"), (262144, 512, "live"), (8192, 512, "nan"),
                                      # uneven hand-out: 300 batches on 256 workgroups (one or two per workgroup), 768 batches (three: an odd count)
                                      (76800, 512, "randn"), (196608, 256, "ties")])
def test_vq_assign_streaming_kernel_matches_resident(N, K, case):
    """The streaming assignment kernel (one 16-wave workgroup per CU, wave-local exact re-evaluation; bf16 rows of 64 channels, whole
    batches) against the float64 arg-min and against the resident kernel of round 2: indices, z_q, histogram and the number of rows
    re-evaluated are identical for every tile count; the squared-error sum differs by float32 summation order only.  Cases: plain
    randn, constructed ties (duplicate code, sign-symmetric pair, exact hit), a codebook that is not a whole key group (K = 200:
    13 code blocks), every row a tie between 2 / 64 duplicates (in-lane candidate flush), a live codebook at the full configs[1] row
    count, rows of NaN / inf."""
    from frl_hip import ops
    dev = _dev()
    d, dtype = 64, torch.bfloat16
    g = torch.Generator().manual_seed(N + K)
    z = torch.randn(N, d, generator=g)
    if case.startswith("dup"):
        dup = int(case[3:])
        e = torch.randn(K // dup, d, generator=g).repeat_interleave(dup, dim=0).contiguous()
    elif case == "live":
        e = z[torch.randperm(N, generator=g)[:K]].to(dtype).float().contiguous()
    else:
        e = torch.randn(K, d, generator=g)
    if case == "ties":
        e[K - 1] = e[3]; z[5] = 0.0; e[7] = -e[2]; z[6] = e[4]
    if case == "nan":
        z[17] = float("nan"); z[300, 3] = float("inf"); z[4097] = float("-inf")
    zt = z.to(dtype)
    e_eff = e.to(dtype).float()
    zd, ed = zt.to(dev), e.to(dev)
    prev = ops.vq_stream_tiles(0)
    try:
        ref = ops.vq_assign(zd, ed)
        if case != "nan":
            if case.startswith("dup"):
                want = torch.from_numpy(O.vq_argmin_np(zt.float().numpy(), e_eff[::dup].contiguous().numpy())) * dup
            else:
                want = torch.from_numpy(O.vq_argmin_np(zt.float().numpy(), e_eff.numpy()))
            assert torch.equal(ref[0].cpu().long(), want)
        for nt in (1, 2, 4):
            ops.vq_stream_tiles(nt)
            if N % (256 * nt):
                continue
            prep = ops.vq_prepare(ed, N, dtype)
            for _ in range(2):                                           # (twice: the control block re-arms itself)
                idx, zq, stats, counts = ops.vq_assign(zd, ed, prep)
                assert torch.equal(idx, ref[0]), f"nt={nt}: {(idx != ref[0]).sum().item()} mismatching indices"
                assert torch.equal(zq.view(torch.int16), ref[1].view(torch.int16)) and torch.equal(counts, ref[3])
                assert stats[2].item() == ref[2][2].item()
                if case != "nan":
                    assert abs(stats[0].item() - ref[2][0].item()) <= 2e-6 * ref[2][0].item()
                    assert abs(stats[1].item() - ref[2][1].item()) <= 1e-6 * ref[2][1].item()
    finally:
        ops.vq_stream_tiles(prev)


def test_vq_golden_fixture(golden_dir):
    from frl_hip import ops
    dev = _dev()
    fx = np.load(f"{golden_dir}/vq_seed7.npz")
    z, e = torch.from_numpy(fx["z"]), torch.from_numpy(fx["e"])
    idx, zq, stats, counts = ops.vq_assign(z.to(dev), e.to(dev))
    assert np.array_equal(idx.cpu().numpy().astype(np.int64), fx["idx"])
    n, d = z.shape
    mse = stats[0].item() / (n * d)
    assert abs(mse * 1.25 - float(fx["vq_loss"])) <= 1e-5 * float(fx["vq_loss"])
    assert abs(stats[1].item() - float(fx["perplexity"])) <= 1e-4 * float(fx["perplexity"])
    # backward: g_z, g_E
    gout = torch.from_numpy(fx["gout"]).float()
    gz, ge, sums = ops.vq_bwd(gout.to(dev), z.to(dev), e.to(dev), idx, counts, None, 0.25, want_sums=True)
    assert np.abs(gz.cpu().numpy() - fx["grad_z"]).max() <= 1e-5 * max(1.0, np.abs(fx["grad_z"]).max())
    assert np.abs(ge.cpu().numpy() - fx["grad_e"]).max() <= 1e-5 * max(1e-3, np.abs(fx["grad_e"]).max())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_vq_bwd_and_ema(dtype):
    from frl_hip import ops
    dev = _dev()
    N, K, d = 3000, 96, 64
    zt, et, e_eff, idx_ref = _vq_case(N, K, d, dtype, seed=3)
    g = torch.Generator().manual_seed(9)
    gout = torch.randn(N, d, generator=g).to(dtype)
    idx, zq, stats, counts = ops.vq_assign(zt.to(dev), et.to(dev))
    gs = torch.tensor([0.7, 0.7], device=dev)
    gz, ge, sums = ops.vq_bwd(gout.to(dev), zt.to(dev), et.to(dev), idx, counts, gs, 0.25, want_sums=True)
    zq_ref = e_eff[idx_ref].double()
    gz_ref = gout.double() + 0.7 * 0.25 * 2 / (N * d) * (zt.double() - zq_ref)
    tol = 1e-6 if dtype == torch.float32 else 1e-2
    assert (gz.float().cpu().double() - gz_ref).abs().max().item() <= tol * gz_ref.abs().max().item()
    sums_ref = torch.zeros(K, d, dtype=torch.float64).index_add_(0, idx_ref, zt.double())
    cnt = torch.bincount(idx_ref, minlength=K).double()
    ge_ref = 0.7 * 2 / (N * d) * (cnt[:, None] * e_eff.double() - sums_ref)
    assert (sums.cpu().double() - sums_ref).abs().max().item() <= 1e-5 * sums_ref.abs().max().item()
    assert (ge.cpu().double() - ge_ref).abs().max().item() <= 1e-4 * ge_ref.abs().max().item()
    # EMA
    ema_c = torch.rand(K, generator=g) * 10
    ema_s = torch.randn(K, d, generator=g)
    cb_ref, c_ref, s_ref = O.vq_ema_update(et.double(), ema_c.double(), ema_s.double(), zt.double(), idx_ref, 0.99, 1e-5)
    cb, ec, es = et.clone().to(dev), ema_c.to(dev), ema_s.to(dev)
    ops.vq_ema_update(sums, counts, ec, es, cb, 0.99, 1e-5)
    assert (ec.cpu().double() - c_ref).abs().max().item() <= 1e-5 * c_ref.abs().max().item()
    assert (es.cpu().double() - s_ref).abs().max().item() <= 1e-5 * s_ref.abs().max().item()
    assert (cb.cpu().double() - cb_ref).abs().max().item() <= 1e-4 * cb_ref.abs().max().item()


@pytest.mark.gpu
def test_hip_adamw_clip_matches_torch():
    """csrc/optim.hip (two launches) vs clip_grad_norm_ + torch.optim.AdamW over 4 steps, incl. a parameter without gradient."""
    from frl_hip.training.optim import HipAdamW
    DEV = "cuda:0"
    g = torch.Generator().manual_seed(3)
    shapes = [(64, 64, 3), (64,), (512, 64), (5000,), (12, 64, 1, 1), (7,)]
    ref = [torch.nn.Parameter(torch.randn(*s, generator=g)) for s in shapes]
    dev = [torch.nn.Parameter(p.detach().clone().to(DEV)) for p in ref]
    groups = lambda ps: [{"params": ps[:2] + ps[3:], "weight_decay": 0.01}, {"params": [ps[2]], "weight_decay": 0.0}]
    o_ref = torch.optim.AdamW(groups(ref), lr=3e-3, betas=(0.9, 0.95))
    o_dev = HipAdamW(groups(dev), lr=3e-3, betas=(0.9, 0.95))
    order = [0, 1, 3, 4, 5, 2]                                  # HipAdamW keeps group order
    for step in range(4):
        scale = [5.0, 0.01, 1.0, 30.0][step]                     # clip active / inactive
        for i, (pr, pd) in enumerate(zip(ref, dev)):
            if i == 5 and step % 2 == 1:
                pr.grad, pd.grad = None, None                    # parameter unused this step
                continue
            gr = torch.randn(*shapes[i], generator=g) * scale
            pr.grad, pd.grad = gr.clone(), gr.clone().to(DEV)
        for grp_r, grp_d in zip(o_ref.param_groups, o_dev.param_groups):
            grp_r["lr"] = grp_d["lr"] = 3e-3 * (1.0 - 0.1 * step)
        n_ref = torch.nn.utils.clip_grad_norm_(ref, 1.0)
        o_ref.step()
        n_dev = o_dev.step(1.0)
        assert abs(n_dev.item() - n_ref.item()) <= 1e-5 * n_ref.item()
        for pr, pd in zip(ref, dev):
            assert (pd.detach().cpu() - pr.detach()).abs().max().item() <= 2e-6
    assert [dev[i] is p for i, p in zip(order, o_dev.params)] == [True] * 6


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("K,d,N", [(64, 64, 1000), (512, 64, 4096), (37, 12, 77), (8192, 128, 300)])
def test_vq_revive_dead_codes_matches_oracle(dtype, K, d, N):
    from frl_hip import ops
    dev = _dev()
    g = torch.Generator().manual_seed(K + N)
    cb = torch.randn(K, d, generator=g)
    z = torch.randn(N, d, generator=g).to(dtype)
    window = torch.randint(0, 4, (K,), generator=g, dtype=torch.int64)
    m, v = torch.rand(K, d, generator=g), torch.rand(K, d, generator=g)
    seed = (1 << 63) + 12345                                            # exercises the 64-bit wrap of seed + k
    want, dead = O.revive_dead_codes_np(cb.numpy(), window.numpy(), 2, z.float().numpy(), seed)
    cb_d, m_d, v_d = cb.to(dev), m.to(dev), v.to(dev)
    revived = ops.vq_revive_dead_codes(cb_d, window.to(dev), 2, z.to(dev), seed, m_d, v_d)
    assert int(revived.item()) == int(dead.sum()) and 0 < dead.sum() < K
    assert np.array_equal(cb_d.cpu().numpy().view(np.uint32), want.view(np.uint32))          # bit-exact
    dm = torch.from_numpy(dead)
    assert not m_d.cpu()[dm].any() and not v_d.cpu()[dm].any()
    assert torch.equal(m_d.cpu()[~dm], m[~dm]) and torch.equal(v_d.cpu()[~dm], v[~dm])
    again = ops.vq_revive_dead_codes(cb_d, torch.full((K,), 5, dtype=torch.int64, device=dev), 2, z.to(dev), seed, revived=revived)
    assert int(again.item()) == int(dead.sum())                        # nothing dead: nothing touched, counter unchanged
    assert np.array_equal(cb_d.cpu().numpy().view(np.uint32), want.view(np.uint32))


def test_codebook_manager_revives_unused_codes_during_training():
    from frl_hip.models import VQVAE
    from frl_hip.training.codebook_manager import CodebookManager
    from frl_hip.training.trainer import VQVAETrainer
    dev = _dev()
    torch.manual_seed(0)
    m = VQVAE(in_features=64, codebook_size=64, emb_dim=64, type_encoder_dropout=0.0, phase_tcn_dropout=0.0,
              compute_dtype=torch.bfloat16).to(dev)
    with torch.no_grad():                                               # half of the codes far away from every encoder output
        m.quant.codebook[32:] += 1000.0
    mgr = CodebookManager(num_codes=m.quant.codebook_size, code_dim=m.quant.emb_dim, reset_every=2, min_count=1, seed=3)
    m.attach_codebook_manager(mgr)
    tr = VQVAETrainer(m, lr=1e-3, total_steps=10)
    tiles = torch.randn(2, 5, 32, 32, 64, device=dev).bfloat16()
    tr.step(tiles)
    assert int(mgr.revived.item()) == 0 and int(mgr.window.sum().item()) == 2 * 32 * 32
    assert m.quant.codebook[32:].abs().min().item() > 500
    tr.step(tiles)                                                      # second step: the window closes, dead codes are re-seeded
    assert int(mgr.revived.item()) >= 32 and int(mgr.window.sum().item()) == 0
    assert m.quant.codebook.abs().max().item() < 100                    # the far-away codes are gone
    out = tr.step(tiles)
    assert torch.isfinite(out["loss"]).item()
    idx = m.forward_tiles(tiles)["idx"]
    assert (idx >= 32).any().item()                                     # revived codes are in use again


def test_loss_head_matches_torch_arithmetic_and_flags_non_finite_values():
    """frl_scalar_combine / frl_scalar_fanout: the weighted sum of the loss terms, its isfinite flag and the term gradients in one launch
    each (scripts/train_vqvae.py:236-248, step.py:1057-1074) against plain torch arithmetic."""
    from frl_hip import functional as Fh
    dev = _dev()
    vals, coefs = [0.75, 2.5, -1.25, 3.0], [1.0, 0.25, 2.0, 0.5]
    terms = [torch.tensor(v, device=dev, requires_grad=True) for v in vals]
    loss, ok = Fh.scalar_combine(terms, coefs)
    ref = sum(c * v for c, v in zip(coefs, vals))
    assert abs(loss.item() - ref) <= 1e-6 * abs(ref) and ok.item() == 1.0
    (3.0 * loss).backward()
    for t, c in zip(terms, coefs):
        assert abs(t.grad.item() - 3.0 * c) <= 1e-6
    for bad in (float("nan"), float("inf"), -float("inf")):
        loss, ok = Fh.scalar_combine([torch.tensor(1.0, device=dev), torch.tensor(bad, device=dev)], [1.0, 0.5])
        assert ok.item() == 0.0 and not torch.isfinite(loss).item()
    # off the GPU path the same call is plain torch arithmetic without a flag
    loss, ok = Fh.scalar_combine([torch.tensor(2.0), torch.tensor(3.0)], [1.0, 0.5])
    assert ok is None and loss.item() == 3.5
