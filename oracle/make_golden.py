"""Generate tests/golden/* by importing the reference model (BUILD CONTAINER ONLY).

Run:  python oracle/make_golden.py            (needs /root/reference; CPU only)

The reference's Python never travels to the GPU box: only the small .npz/.json
fixtures written here do.  The script also asserts that the oracle restatement
(oracle/frl_oracle.py) agrees with the imported reference in float64 to 1e-12,
which is what "encoder parity pinned" means in DESIGN.md.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/frl"
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import frl_oracle as O  # noqa: E402
from models import RepresentationModel  # noqa: E402  (reference import)
from training.representation import curriculum as ref_curr  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
os.makedirs(GOLD, exist_ok=True)

TINY_KW = dict(type_in_channels=8, phase_in_channels=8, z_type_dim=8, z_phase_dim=4,
               type_encoder_channels=(16, 8), type_encoder_dropout=0.0, type_encoder_num_groups=4,
               spatial_conv_gate_hidden=8, phase_tcn_channels=(8, 8, 8), phase_tcn_dropout=0.0,
               phase_tcn_num_groups=4)
TINY_HP = dict(type_encoder_num_groups=4, phase_tcn_num_groups=4, phase_tcn_dilations=(1, 2, 4))
FULL_HP = dict(type_encoder_num_groups=8, phase_tcn_num_groups=8, phase_tcn_dilations=(1, 2, 4))


def maxdiff(a, b):
    return float((a - b).abs().max())


def run_case(model, tile, hp, n_loc=7, with_grads=True):
    """Runs reference + oracle in float64; returns dict of numpy arrays."""
    model = model.double()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    x_type, x_phase = O.tile_to_inputs(tile)
    x_type = x_type.contiguous().requires_grad_(with_grads)
    x_phase = x_phase.contiguous().requires_grad_(with_grads)
    h_ref = model.encoder(x_type)
    z_ref, gate_ref = model(x_type, return_gate=True)
    zp_ref = model.forward_phase(x_phase, z_ref.detach())
    # oracle vs reference
    z_o, gate_o, h_o = O.model_forward(sd, x_type.detach(), hp)
    zp_o = O.model_forward_phase(sd, x_phase.detach(), z_o, hp)
    errs = dict(h=maxdiff(h_o, h_ref), z=maxdiff(z_o, z_ref), gate=maxdiff(gate_o, gate_ref),
                zp=maxdiff(zp_o, zp_ref))
    # sparse path
    b, c, t, hh, ww = x_phase.shape
    g = torch.Generator().manual_seed(5)
    ys = torch.randint(0, hh, (n_loc,), generator=g)
    xs = torch.randint(0, ww, (n_loc,), generator=g)
    xpx = x_phase.detach()[0][:, :, ys, xs].permute(2, 0, 1).contiguous()  # [N,C,T]
    zpx = z_ref.detach()[0][:, ys, xs].permute(1, 0).contiguous()  # [N,d]
    zl_ref, gam_ref, bet_ref, hpre_ref = model.forward_phase_at_locations(xpx, zpx, True, True)
    zl_o, gam_o, bet_o, hpre_o = O.model_forward_phase_at_locations(sd, xpx, zpx, hp)
    errs.update(loc=maxdiff(zl_o, zl_ref), gam=maxdiff(gam_o, gam_ref), hpre=maxdiff(hpre_o, hpre_ref))
    dense_at = zp_ref.detach()[0][:, :, ys, xs].permute(2, 1, 0)
    errs["loc_vs_dense"] = maxdiff(dense_at, zl_ref)
    assert max(errs.values()) < 1e-11, errs
    out = dict(tile=tile.numpy(), h=h_ref.detach().numpy(), z_type=z_ref.detach().numpy(),
               gate=gate_ref.detach().numpy(), z_phase=zp_ref.detach().numpy(),
               loc_yx=torch.stack([ys, xs], 1).numpy(), loc_z=zl_ref.detach().numpy(),
               loc_gamma=gam_ref.detach().numpy(), loc_beta=bet_ref.detach().numpy(),
               loc_hpre=hpre_ref.detach().numpy())
    if with_grads:
        loss = z_ref.pow(2).mean() + zp_ref.pow(2).mean()
        model.zero_grad()
        loss.backward()
        out["loss"] = np.asarray(loss.item())
        out["grad.x_type"] = x_type.grad.numpy()
        out["grad.x_phase"] = x_phase.grad.numpy()
        # oracle autograd must agree with reference autograd
        leaf = {k: v.clone().requires_grad_(v.is_floating_point() and k not in O.FIXED_BUFFERS)
                for k, v in sd.items()}
        xt = x_type.detach().clone().requires_grad_(True)
        zo, _, _ = O.model_forward(leaf, xt, hp)
        zpo = O.model_forward_phase(leaf, x_phase.detach(), zo.detach(), hp)
        (zo.pow(2).mean() + zpo.pow(2).mean()).backward()
        for name, p in model.named_parameters():
            out["grad." + name] = p.grad.numpy()
            assert maxdiff(leaf[name].grad, p.grad) < 1e-11, name
        assert maxdiff(xt.grad, x_type.grad) < 1e-11
    for k, v in sd.items():
        out["state." + k] = v.numpy()
    return out, errs


def main():
    report = {}
    # (1) tiny configs, seeds 0 and 1, float64 fixtures
    for seed in (0, 1):
        torch.manual_seed(seed)
        m = RepresentationModel(**TINY_KW).eval()
        tile = torch.randn(2, 5, 8, 8, 8, dtype=torch.float64)
        out, errs = run_case(m, tile, TINY_HP)
        np.savez_compressed(os.path.join(GOLD, f"tiny_seed{seed}.npz"), **out)
        report[f"tiny_seed{seed}"] = errs
    # (1b) tiny config with min_gate > 0 and channel-changing TCN (projection path)
    torch.manual_seed(2)
    kw = dict(TINY_KW)
    kw.update(phase_in_channels=16, phase_tcn_channels=(8, 8, 8))
    m = RepresentationModel(**kw).eval()
    m.set_spatial_min_gate(0.55)
    tile = torch.randn(1, 5, 8, 8, 8, dtype=torch.float64)
    md = m.double()
    sd = {k: v.detach().clone() for k, v in md.state_dict().items()}
    x_type, _ = O.tile_to_inputs(tile)
    z_ref, g_ref = md(x_type.contiguous(), return_gate=True)
    hp = dict(TINY_HP, min_gate=0.55)
    z_o, g_o, _ = O.model_forward(sd, x_type, hp)
    xp16 = torch.randn(9, 16, 5, dtype=torch.float64)
    t_ref = md.phase_tcn(xp16)
    t_o = O.tcn_forward(sd, xp16, (1, 2, 4), 4)
    errs = dict(z=maxdiff(z_o, z_ref), gate=maxdiff(g_o, g_ref), tcn_proj=maxdiff(t_o, t_ref))
    assert max(errs.values()) < 1e-11, errs
    out = dict(tile=tile.numpy(), z_type=z_ref.detach().numpy(), gate=g_ref.detach().numpy(),
               min_gate=np.asarray(0.55), xp16=xp16.numpy(), tcn_out=t_ref.detach().numpy())
    for k, v in sd.items():
        out["state." + k] = v.numpy()
    np.savez_compressed(os.path.join(GOLD, "tiny_mingate_proj.npz"), **out)
    report["tiny_mingate_proj"] = errs

    # (2) full-size: class defaults, C_in=64, one 5x32x32x64 tile, eval; stored float32
    torch.manual_seed(0)
    m = RepresentationModel(64, 64).eval()
    tile = torch.randn(1, 5, 32, 32, 64).double()
    out, errs = run_case(m, tile, FULL_HP, n_loc=16, with_grads=False)
    out = {k: (v.astype(np.float32) if v.dtype == np.float64 else v) for k, v in out.items()}
    np.savez_compressed(os.path.join(GOLD, "full_seed0.npz"), **out)
    report["full_seed0"] = errs

    # (3) VQ vectors (oracle definition; reference has no quantizer -> unpinned)
    g = torch.Generator().manual_seed(7)
    z = torch.randn(2048, 64, generator=g)
    e = torch.randn(256, 64, generator=g)
    # constructed exact ties: duplicate rows and a sign-symmetric pair
    e[200] = e[13]
    z[5] = 0.0
    e[100] = -e[40]
    zz = z.double().requires_grad_(True)
    ee = e.double().requires_grad_(True)
    z_st, vq_loss, perp, idx, l_cb, l_cm = O.vq_forward(zz, ee, 0.25)
    gout = torch.randn(2048, 64, generator=g).double()
    (vq_loss + (z_st * gout).sum()).backward()
    cd = torch.cdist(z, e).argmin(1)
    np.savez_compressed(os.path.join(GOLD, "vq_seed7.npz"), z=z.numpy(), e=e.numpy(), idx=idx.numpy(),
                        vq_loss=np.asarray(vq_loss.item()), perplexity=np.asarray(perp.item()),
                        l_codebook=np.asarray(l_cb.item()), l_commit=np.asarray(l_cm.item()),
                        gout=gout.numpy(), grad_z=zz.grad.numpy(), grad_e=ee.grad.numpy(),
                        cdist_fp32_mismatch=np.asarray(int((cd != idx).sum())))
    report["vq_seed7"] = dict(cdist_fp32_mismatch=int((cd != idx).sum()))

    # (4) schedules -- pure functions of the reference (curriculum.py:16-83)
    sched = dict(
        ramp_weight=[[e_, 10, 20, ref_curr.ramp_weight(e_, 10, 20)] for e_ in (0, 9, 10, 11, 20, 29, 30, 99)],
        min_gate=[[e_, 5, 10, ref_curr.compute_smoothing_min_gate(e_, 5, 10)] for e_ in (0, 5, 7, 15, 40)],
        input_dropout_linear=[[e_, ref_curr.compute_input_dropout_rate(
            dict(schedule="linear", start=0.0, end=0.1, epochs=20), e_, 200)] for e_ in (0, 1, 10, 20, 50)],
        input_dropout_cosine=[[e_, ref_curr.compute_input_dropout_rate(
            dict(schedule="cosine", start=0.02, end=0.2, epochs=10), e_, 200)] for e_ in (0, 3, 10, 11)],
        input_dropout_const=[[3, ref_curr.compute_input_dropout_rate(0.05, 3, 10)]],
    )
    with open(os.path.join(GOLD, "schedules.json"), "w") as f:
        json.dump(sched, f, indent=1)

    # (5) oracle-only VQ-VAE step (unpinned by the reference): tiny, 3-step trajectory
    torch.manual_seed(0)
    m = RepresentationModel(**TINY_KW).eval().double()
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    sd = O.init_extra_state(sd, in_features=8, d=8, zp=4, k=16, hidden=16, seed=3, dtype=torch.float64)
    g = torch.Generator().manual_seed(11)
    sd["quant.codebook"] = torch.randn(16, 8, generator=g, dtype=torch.float64) * 0.5
    tiles = torch.randn(3, 2, 5, 8, 8, 8, generator=g, dtype=torch.float64)
    hp = dict(TINY_HP, beta=0.25)
    outs, grads = O.vqvae_loss_and_grads(sd, tiles[0], hp)
    tr = O.OracleTrainer(sd, hp, lr=1e-3, total_steps=10)
    traj = [float(tr.step(tiles[i])["loss"]) for i in range(3)]
    fx = dict(tiles=tiles.numpy(), traj=np.asarray(traj), loss=np.asarray(float(outs["loss"])),
              idx=outs["idx"].numpy(), l_type=np.asarray(float(outs["l_type"])),
              l_phase=np.asarray(float(outs["l_phase"])), vq_loss=np.asarray(float(outs["vq_loss"])),
              perplexity=np.asarray(float(outs["perplexity"])), xhat_type=outs["xhat_type"].numpy())
    for k, v in sd.items():
        fx["state." + k] = v.numpy()
    for k, v in grads.items():
        fx["grad." + k] = v.numpy()
    np.savez_compressed(os.path.join(GOLD, "vqvae_tiny_seed0.npz"), **fx)
    report["vqvae_tiny_seed0"] = dict(traj=traj)

    with open(os.path.join(GOLD, "PINNING_REPORT.json"), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
