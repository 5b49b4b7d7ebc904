"""CPU-side checks: the C-ABI library exports every symbol of include/frl_hip.h with matching arity, the product
package fails loudly without a GPU, and the oracle reproduces the golden vectors generated from the reference."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

import frl_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_decls():
    txt = open(os.path.join(ROOT, "include", "frl_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(?:int|size_t|const char\*)\s+(frl_\w+)\s*\(([^;]*?)\)\s*;", txt, flags=re.S):
        args = m.group(2).strip()
        n = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
        decls[m.group(1)] = n
    return decls


def test_header_symbols_exported_and_signatures_match():
    from frl_hip import _lib
    decls = _header_decls()
    assert len(decls) >= 35
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name, nargs in decls.items():
        assert hasattr(lib, name), f"{name} declared in frl_hip.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} missing from the ctypes table"
        assert len(_lib.SIGNATURES[name][1]) == nargs, f"{name}: header has {nargs} args, ctypes table {len(_lib.SIGNATURES[name][1])}"
    for name in _lib.SIGNATURES:
        assert name in decls, f"{name} bound in Python but not declared in frl_hip.h"
    assert lib.frl_version() >= 100


def test_product_path_has_no_cpu_fallback():
    from frl_hip import ops
    x = torch.randn(16, 8)
    w = torch.randn(4, 8)
    with pytest.raises(Exception):
        ops.conv1x1_fwd(x, w, None)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "vq-vae_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".hpp")):
                src = open(os.path.join(dp, fn)).read()
                assert "frl_oracle" not in src and "import oracle" not in src, f"{fn} references the oracle"


def _state(fx, dtype=torch.float64):
    return {k[6:]: torch.from_numpy(fx[k]).to(dtype) for k in fx.files if k.startswith("state.")}


@pytest.mark.parametrize("name,hp", [("tiny_seed0", dict(type_encoder_num_groups=4, phase_tcn_num_groups=4)),
                                     ("tiny_seed1", dict(type_encoder_num_groups=4, phase_tcn_num_groups=4)),
                                     ("full_seed0", dict())])
def test_oracle_matches_reference_golden(golden_dir, name, hp):
    fx = np.load(os.path.join(golden_dir, name + ".npz"))
    dt = torch.float64
    sd = _state(fx, dt)
    tile = torch.from_numpy(fx["tile"]).to(dt)
    tol = 1e-12 if name.startswith("tiny") else 2e-6  # full fixture is stored in float32
    x_type, x_phase = O.tile_to_inputs(tile)
    z, gate, h = O.model_forward(sd, x_type, hp)
    zp = O.model_forward_phase(sd, x_phase, z, hp)
    for got, key in ((h, "h"), (z, "z_type"), (gate, "gate"), (zp, "z_phase")):
        assert np.abs(got.numpy() - fx[key]).max() <= tol, key
    yx = fx["loc_yx"]
    xpx = x_phase[0][:, :, yx[:, 0], yx[:, 1]].permute(2, 0, 1)
    zpx = torch.from_numpy(fx["z_type"]).to(dt)[0][:, yx[:, 0], yx[:, 1]].permute(1, 0)
    zl, gam, bet, hpre = O.model_forward_phase_at_locations(sd, xpx, zpx, hp)
    assert np.abs(zl.numpy() - fx["loc_z"]).max() <= tol
    assert np.abs(gam.numpy() - fx["loc_gamma"]).max() <= tol


def test_oracle_mingate_and_projection(golden_dir):
    fx = np.load(os.path.join(golden_dir, "tiny_mingate_proj.npz"))
    sd = _state(fx)
    hp = dict(type_encoder_num_groups=4, phase_tcn_num_groups=4, min_gate=float(fx["min_gate"]))
    x_type, _ = O.tile_to_inputs(torch.from_numpy(fx["tile"]))
    z, gate, _ = O.model_forward(sd, x_type, hp)
    assert np.abs(z.numpy() - fx["z_type"]).max() <= 1e-12
    assert gate.min().item() >= float(fx["min_gate"]) - 1e-12
    t = O.tcn_forward(sd, torch.from_numpy(fx["xp16"]), (1, 2, 4), 4)
    assert np.abs(t.numpy() - fx["tcn_out"]).max() <= 1e-12


def test_oracle_gradients_match_reference(golden_dir):
    fx = np.load(os.path.join(golden_dir, "tiny_seed0.npz"))
    sd = _state(fx)
    hp = dict(type_encoder_num_groups=4, phase_tcn_num_groups=4)
    leaf = {k: v.clone().requires_grad_(k not in O.FIXED_BUFFERS) for k, v in sd.items()}
    x_type, x_phase = O.tile_to_inputs(torch.from_numpy(fx["tile"]))
    z, _, _ = O.model_forward(leaf, x_type, hp)
    zp = O.model_forward_phase(leaf, x_phase, z.detach(), hp)
    loss = z.pow(2).mean() + zp.pow(2).mean()
    loss.backward()
    assert abs(loss.item() - float(fx["loss"])) < 1e-12
    for k in fx.files:
        if k.startswith("grad.") and k[5:] in leaf:
            assert np.abs(leaf[k[5:]].grad.numpy() - fx[k]).max() <= 1e-12, k


def test_oracle_vq_definition(golden_dir):
    fx = np.load(os.path.join(golden_dir, "vq_seed7.npz"))
    z, e = fx["z"], fx["e"]
    idx = O.vq_argmin_np(z, e)
    assert np.array_equal(idx, fx["idx"])
    # brute force in float64 with the expanded form agrees away from constructed ties
    d2 = (z.astype(np.float64) ** 2).sum(1)[:, None] - 2 * z.astype(np.float64) @ e.astype(np.float64).T + (e.astype(np.float64) ** 2).sum(1)[None]
    alt = d2.argmin(1)
    assert (alt != idx).sum() <= 4
    # constructed ties resolve to the first index
    assert idx[5] == min(np.flatnonzero(np.isclose(((z[5][None] - e) ** 2).sum(1), ((z[5][None] - e) ** 2).sum(1).min(), rtol=0, atol=0)))
    zz = torch.from_numpy(z).double().requires_grad_(True)
    ee = torch.from_numpy(e).double().requires_grad_(True)
    z_st, vq_loss, perp, _, lcb, lcm = O.vq_forward(zz, ee, 0.25)
    assert abs(vq_loss.item() - float(fx["vq_loss"])) < 1e-12
    assert abs(lcb.item() - lcm.item()) < 1e-15
    (vq_loss + (z_st * torch.from_numpy(fx["gout"])).sum()).backward()
    assert np.abs(zz.grad.numpy() - fx["grad_z"]).max() < 1e-12
    assert np.abs(ee.grad.numpy() - fx["grad_e"]).max() < 1e-12


def test_vq_straight_through_gradients_closed_form():
    """Stop-gradients make numerical gradcheck of the whole loss meaningless; check each detached term numerically
    and the assembled gradient against the closed forms of SURVEY.md section 8a row a11."""
    torch.manual_seed(0)
    z = torch.randn(12, 4, dtype=torch.float64, requires_grad=True)
    e = torch.randn(5, 4, dtype=torch.float64, requires_grad=True)
    idx = torch.from_numpy(O.vq_argmin_np(z.detach().numpy(), e.detach().numpy()))
    zc, ec = z.detach().clone(), e.detach().clone()
    assert torch.autograd.gradcheck(lambda zz: ((zz - ec[idx]) ** 2).mean(), (z,), eps=1e-6, atol=1e-7)   # L_commit
    assert torch.autograd.gradcheck(lambda ee: ((zc - ee[idx]) ** 2).mean(), (e,), eps=1e-6, atol=1e-7)   # L_codebook
    gout = torch.randn(12, 4, dtype=torch.float64)
    z_st, vq_loss, *_ = O.vq_forward(z, e, 0.25, idx=idx)
    (vq_loss + (z_st * gout).sum()).backward()
    n, d = zc.shape
    gz = gout + 0.25 * 2 * (zc - ec[idx]) / (n * d)
    ge = torch.zeros_like(ec).index_add_(0, idx, 2 * (ec[idx] - zc) / (n * d))
    assert (z.grad - gz).abs().max() < 1e-14 and (e.grad - ge).abs().max() < 1e-14
    assert (z_st.detach() - ec[idx]).abs().max() < 1e-14  # forward value is z_q


def test_oracle_schedules_match_reference(golden_dir):
    sched = json.load(open(os.path.join(golden_dir, "schedules.json")))
    from frl_hip.training import schedules as S
    for e_, s, r, want in sched["ramp_weight"]:
        assert S.ramp_weight(e_, s, r) == want
    for e_, f, r, want in sched["min_gate"]:
        assert S.compute_smoothing_min_gate(e_, f, r) == want
    for e_, want in sched["input_dropout_linear"]:
        assert abs(S.compute_input_dropout_rate(dict(schedule="linear", start=0.0, end=0.1, epochs=20), e_, 200) - want) < 1e-15
    for e_, want in sched["input_dropout_cosine"]:
        assert abs(S.compute_input_dropout_rate(dict(schedule="cosine", start=0.02, end=0.2, epochs=10), e_, 200) - want) < 1e-15
    assert S.compute_input_dropout_rate(0.05, 3, 10) == sched["input_dropout_const"][0][1]


def test_oracle_vqvae_step_fixture(golden_dir):
    fx = np.load(os.path.join(golden_dir, "vqvae_tiny_seed0.npz"))
    sd = _state(fx)
    hp = dict(type_encoder_num_groups=4, phase_tcn_num_groups=4, beta=0.25)
    tiles = torch.from_numpy(fx["tiles"])
    outs, grads = O.vqvae_loss_and_grads(sd, tiles[0], hp)
    assert abs(float(outs["loss"]) - float(fx["loss"])) < 1e-12
    assert np.array_equal(outs["idx"].numpy(), fx["idx"])
    tr = O.OracleTrainer(sd, hp, lr=1e-3, total_steps=10)
    traj = [float(tr.step(tiles[i])["loss"]) for i in range(3)]
    assert np.abs(np.asarray(traj) - fx["traj"]).max() < 1e-10


@pytest.mark.parametrize("case", ["a", "b", "c", "d"])
def test_oracle_mutual_knn_equals_reference_output(golden_dir, case):
    """tests/golden/mutual_knn_*.npz: pairs returned by the reference's pairs_mutual_knn_chunked (oracle/make_pairs_golden.py)."""
    import frl_oracle as O
    fx = np.load(os.path.join(golden_dir, f"mutual_knn_{case}.npz"))
    offsets = fx["offsets"].tolist()
    coords = [fx["coords"][offsets[p]:offsets[p + 1]] for p in range(len(offsets) - 1)]
    pairs, knn = O.mutual_knn_pairs_np(fx["features"], coords, offsets, int(fx["k"]), float(fx["min_sp"]))
    assert np.array_equal(pairs, fx["pairs"])
    have = set(map(tuple, pairs.tolist()))
    assert all((j, i) in have for i, j in have)                           # both directions of every mutual pair


@pytest.mark.parametrize("case", ["a", "b", "c", "d"])
def test_oracle_contrastive_loss_equals_reference_output(golden_dir, case):
    """tests/golden/contrastive_*.npz: loss and gradient of the reference's contrastive_loss in float64 (oracle/make_contrastive_golden.py)."""
    import frl_oracle as O
    fx = np.load(os.path.join(golden_dir, f"contrastive_{case}.npz"))
    emb = torch.from_numpy(fx["emb"]).requires_grad_(True)
    pw = torch.from_numpy(fx["pw"]) if "pw" in fx.files else None
    nw = torch.from_numpy(fx["nw"]) if "nw" in fx.files else None
    loss = O.contrastive_loss_oracle(emb, torch.from_numpy(fx["pos"]), torch.from_numpy(fx["neg"]), pw, nw, float(fx["t"]), str(fx["sim"]))
    loss.backward()
    assert abs(float(loss.detach()) - float(fx["loss64"])) < 1e-12 * max(1.0, abs(float(fx["loss64"])))
    assert np.abs(emb.grad.numpy() - fx["grad64"]).max() < 1e-12 * max(1.0, np.abs(fx["grad64"]).max())


def test_oracle_extract_at_locations(golden_dir):
    import frl_oracle as O
    fx = np.load(os.path.join(golden_dir, "extract_locations.npz"))
    assert np.array_equal(O.extract_at_locations_np(fx["feat"], fx["coords"]), fx["out"])
