"""Host-side logic that needs no GPU: dataset / collate contract, config loader, schedules, model construction + state-dict keys."""
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_forest_dataset_tile_layout_and_collate():
    from frl_hip.data import ForestDataset, collate_fn
    ds = ForestDataset(num_tiles=10, time=5, size=32, features=64, seed=1234, channels_first=True, partial_edge=20)
    assert len(ds) == 10
    s = ds[0]
    assert s["tile"].shape == (5, 32, 32, 64) and s["tile"].dtype == np.float32          # (time, y, x, feature)
    assert s["static"].shape == (64, 32, 32) and s["annual"].shape == (64, 5, 32, 32)     # ForestDatasetV2 group views
    assert not s["mask"][20:].any() and s["mask"][:20].all() and np.all(s["tile"][:, 20:] == 0)   # zero-padded partial patch
    assert set(s["metadata"]) == {"spatial_window", "channel_names", "patch_idx"}
    assert np.array_equal(ds[3]["tile"], ds[3]["tile"])                                   # deterministic per index
    b = collate_fn([ds[i] for i in range(4)])
    assert b["tile"].shape == (4, 5, 32, 32, 64) and isinstance(b["metadata"], list) and len(b["metadata"]) == 4
    with pytest.raises(IndexError):
        ds[10]
    assert collate_fn([ds[1]])["tile"].shape[0] == 1


def test_vae_v0_yaml_loader():
    from frl_hip.config import load_vae_config
    cfg = load_vae_config(os.path.join(ROOT, "configs", "vae_v0.yaml"))
    assert cfg.batch_size == 256 and cfg.codebook_size == 512 and cfg.emb_dim == 64 and cfg.beta == 0.25
    assert cfg.optimizer.name == "adamw" and cfg.optimizer.lr == 1e-4 and cfg.optimizer.scheduler["eta_min"] == 1e-6
    assert cfg.beta_schedule["schedule_type"] == "linear" and cfg.quantizer == "st"


def test_vae_v0_reference_key_set_is_accepted(tmp_path):
    """Every key of the reference's configs/vae_v0.yaml (flat layout) is accepted; zarr-windowing keys land in `extra`."""
    from frl_hip.config import load_vae_config
    p = tmp_path / "ref_like.yaml"
    p.write_text("zarr_path: /x\npatch_size: 256\nbatch_size: 4\nnum_epochs: 200\nbeta: 0.1\nlambda_cat: 1.0\n"
                 "optimizer:\n  name: adam\n  lr: 1e-4\n  weight_decay: 0.0\n  scheduler:\n    name: cosine\n    T_max_epochs: 150\n    eta_min: 1.0e-6\n"
                 "beta_schedule:\n  enabled: true\n  schedule_type: linear\n  start_epoch: 0\n  end_epoch: 100\n  start_value: 0.1\n  end_value: 1.0\n"
                 "debug_window: true\ndebug_window_origin: [2560, 5120]\ndebug_window_size: [1024, 1024]\ndebug_block_dims: [1, 1]\n"
                 "full_block_dims: [7, 7]\nnum_workers: 0\npin_memory: true\nrun_root: runs\nexperiment_name: vae_v0_debug\nckpt_dir: checkpoints\n")
    cfg = load_vae_config(str(p))
    assert cfg.beta == 0.1 and cfg.optimizer.lr == 1e-4 and cfg.extra["full_block_dims"] == [7, 7]
    from frl_hip.training.schedules import beta_schedule
    assert beta_schedule(0, cfg.beta_schedule) == 0.1 and beta_schedule(100, cfg.beta_schedule) == 1.0
    assert abs(beta_schedule(50, cfg.beta_schedule) - 0.55) < 1e-12


def test_schedules():
    from frl_hip.training import schedules as S
    assert S.cosine_lr(0, 100, 1e-3, 1e-5) == 1e-3 and abs(S.cosine_lr(100, 100, 1e-3, 1e-5) - 1e-5) < 1e-15
    assert S.warmup_cosine_factor(0, 10, 100, 0.01) == 1e-8 and S.warmup_cosine_factor(5, 10, 100, 0.01) == 0.5
    assert abs(S.warmup_cosine_factor(100, 10, 100, 0.01) - 0.01) < 1e-12


def test_model_api_surface_and_errors():
    from frl_hip.models import RepresentationModel, VQVAE
    m = RepresentationModel(64, 64)
    assert m.VERSION == "4" and m.type_projection is None and m.project_type(torch.ones(2, 3)).shape == (2, 3)
    for name in ("encoder", "spatial_conv", "phase_tcn", "phase_head", "phase_film"):
        assert hasattr(m, name)
    m.set_spatial_min_gate(0.3)
    m.set_input_dropout_rate(0.0)
    assert m.spatial_conv.min_gate == 0.3
    with pytest.raises(ValueError):
        RepresentationModel(8, 8, z_type_dim=8, type_encoder_channels=(16, 4))
    with pytest.raises(ValueError):
        RepresentationModel.from_config({"version": "3"}, 8, 8)
    cfg = {"version": "4", "latents": {"z_type_dim": 48, "z_phase_dim": 8},
           "type_encoder": {"channels": [128, 48], "dropout": [0.0, 0.0], "input_dropout": {"schedule": "linear", "start": 0.0, "end": 0.1, "epochs": 20}},
           "phase_tcn": {"channels": [64, 64, 64], "dilations": [1, 2, 4], "dropout": 0.0}, "type_projection": {"enabled": False}}
    m2 = RepresentationModel.from_config(cfg, 16, 8)
    assert m2.z_type_dim == 48 and m2.z_phase_dim == 8 and "encoder.layers.3.weight" in m2.state_dict()
    v = VQVAE(in_features=64, codebook_size=512, emb_dim=64)
    names = dict(v.named_parameters())
    assert "quant.codebook" in names and names["quant.codebook"].shape == (512, 64)
    assert v.quant.codebook_size == 512 and v.quant.emb_dim == 64
    v.attach_codebook_manager(object())
    with pytest.raises(Exception):       # CPU tensors are refused: HIP only
        v.forward_tiles(torch.zeros(1, 5, 32, 32, 64))


def test_checkpoint_round_trip_cpu(tmp_path):
    from frl_hip.models import RepresentationModel
    cfg = {"version": "4", "latents": {"z_type_dim": 8, "z_phase_dim": 4}, "type_encoder": {"channels": [16, 8], "dropout": 0.0, "num_groups": 4},
           "spatial_conv": {"gate_hidden": 8}, "phase_tcn": {"channels": [8, 8, 8], "dropout": 0.0, "num_groups": 4}}
    m = RepresentationModel.from_config(cfg, 8, 8)
    p = tmp_path / "encoder_last.pt"
    torch.save({"model_version": "4", "model_config": cfg, "type_in_channels": 8, "phase_in_channels": 8,
                "model_state_dict": m.state_dict(), "epoch": 3}, p)
    m2 = RepresentationModel.from_checkpoint(p, device="cpu")
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))
    assert not any(q.requires_grad for q in m2.parameters())
    torch.save({"model_version": "3"}, p)
    with pytest.raises(RuntimeError):
        RepresentationModel.from_checkpoint(p, device="cpu")


# ------------------------------------------------------------------------------------------------ tile ingest
def test_chunk_batch_sampler_reproduces_reference_batches(golden_dir):
    """tests/golden/sampler_batches.json was written by the reference's own ChunkBatchSampler (make_sampler_golden.py)."""
    import json
    from frl_hip.data.samplers import ChunkBatchSampler
    sys_path_cases = json.load(open(os.path.join(golden_dir, "sampler_batches.json")))
    for case in sys_path_cases:
        rng = np.random.default_rng(case["layout_seed"])
        chunks, nxt = [], 0
        for n in case["chunk_sizes"]:
            chunks.append(np.arange(nxt, nxt + n, dtype=np.int64))
            nxt += n
        if case.get("scramble"):
            chunks = [rng.permutation(c) for c in chunks]
        s = ChunkBatchSampler(chunks, case["batch_size"], drop_last=case["drop_last"], replacement_within_chunk=case["replacement"],
                              seed=case["seed"])
        np.random.seed(case["np_seed"])
        got = [[list(map(int, b)) for b in s] for _ in range(case["epochs"])]
        assert got == case["batches"], case["name"]
        assert len(s) == case["length"]
        if not case["replacement"]:                       # chunk-locked: a batch never mixes chunks
            owner = {int(i): c for c, m in enumerate(chunks) for i in m}
            assert all(len({owner[i] for i in b}) == 1 for ep in got for b in ep)
            if not case["drop_last"]:
                assert sorted(i for b in got[0] for i in b) == list(range(nxt))


def test_shard_batches_gives_every_rank_the_same_number_of_steps():
    from frl_hip.data.samplers import shard_batches
    batches = [[i] for i in range(11)]
    parts = [shard_batches(batches, r, 4) for r in range(4)]
    assert [len(p) for p in parts] == [2, 2, 2, 2] and sorted(b[0] for p in parts for b in p) == list(range(8))


def test_tile_store_and_chunk_dataset(tmp_path):
    from frl_hip.data.tile_loader import ChunkTileDataset
    from frl_hip.data.tile_store import TileStore, write_tile_store
    rng = np.random.default_rng(0)
    cube = rng.standard_normal((5, 70, 100, 8)).astype(np.float32)
    cube[2, 5, 6, 3] = np.nan
    write_tile_store(str(tmp_path / "s"), cube, (64, 64), dtype="float32")
    st = TileStore(str(tmp_path / "s"))
    assert st.grid == (2, 2) and st.num_chunks == 4 and st.chunk(1, 1).shape == (5, 64, 64, 8)
    assert np.isnan(st.chunk(1, 1)[:, 6:, :, :]).all()                                     # edge chunks padded with no-data
    ds = ChunkTileDataset(st, 32)
    assert [len(c) for c in ds.xy_by_chunk] == [4, 4, 2, 2] and len(ds) == 12
    seen = np.zeros((70, 100), dtype=int)
    for i in range(len(ds)):
        s = ds[i]
        r, c, h, w = s["metadata"]["spatial_window"]
        assert s["tile"].shape == (5, 32, 32, 8) and s["mask"].shape == (32, 32)
        assert np.array_equal(s["tile"][:, :h, :w], cube[:, r:r + h, c:c + w], equal_nan=True)
        assert s["mask"][:h, :w].all() and s["mask"].sum() == h * w                          # partial patches: zero padded, masked
        assert np.all(s["tile"][:, h:] == 0) and np.all(s["tile"][:, :, w:] == 0)
        seen[r:r + h, c:c + w] += 1
    assert (seen == 1).all()                                                               # tiles partition the raster
    with pytest.raises(IndexError):
        ds[12]
    with pytest.raises(ValueError):
        ChunkTileDataset(st, 48)


@pytest.mark.parametrize("features,dtype", [(8, np.float32), (64, np.float16)])
def test_norm_records_equal_the_per_channel_oracle(features, dtype):
    """The record table (what the device kernel consumes) evaluated in float32 numpy == the oracle's per-channel restatement."""
    import frl_oracle as O
    from frl_hip.data.normalization import NormPreset, norm_table
    from tile_cases import apply_records_np, preset_mix, raw_rows
    presets, stats = preset_mix(features)
    table = norm_table([NormPreset.from_dict(p) for p in presets], stats)
    assert table.dtype == np.uint8 and table.size == 32 * features
    raw = raw_rows((3, 50, features), dtype, seed=features)
    valid = (np.random.default_rng(1).random((3, 50)) > 0.1).astype(np.uint8)
    ref, ref_mask = O.normalize_tiles_np(raw, valid, presets, stats)
    got, got_mask = apply_records_np(raw, valid, table)
    assert np.array_equal(got_mask, ref_mask) and 0 < ref_mask.sum() < ref_mask.size
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    with pytest.raises(ValueError):
        NormPreset.from_dict({"type": "whiten"})


# ------------------------------------------------------------------------------------------------ checkpoints
def test_checkpoint_rotation_reproduces_the_reference_manager(golden_dir, tmp_path):
    """tests/golden/checkpoint_policy.json: directory listings after every epoch, produced by the reference's CheckpointManager."""
    import json
    from frl_hip.training.checkpointing import CheckpointManager, CheckpointPolicy
    nan = float("nan")

    def save_fn(state, path):
        with open(path, "w") as fh:
            json.dump({k: (None if isinstance(v, float) and v != v else v) for k, v in state.items()}, fh)

    def load_fn(path):
        with open(path) as fh:
            return {k: (nan if v is None else v) for k, v in json.load(fh).items()}

    for case in json.load(open(os.path.join(golden_dir, "checkpoint_policy.json"))):
        d = tmp_path / case["name"]
        pol = CheckpointPolicy(**case["cfg"])
        mgr = CheckpointManager(d, pol, save_fn, load_fn)
        values = [nan if v is None else v for v in case["values"]]
        for ep, v in enumerate(values):
            mgr.save(ep, {"epoch": ep + 1, pol.monitor: v}, {pol.monitor: v})
            assert sorted(os.listdir(d)) == case["listings"][ep], (case["name"], ep)
        mgr2 = CheckpointManager(d, pol, save_fn, load_fn)                      # restart: top-k list rebuilt from disk
        mgr2.restore_top_k()
        for j, v in enumerate(case["resume_values"]):
            ep = len(values) + j
            mgr2.save(ep, {"epoch": ep + 1, pol.monitor: v}, {pol.monitor: v})
            assert sorted(os.listdir(d)) == case["listings"][ep], (case["name"], "resume", j)
        with pytest.raises(KeyError):
            mgr2.save(99, {}, {"other": 1.0})


def test_checkpoint_state_round_trip(tmp_path):
    """The dict has the reference's keys, loads with weights_only=True, and `from_checkpoint`-style fields survive."""
    from frl_hip.training.checkpointing import CheckpointManager, CheckpointPolicy, build_checkpoint_state, resume_from_checkpoint
    torch.manual_seed(0)
    model = torch.nn.Linear(4, 3)
    model.type_in_channels, model.phase_in_channels = 4, 4
    opt = torch.optim.AdamW(model.parameters(), lr=3e-4)
    model(torch.randn(2, 4)).sum().backward()
    opt.step()
    state = build_checkpoint_state(model, opt, epoch=6, metrics={"val/loss": 0.25}, model_config={"version": "4"},
                                   scheduler_state={"last_epoch": 7})
    assert {"epoch", "model_version", "model_config", "type_in_channels", "phase_in_channels", "model_state_dict", "optimizer_state_dict",
            "scheduler_state_dict", "val/loss"} <= set(state) and state["epoch"] == 7 and state["model_version"] == "4"
    mgr = CheckpointManager(tmp_path, CheckpointPolicy(monitor="val/loss", save_every_n_epochs=7))
    mgr.save(6, state, {"val/loss": 0.25})
    assert sorted(os.listdir(tmp_path)) == ["encoder_best_1_epoch_007.pt", "encoder_epoch_007.pt", "encoder_last.pt"]
    m2 = torch.nn.Linear(4, 3)
    o2 = torch.optim.AdamW(m2.parameters(), lr=1.0)
    mgr2 = CheckpointManager(tmp_path, CheckpointPolicy(monitor="val/loss"))
    start, lr, sched = resume_from_checkpoint(m2, o2, tmp_path, manager=mgr2)
    assert start == 7 and abs(lr - 3e-4) < 1e-12 and sched == {"last_epoch": 7} and len(mgr2.best) == 1
    assert all(torch.equal(a, b) for a, b in zip(model.state_dict().values(), m2.state_dict().values()))
    assert resume_from_checkpoint(m2, o2, tmp_path, no_resume=True)[0] == 0


def test_denormalize_inverts_the_ingest_normalisation():
    import frl_oracle as O
    from frl_hip.data.normalization import NormPreset
    from frl_hip.training.export import denormalize
    from tile_cases import PRESET_CYCLE
    pairs = [(p, s) for p, s in PRESET_CYCLE if not (p.get("clamp") or {}).get("enabled") and p.get("in_min") != p.get("in_max") or p["type"] == "identity"]
    presets, stats = [p for p, _ in pairs], [s for _, s in pairs]
    raw = np.random.default_rng(0).standard_normal((50, len(pairs))).astype(np.float32)
    normed, mask = O.normalize_tiles_np(raw, None, presets, stats)
    assert mask.all()
    back = denormalize(normed, [NormPreset.from_dict(p) for p in presets], stats)
    assert np.abs(back - raw).max() < 1e-5


def test_legacy_trainer_script_calls_work_verbatim():
    """The statements of scripts/train_vqvae.py:153-198,221-224 that touch the dataset and the model, run as written against the
    build's classes: chunk sampler from ds.xy_by_chunk, cat_vocab_sizes / naip_bands / cont_dim from the dataset attributes, the VQVAE
    constructor call with the legacy keyword set, CodebookManager(num_codes=model.quant.codebook_size, ...), the two optimizer groups."""
    from frl_hip.data import ChunkBatchSampler, ForestDataset
    from frl_hip.models import VQVAE
    from frl_hip.training.codebook_manager import CodebookManager
    ds = ForestDataset(num_tiles=40, features=8, size=8, tiles_per_chunk=16)
    batch_sampler = ChunkBatchSampler(ds.xy_by_chunk, batch_size=8, drop_last=False, replacement_within_chunk=False, seed=42)
    assert sum(len(b) for b in batch_sampler) == 40
    cat_vocab_sizes = {}
    for name in ds.cat_names:
        entry = ds.schema_cat.get(name)
        if entry is not None:
            cat_vocab_sizes[name] = int(entry["num_ids"])
    naip_bands = int(ds.naip.shape[-1])
    cont_dim = len(ds.cont_names)
    assert (cat_vocab_sizes, naip_bands, cont_dim) == ({}, 0, 8)
    model = VQVAE(cont_dim=cont_dim, cat_vocab_sizes=cat_vocab_sizes, naip_bands=naip_bands, emb_dim=8, codebook_size=16, beta=0.25,
                  hidden=16, quantizer="st", cat_emb_dim=8, ema_decay=0.99, ema_eps=1e-5,
                  type_encoder_channels=(16, 8), type_encoder_num_groups=4, spatial_conv_gate_hidden=8, phase_tcn_channels=(8, 8, 8),
                  phase_tcn_num_groups=4, z_phase_dim=4)
    assert model.in_features == 8 and model.cont_dim == 8 and model.legacy_inputs["naip_bands"] == 0
    manager = CodebookManager(num_codes=model.quant.codebook_size, code_dim=model.quant.emb_dim)
    model.attach_codebook_manager(manager)
    codebook_params = [p for n, p in model.named_parameters() if ".quant.codebook" in n or n.endswith("quant.codebook")]
    other_params = [p for n, p in model.named_parameters() if (".quant.codebook" not in n and not n.endswith("quant.codebook"))]
    assert len(codebook_params) == 1 and codebook_params[0] is model.quant.codebook and len(other_params) > 10
    class_weights = {name: ds.class_weights_by_cat_name(name) for name in ds.cat_names}
    assert class_weights == {}
    with pytest.raises(ValueError, match="cat_vocab_sizes"):
        VQVAE(cont_dim=8, cat_vocab_sizes={"evt": 12}, naip_bands=4, emb_dim=8)
    with pytest.raises(ValueError, match="disagree"):
        VQVAE(cont_dim=8, in_features=16, emb_dim=8)
    with pytest.raises(KeyError):
        ds.class_weights_by_cat_name("evt")


def test_lambda_vq_schedule_flags():
    """`--anneal_vq_*` flags of scripts/train_vqvae.py:433-456 (build definition of the curves: training/schedules.py)."""
    from frl_hip.training.schedules import build_lambda_vq
    off = build_lambda_vq(0.7, {})
    assert [off(s) for s in (0, 10, 10**6)] == [0.7, 0.7, 0.7]
    # the script's defaults once enabled: warm-up 10000 to ceil 0.1, hold 15000, decay 5000 to final 0.08
    d = build_lambda_vq(1.0, dict(anneal_vq_enable=True, anneal_vq_schedule="warmup_hold_decay", anneal_vq_start=0, anneal_vq_floor=0.0,
                                  anneal_vq_ceil=0.1, anneal_vq_warmup=10000, anneal_vq_hold=15000, anneal_vq_decay=5000,
                                  anneal_vq_final=0.08))
    assert d(0) == 0.0 and abs(d(5000) - 0.05) < 1e-15 and d(10000) == 0.1 and d(24999) == 0.1
    assert abs(d(27500) - 0.09) < 1e-15 and d(30000) == 0.08 and d(10**7) == 0.08
    lin = build_lambda_vq(2.0, dict(anneal_vq_enable=True, anneal_vq_schedule="linear", anneal_vq_start=100, anneal_vq_duration=100,
                                    anneal_vq_floor=0.5, anneal_vq_ceil=None))
    assert lin(0) == 0.5 and lin(100) == 0.5 and lin(150) == 1.25 and lin(200) == 2.0 and lin(999) == 2.0        # ceil None -> lambda_vq
    cos = build_lambda_vq(1.0, dict(anneal_vq_enable=True, anneal_vq_schedule="cosine", anneal_vq_duration=10, anneal_vq_ceil=1.0))
    assert cos(0) == 0.0 and abs(cos(5) - 0.5) < 1e-15 and cos(10) == 1.0 and cos(2) < 0.2 * 1.0
    ex = build_lambda_vq(1.0, dict(anneal_vq_enable=True, anneal_vq_schedule="exponential", anneal_vq_duration=10, anneal_vq_ceil=1.0, anneal_vq_k=5.0))
    assert ex(0) == 0.0 and ex(10) == 1.0 and ex(5) > 0.9
    st = build_lambda_vq(1.0, dict(anneal_vq_enable=True, anneal_vq_schedule="stepwise", anneal_vq_floor=0.0,
                                   anneal_vq_milestones=["8000:0.1", "1000:0.01"]))
    assert [st(s) for s in (0, 999, 1000, 7999, 8000, 10**6)] == [0.0, 0.0, 0.01, 0.01, 0.1, 0.1]
    const = build_lambda_vq(1.0, dict(anneal_vq_enable=True, anneal_vq_schedule="constant", anneal_vq_start=5, anneal_vq_ceil=0.3, anneal_vq_floor=0.1))
    assert const(4) == 0.1 and const(5) == 0.3
    with pytest.raises(ValueError):
        build_lambda_vq(1.0, dict(anneal_vq_enable=True, anneal_vq_schedule="sawtooth"))


def test_build_trainer_from_vae_config(tmp_path):
    """configs/vae_v0.yaml gets the consumer the reference lacks: model, optimizer groups, cosine LR over T_max_epochs, beta ramp,
    lambda_vq(step) and the checkpoint directory all come from the file."""
    import yaml
    from frl_hip.config import build_trainer_from_config, load_vae_config
    raw = yaml.safe_load(open(os.path.join(ROOT, "configs", "vae_v0.yaml")))
    raw.update(codebook_size=16, emb_dim=8, hidden=16, run_root=str(tmp_path), lambda_vq=0.5, anneal_vq_enable=True, anneal_vq_schedule="linear",
               anneal_vq_duration=10, anneal_vq_ceil=None, clip_grad=0.5)
    raw["beta_schedule"] = dict(enabled=True, schedule_type="linear", start_epoch=0, end_epoch=10, start_value=0.1, end_value=1.0)
    raw["optimizer"]["scheduler"] = dict(name="cosine", T_max_epochs=2, eta_min=1e-6)
    path = tmp_path / "cfg.yaml"
    path.write_text(yaml.safe_dump(raw))
    cfg = load_vae_config(str(path))
    assert cfg.hidden == 16 and cfg.anneal_vq["anneal_vq_schedule"] == "linear" and cfg.clip_grad == 0.5
    model, trainer, ckpt, run_dir = build_trainer_from_config(
        cfg, steps_per_epoch=5, in_features=8,
        model_kwargs=dict(type_encoder_channels=(16, 8), type_encoder_num_groups=4, spatial_conv_gate_hidden=8, phase_tcn_channels=(8, 8, 8),
                          phase_tcn_num_groups=4, z_phase_dim=4))
    assert model.quant.codebook_size == 16 and model.quant.emb_dim == 8 and model.decoder_type.layers[0].out_channels == 16
    assert trainer.total_steps == 10 and trainer.lr == 1e-4 and trainer.min_lr == 1e-6 and trainer.max_norm == 0.5
    groups = trainer.opt.param_groups
    assert len(groups) == 2 and groups[0]["weight_decay"] == 0.01 and groups[1]["weight_decay"] == 0.0 and groups[0]["betas"] == (0.9, 0.95)
    assert len(groups[1]["params"]) == 1 and groups[1]["params"][0] is model.quant.codebook
    assert model.quant.beta == 0.1                                   # beta_schedule at epoch 0
    trainer.set_epoch(5)
    assert abs(model.quant.beta - 0.55) < 1e-12
    assert trainer.lambda_vq_schedule(0) == 0.0 and trainer.lambda_vq_schedule(10) == 0.5
    assert run_dir == os.path.join(str(tmp_path), cfg.experiment_name) and os.path.isdir(os.path.join(run_dir, cfg.ckpt_dir))
    assert ckpt.dir == __import__("pathlib").Path(run_dir) / cfg.ckpt_dir
