"""Tile ingest on the GPU: `frl_normalize_tiles` through the C ABI against the oracle's per-channel numpy
restatement -- bit-exact for float32 output, exactly its round-to-nearest-even for bf16 -- and the prefetcher end to end."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import frl_oracle as O  # noqa: E402
from tile_cases import preset_mix, raw_rows  # noqa: E402

DEV = "cuda:0"


def _table(presets, stats):
    from frl_hip.data.normalization import NormPreset, norm_table
    return torch.from_numpy(norm_table([NormPreset.from_dict(p) for p in presets], stats)).to(DEV)


@pytest.mark.parametrize("features", [8, 16, 64, 128, 512])
@pytest.mark.parametrize("raw_dtype", [np.float16, np.float32])
@pytest.mark.parametrize("lead", [(1,), (37,), (3, 5, 33), (2, 5, 32, 32)])
def test_normalize_tiles_bit_exact(features, raw_dtype, lead):
    from frl_hip import ops
    presets, stats = preset_mix(features)
    raw = raw_rows(lead + (features,), raw_dtype, seed=features + len(lead))
    valid = (np.random.default_rng(2).random(lead) > 0.15).astype(np.uint8)
    ref, ref_mask = O.normalize_tiles_np(raw, valid, presets, stats)
    raw_d, valid_d = torch.from_numpy(raw).to(DEV), torch.from_numpy(valid).to(DEV)
    out, mask = ops.normalize_tiles(raw_d, _table(presets, stats), valid=valid_d, out_dtype=torch.float32)
    assert np.array_equal(mask.cpu().numpy(), ref_mask)
    assert np.array_equal(out.cpu().numpy().view(np.uint32), ref.view(np.uint32))          # bit-exact
    out16, mask16 = ops.normalize_tiles(raw_d, _table(presets, stats), valid=valid_d.bool(), out_dtype=torch.bfloat16)
    want16 = torch.from_numpy(ref).to(torch.bfloat16)                                        # RNE of the float32 result
    assert torch.equal(out16.cpu().view(torch.int16), want16.view(torch.int16)) and torch.equal(mask16, mask)
    # no explicit validity: only the finiteness rule
    ref2, ref_mask2 = O.normalize_tiles_np(raw, None, presets, stats)
    out2, mask2 = ops.normalize_tiles(raw_d, _table(presets, stats), out_dtype=torch.float32)
    assert np.array_equal(mask2.cpu().numpy(), ref_mask2) and np.array_equal(out2.cpu().numpy().view(np.uint32), ref2.view(np.uint32))


def test_normalize_tiles_edges_and_errors():
    from frl_hip import ops
    from frl_hip._lib import FrlHipError
    presets, stats = preset_mix(64)
    tb = _table(presets, stats)
    empty = torch.empty(0, 64, dtype=torch.float16, device=DEV)
    out, mask = ops.normalize_tiles(empty, tb)
    assert out.shape == (0, 64) and mask.shape == (0,)
    allnan = torch.full((5, 64), float("nan"), dtype=torch.float32, device=DEV)
    out, mask = ops.normalize_tiles(allnan, tb, out_dtype=torch.float32)
    assert not mask.any() and not out.any()
    with pytest.raises(FrlHipError):                                                         # 24 features: not a supported row width
        p24, s24 = preset_mix(24)
        ops.normalize_tiles(torch.zeros(4, 24, device=DEV), _table(p24, s24))
    with pytest.raises(ValueError):
        ops.normalize_tiles(torch.zeros(4, 64, device=DEV), tb[:-32])
    with pytest.raises(ValueError):
        ops.normalize_tiles(torch.zeros(4, 64, dtype=torch.float64, device=DEV), tb)


@pytest.mark.parametrize("tile,features,raw_dtype", [(32, 64, np.float16), (8, 16, np.float32), (12, 8, np.float32)])
def test_normalize_chunk_tiles_cuts_pads_and_normalises(tile, features, raw_dtype):
    """Device-side tile cut == host-side window read + zero padding (forest_dataset_v2.py:357-369) + oracle normalisation."""
    from frl_hip import ops
    presets, stats = preset_mix(features)
    T, CY, CX = 5, 4 * tile, 3 * tile
    chunk = raw_rows((T, CY, CX, features), raw_dtype, seed=tile, inf_rows=False)
    # full tiles, partial tiles (raster edge), a 1 x 1 sliver and a repeated tile
    desc = np.array([[0, 0, tile, tile], [tile, 2 * tile, tile, tile], [3 * tile, 0, tile - 3, tile], [2 * tile, tile, tile, 5],
                     [3 * tile, 2 * tile, 1, 1], [0, 0, tile, tile]], dtype=np.int32)
    raw = np.zeros((len(desc), T, tile, tile, features), dtype=raw_dtype)
    valid = np.zeros((len(desc), T, tile, tile), dtype=np.uint8)
    for b, (y0, x0, h, w) in enumerate(desc):
        raw[b, :, :h, :w] = chunk[:, y0:y0 + h, x0:x0 + w]
        valid[b, :, :h, :w] = 1
    ref, ref_mask = O.normalize_tiles_np(raw, valid, presets, stats)
    out, mask = ops.normalize_chunk_tiles(torch.from_numpy(chunk).to(DEV), torch.from_numpy(desc).to(DEV), tile, _table(presets, stats),
                                          out_dtype=torch.float32)
    assert out.shape == raw.shape and np.array_equal(mask.cpu().numpy(), ref_mask)
    assert np.array_equal(out.cpu().numpy().view(np.uint32), ref.view(np.uint32))
    out16, _ = ops.normalize_chunk_tiles(torch.from_numpy(chunk).to(DEV), torch.from_numpy(desc).to(DEV), tile, _table(presets, stats))
    assert torch.equal(out16.cpu().view(torch.int16), torch.from_numpy(ref).to(torch.bfloat16).view(torch.int16))
    with pytest.raises(ValueError):
        ops.normalize_chunk_tiles(torch.from_numpy(chunk).to(DEV), torch.from_numpy(desc[:, :3].copy()).to(DEV), tile, _table(presets, stats))


def _store(tmp_path, shape=(5, 100, 130, 64), chunks=(64, 64), dtype="float16", seed=3):
    from frl_hip.data.tile_store import TileStore, write_tile_store
    presets, stats = preset_mix(shape[-1])
    names = [f"f{i:03d}" for i in range(shape[-1])]
    cube = raw_rows(shape, np.float32, seed=seed, inf_rows=False)
    write_tile_store(str(tmp_path / "store"), cube, chunks, names, normalization={n: p for n, p in zip(names, presets)},
                     stats={n: s for n, s in zip(names, stats) if s is not None}, dtype=dtype)
    return TileStore(str(tmp_path / "store")), presets, stats


def test_prefetcher_delivers_oracle_normalised_batches(tmp_path):
    from frl_hip.data import ChunkBatchSampler, ChunkTileDataset, TilePrefetcher
    st, presets, stats = _store(tmp_path)
    ds = ChunkTileDataset(st, 32)
    sampler = ChunkBatchSampler(ds.xy_by_chunk, 3, seed=5)
    np.random.seed(0)
    batches = list(sampler)
    assert len(batches) > 6 and any(len(b) < 3 for b in batches)                             # ragged tails included
    batches.append(list(ds.xy_by_chunk[0]))                                                  # a whole chunk: device-side tile cut
    batches.append(list(ds.xy_by_chunk[5]))                                                  # the corner chunk: partial tiles only
    pf = TilePrefetcher(ds, batches, device=DEV, out_dtype=torch.float32, depth=2, workers=3)
    assert pf._single_chunk(batches[-1]) == (1, 2) and any(pf._single_chunk(b) is None for b in batches)     # both upload paths run
    seen = []
    for item in pf:
        idx = item["indices"]
        raw = np.stack([ds[i]["tile"] for i in idx])
        valid = np.stack([np.broadcast_to(ds[i]["mask"], (5, 32, 32)) for i in idx]).astype(np.uint8)
        ref, ref_mask = O.normalize_tiles_np(raw, valid, presets, stats)
        assert item["tile"].shape == (len(idx), 5, 32, 32, 64) and item["tile"].dtype == torch.float32
        assert np.array_equal(item["mask"].cpu().numpy(), ref_mask)
        assert np.array_equal(item["tile"].cpu().numpy().view(np.uint32), ref.view(np.uint32))
        seen += idx
    assert sorted(seen) == sorted(list(range(len(ds))) + list(ds.xy_by_chunk[0]) + list(ds.xy_by_chunk[5]))
    # early exit of the consumer must not hang the producer thread
    it = iter(TilePrefetcher(ds, batches, device=DEV, depth=2, workers=2))
    first = next(it)
    assert first["tile"].dtype == torch.bfloat16
    it.close()


def test_training_step_from_the_tile_store(tmp_path):
    """Store -> sampler -> prefetcher -> VQ-VAE train steps: the mask produced by the ingest kernel drives the masked L2 losses."""
    from frl_hip.data import ChunkBatchSampler, ChunkTileDataset, TilePrefetcher
    from frl_hip.models import VQVAE
    from frl_hip.training.trainer import VQVAETrainer
    st, _, _ = _store(tmp_path, shape=(5, 64, 96, 64), chunks=(64, 32))
    ds = ChunkTileDataset(st, 32)
    torch.manual_seed(0)
    m = VQVAE(in_features=64, codebook_size=64, emb_dim=64, compute_dtype=torch.bfloat16).to(DEV)
    tr = VQVAETrainer(m, lr=1e-3, total_steps=10)
    np.random.seed(1)
    losses = []
    for item in TilePrefetcher(ds, ChunkBatchSampler(ds.xy_by_chunk, 2, seed=1), device=DEV):
        losses.append(float(tr.step(item["tile"], mask=item["mask"])["loss"]))
    assert len(losses) == 3 and all(np.isfinite(losses))
