from .forest_dataset import ForestDataset, SyntheticTileStream, collate_fn  # noqa: F401
from .normalization import NormPreset, norm_table  # noqa: F401
from .samplers import ChunkBatchSampler, shard_batches  # noqa: F401
from .tile_loader import ChunkTileDataset, TilePrefetcher  # noqa: F401
from .tile_store import TileStore, write_tile_store  # noqa: F401
