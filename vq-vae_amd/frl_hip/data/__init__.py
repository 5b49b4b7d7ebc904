from .forest_dataset import ForestDataset, SyntheticTileStream, collate_fn  # noqa: F401
