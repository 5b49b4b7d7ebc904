from .spatial import extract_at_locations, extract_temporal_at_locations  # noqa: F401
