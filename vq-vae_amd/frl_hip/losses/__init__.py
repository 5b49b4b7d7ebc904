from .pairs import pairs_mutual_knn_chunked  # noqa: F401
