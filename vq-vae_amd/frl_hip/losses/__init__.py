from .pairs import pairs_mutual_knn_chunked  # noqa: F401
from .contrastive import contrastive_loss  # noqa: F401
