"""Model classes mirroring frl/models/__init__.py exports that live on the VQ-VAE hot path."""
from .blocks import (Conv2DEncoder, Conv2DHead, EdgeAwareSmoothingConv2D, FiLMLayer, GatedResidualBlock,  # noqa: F401
                     TCNEncoder)
from .representation import RepresentationModel  # noqa: F401
from .vqvae import VQVAE, VectorQuantizer  # noqa: F401
