"""frl_hip: MI355X-native (gfx950) VQ-VAE training hot path behind the reference's frl model API."""
from . import _lib  # noqa: F401

__version__ = "0.1.0"
