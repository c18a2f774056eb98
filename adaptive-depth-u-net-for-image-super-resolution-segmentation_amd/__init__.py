"""adunet_amd -- MI355X (gfx950) native adaptive-depth U-Net hot path.

Host-side mirror of the reference's Keras call surface (shared/custom_layers.py,
Super_resolution/code/train_adaptive_unet.py) over hand-written HIP kernels reached through the
C ABI of include/adunet.h.  Importing the package does not need a GPU; computing anything does.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
__version__ = "0.1.0"
