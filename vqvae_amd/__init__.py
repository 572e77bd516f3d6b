"""vqvae_amd -- MI355X (gfx950) native VQ-VAE forward path.

The product is libvqvae_hip.so (hand-written HIP kernels behind the C ABI of
include/vqvae_hip.h); this package is its host-side mirror of the reference's
module interface.  See DESIGN.md and INTEGRATION.md.
"""
from ._lib import LIB_PATH, VqvaeHipError, load  # noqa: F401

__all__ = ["LIB_PATH", "VqvaeHipError", "load"]
