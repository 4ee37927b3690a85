"""GL-Fusion hot path on MI355X (gfx950): PyTorch-ROCm host code over hand-written HIP kernels.

The directory is named ``gl-fusion_amd`` (not an importable identifier); import it as
``glfusion_amd`` (a shim package at the repo root points here), or put this directory on
``sys.path`` to get the reference's own ``models`` package name.
"""
from . import _lib  # noqa: F401

__all__ = ["ops", "fusion", "models", "ddp", "engine"]
__version__ = "0.1.0"
