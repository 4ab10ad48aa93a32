"""dpdfnet_amd -- MI355X-native DPDFNet speech enhancement.

Drop-in for the reference package's hot path (reference package/src/dpdfnet/__init__.py:3-9): the
same five public names, executed by hand-written gfx950 HIP kernels through a C ABI
(include/dpdfnet_hip.h) instead of a per-frame onnxruntime CPU session.
"""
import os as _os
from typing import TYPE_CHECKING

# The engine drives four HIP streams per handle concurrently; HIP's default of four hardware queues per process makes them
# share queues (and serialise) as soon as anything else owns a stream.  Must be set before the HIP runtime initialises;
# an application that sets it itself wins (csrc/dpdf_model.hip: dpdf_default_hw_queues).
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

__all__ = [
    "enhance",
    "enhance_batch",
    "enhance_file",
    "enhance_dir",
    "available_models",
    "download",
    "StreamEnhancer",
]

if TYPE_CHECKING:  # pragma: no cover
    from .api import available_models, download, enhance, enhance_batch, enhance_dir, enhance_file
    from .stream import StreamEnhancer


def __getattr__(name: str):
    if name in {"enhance", "enhance_batch", "enhance_file", "enhance_dir", "available_models", "download"}:
        from . import api

        return getattr(api, name)
    if name == "StreamEnhancer":
        from .stream import StreamEnhancer

        return StreamEnhancer
    raise AttributeError(f"module 'dpdfnet_amd' has no attribute '{name}'")
