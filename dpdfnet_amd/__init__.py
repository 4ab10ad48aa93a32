"""dpdfnet_amd -- MI355X-native DPDFNet speech enhancement.

Drop-in for the reference package's hot path (reference package/src/dpdfnet/__init__.py:3-9): the
same five public names, executed by hand-written gfx950 HIP kernels through a C ABI
(include/dpdfnet_hip.h) instead of a per-frame onnxruntime CPU session.
"""
from typing import TYPE_CHECKING

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
