"""dpdfnet_amd -- MI355X-native DPDFNet speech enhancement.

Drop-in for the reference package's hot path (reference package/src/dpdfnet/__init__.py:3-9): the
same five public names, executed by hand-written gfx950 HIP kernels through a C ABI
(include/dpdfnet_hip.h) instead of a per-frame onnxruntime CPU session.
"""
import os as _os
from typing import TYPE_CHECKING

# The engine drives four HIP streams per handle concurrently; HIP's default of four hardware queues per process makes them
# share queues (and serialise) as soon as anything else owns a stream: one 10 s clip 11.4 ms alone, 16.0 ms beside a second
# handle.  The variable is read once, when the HIP runtime initialises, so importing this package sets a default of 8 --
# unless the application chose a value itself, or opts out with DPDFNET_NO_HW_QUEUE_DEFAULT=1.  If HIP is already up (torch
# touched the GPU first) the default cannot take effect any more: that is said once, instead of silently differing.
def _default_hw_queues() -> None:
    if _os.environ.get("DPDFNET_NO_HW_QUEUE_DEFAULT", "") not in ("", "0"):
        return
    if "GPU_MAX_HW_QUEUES" in _os.environ:
        return
    import sys as _sys
    torch = _sys.modules.get("torch")
    try:
        hip_up = bool(torch is not None and torch.cuda.is_initialized())
    except Exception:
        hip_up = False
    if hip_up:
        import warnings
        warnings.warn("dpdfnet_amd: the HIP runtime was initialised before this import, GPU_MAX_HW_QUEUES stays at its default "
                      "(4): the engine's four streams may share hardware queues with other stream owners (up to ~40 % longer "
                      "small-batch calls).  Set GPU_MAX_HW_QUEUES=8 in the environment, or import dpdfnet_amd first.",
                      RuntimeWarning, stacklevel=3)
        return
    _os.environ["GPU_MAX_HW_QUEUES"] = "8"


_default_hw_queues()

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
