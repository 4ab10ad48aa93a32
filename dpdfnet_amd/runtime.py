"""Runtime seam: build the GPU frame-function handle for a resolved model.

Counterpart of reference package/src/dpdfnet/onnx_backend.py (`RuntimeModel`,
`build_runtime_model`, `infer_win_len`, `load_initial_state_from_metadata`).  `api` and `stream`
reach the engine only through the names below, so tests can substitute them exactly where the
reference's tests substitute the ONNX session (reference tests/test_package_behaviors.py:95-107)."""
from __future__ import annotations

import threading
from dataclasses import dataclass
from pathlib import Path
from typing import Dict, Optional, Tuple, Union

import numpy as np

from .models import SYNTHETIC_PREFIX, ModelInfo


@dataclass(frozen=True)
class RuntimeModel:
    session: object            # backend.HipModel (or a test double with the same methods)
    init_state: np.ndarray     # reference flat state vector, float32 [S]
    info: ModelInfo
    device: int = 0


_cache: Dict[Tuple[str, int, int], RuntimeModel] = {}
_cache_lock = threading.Lock()


def build_runtime_model(weights_path: Union[str, Path], info: ModelInfo, device: int = 0, replica: int = 0) -> RuntimeModel:
    """Load weights (file or "synthetic:<seed>"), create the HIP model on `device`.  Cached per
    (weights, device, replica): unlike an ORT session the handle is immutable and thread-safe; `replica` > 0 asks
    for a further independent handle on the same GPU (own streams and workspace)."""
    from . import backend, weights

    key = (str(weights_path), int(device), int(replica))
    with _cache_lock:
        hit = _cache.get(key)
        if hit is not None and hit.info.name == info.name:
            return hit
    entries = backend.manifest(info.sample_rate, info.dprnn_num_blocks)
    extras: Dict[str, np.ndarray] = {}
    if isinstance(weights_path, str) and weights_path.startswith(SYNTHETIC_PREFIX):
        blob = weights.synth_blob(entries, int(weights_path[len(SYNTHETIC_PREFIX):] or 0))
    else:
        blob, extras = weights.load_weight_file(weights_path, entries)
    hip = backend.HipModel(info.sample_rate, info.dprnn_num_blocks, blob, device=device,
                           erb_norm_init=extras.get("erb_norm_init"), spec_norm_init=extras.get("spec_norm_init"))
    rt = RuntimeModel(session=hip, init_state=hip.initial_state(), info=info, device=int(device))
    with _cache_lock:
        _cache[key] = rt
    return rt


def infer_win_len(session, default_sr: int) -> int:
    """Window length of the model (reference onnx_backend.py:102-107 derives it from the ONNX input
    shape; here the engine reports it)."""
    win = int(getattr(session, "win_len", 0) or 0)
    return win if win > 1 else int(round(default_sr * 0.02))


def clear_cache() -> None:
    with _cache_lock:
        for rt in _cache.values():
            close = getattr(rt.session, "close", None)
            if close:
                close()
        _cache.clear()
