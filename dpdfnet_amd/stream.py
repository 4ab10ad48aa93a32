"""StreamEnhancer: chunk-by-chunk enhancement with persistent state
(reference package/src/dpdfnet/stream.py:13-200).

Same buffering contract as the reference -- nothing is returned until one window (20 ms) has
arrived, then exactly one hop of output per hop of input, for arbitrary chunk sizes -- but the hot
loop (stream.py:116-156: window, rfft, session.run, irfft, overlap-add) runs on the GPU through
`dpdf_streams_process`, with the RNN state, the analysis tail and the overlap-add tail resident in
HBM.  All complete hops of a `process()` call are handled by ONE device call."""
from __future__ import annotations

from pathlib import Path
from typing import Optional, Union

import numpy as np

from .audio import ensure_sample_rate, to_mono
from .models import DEFAULT_MODEL, resolve_model
from .runtime import RuntimeModel, build_runtime_model, infer_win_len


class StreamEnhancer:
    """Process audio chunk-by-chunk while preserving RNN state across calls.

    Args:
        model: Model name (default: ``"dpdfnet2"``).
        onnx_path: Optional weight-file path (or ``"synthetic:<seed>"``); overrides *model* lookup.
        verbose: Kept for signature compatibility.
    """

    def __init__(self, model: str = DEFAULT_MODEL, onnx_path: Optional[Union[str, Path]] = None,
                 verbose: bool = False) -> None:
        resolved = resolve_model(model=model, onnx_path=onnx_path, auto_download=True, verbose=verbose)
        import os
        device = int(os.environ.get("DPDFNET_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        self._runtime: RuntimeModel = build_runtime_model(resolved.onnx_path, resolved.info, device)
        self._model_sr: int = resolved.info.sample_rate
        self._win_len: int = infer_win_len(self._runtime.session, self._model_sr)
        self._hop_size: int = self._win_len // 2
        self._streams = self._runtime.session.open_streams(1)
        self._input_sr: Optional[int] = None
        self.reset()

    # ------------------------------------------------------------------
    def reset(self) -> None:
        """Reset RNN state and internal buffers (reference stream.py:62-72)."""
        self._streams.reset(-1)
        self._pending: np.ndarray = np.zeros(0, dtype=np.float32)   # samples not yet on the device
        self._primed: bool = False                                   # device holds the first hop of the window
        self._input_sr = None

    def _buffered(self) -> int:
        """len(_in_buf) of the reference: the device-held analysis tail counts once primed."""
        return (self._hop_size if self._primed else 0) + int(self._pending.shape[0])

    def process(self, chunk: np.ndarray, sample_rate: Optional[int] = None) -> np.ndarray:
        """Enhance a chunk; returns enhanced float32 mono samples, possibly empty
        (reference stream.py:74-165)."""
        chunk = to_mono(np.asarray(chunk, dtype=np.float32))
        if chunk.size == 0:
            return np.zeros(0, dtype=np.float32)
        sr_in = sample_rate if sample_rate is not None else self._model_sr
        if self._input_sr is None:
            self._input_sr = sr_in
        elif self._input_sr != sr_in:
            raise ValueError(
                f"Sample rate changed from {self._input_sr} to {sr_in} between "
                "process() calls.  Call reset() before processing a new stream."
            )
        chunk_model = ensure_sample_rate(chunk, sr_in, self._model_sr)
        self._pending = np.concatenate([self._pending, chunk_model])
        hop = self._hop_size
        if not self._primed:
            if self._pending.shape[0] < self._win_len:
                return np.zeros(0, dtype=np.float32)
            self._streams.prime(self._pending[:hop])
            self._pending = self._pending[hop:]
            self._primed = True
        k = self._pending.shape[0] // hop
        if k == 0:
            return np.zeros(0, dtype=np.float32)
        enhanced_model_sr = self._streams.process(self._pending[: k * hop]).reshape(-1)
        self._pending = self._pending[k * hop:]
        if sr_in != self._model_sr:
            return ensure_sample_rate(enhanced_model_sr, self._model_sr, sr_in)
        return enhanced_model_sr

    def flush(self) -> np.ndarray:
        """Drain the last partial window by zero-padding to a full frame (reference stream.py:167-200;
        like the reference it feeds the padding at the model rate, so it is meant for native-rate
        streams)."""
        remainder = self._buffered()
        if remainder == 0:
            return np.zeros(0, dtype=np.float32)
        sr_in = self._input_sr or self._model_sr
        pad = np.zeros(self._win_len - remainder, dtype=np.float32)
        if pad.size == 0:
            return np.zeros(0, dtype=np.float32)
        out = self.process(pad, sample_rate=self._model_sr)
        real_out = min(self._hop_size, len(out))
        trimmed = out[:real_out] if len(out) > 0 else out
        if sr_in != self._model_sr:
            trimmed = ensure_sample_rate(trimmed, self._model_sr, sr_in)
        return trimmed.astype(np.float32)
