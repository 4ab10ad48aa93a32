"""StreamEnhancer: chunk-by-chunk enhancement with persistent state
(reference package/src/dpdfnet/stream.py:13-200).

Same buffering contract as the reference -- nothing is returned until one window (20 ms) has
arrived, then exactly one hop of output per hop of input, for arbitrary chunk sizes -- but the hot
loop (stream.py:116-156: window, rfft, session.run, irfft, overlap-add) runs on the GPU through
`dpdf_streams_process`, with the RNN state, the analysis tail and the overlap-add tail resident in
HBM.  All complete hops of a `process()` call are handled by ONE device call.

`StreamGroup` (also `StreamEnhancer.group(n)`) is the same object for n concurrent streams that advance in
lockstep -- BASELINE config 5 (64 live StreamEnhancer states): one device call per hop for ALL streams instead of
one launch sequence per stream."""
from __future__ import annotations

from pathlib import Path
from typing import Optional, Union

import numpy as np

from .audio import ensure_sample_rate, to_mono
from .models import DEFAULT_MODEL, resolve_model
from .runtime import RuntimeModel, build_runtime_model, infer_win_len


class StreamGroup:
    """n independent streams (own RNN state, analysis tail and overlap-add tail each, all resident on the GPU) fed
    in lockstep: every `process()` takes the same number of new samples for every stream, as an [n, m] array, and
    returns [n, k * hop].  Per stream the result is exactly what a `StreamEnhancer` of its own returns.

    Args:
        n_streams: number of concurrent streams.
        model / onnx_path / verbose: as for `StreamEnhancer`.
    """

    def __init__(self, n_streams: int, model: str = DEFAULT_MODEL, onnx_path: Optional[Union[str, Path]] = None,
                 verbose: bool = False) -> None:
        if int(n_streams) < 1:
            raise ValueError(f"n_streams must be positive, got {n_streams}")
        self._n = int(n_streams)
        resolved = resolve_model(model=model, onnx_path=onnx_path, auto_download=True, verbose=verbose)
        import os
        device = int(os.environ.get("DPDFNET_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        self._runtime: RuntimeModel = build_runtime_model(resolved.onnx_path, resolved.info, device)
        self._model_sr: int = resolved.info.sample_rate
        self._win_len: int = infer_win_len(self._runtime.session, self._model_sr)
        self._hop_size: int = self._win_len // 2
        self._streams = self._runtime.session.open_streams(self._n)
        self._input_sr: Optional[int] = None
        self.reset()

    @property
    def n_streams(self) -> int:
        return self._n

    # ------------------------------------------------------------------
    def reset(self) -> None:
        """Reset RNN state and internal buffers of every stream (reference stream.py:62-72)."""
        self._streams.reset(-1)
        self._pending: np.ndarray = np.zeros((self._n, 0), dtype=np.float32)   # samples not yet on the device
        self._primed: bool = False                                              # device holds the first hop of the window
        self._input_sr = None

    def _buffered(self) -> int:
        """len(_in_buf) of the reference: the device-held analysis tail counts once primed."""
        return (self._hop_size if self._primed else 0) + int(self._pending.shape[1])

    def _empty(self) -> np.ndarray:
        return np.zeros((self._n, 0), dtype=np.float32)

    def process(self, chunks: np.ndarray, sample_rate: Optional[int] = None) -> np.ndarray:
        """chunks [n_streams, m] float32 mono (m may be any size, also 0) -> enhanced [n_streams, k * hop]
        (reference stream.py:74-165, for every stream at once)."""
        chunks = np.asarray(chunks, dtype=np.float32)
        if chunks.ndim != 2 or chunks.shape[0] != self._n:
            raise ValueError(f"Expected chunks of shape [{self._n}, samples], got {chunks.shape}")
        if chunks.shape[1] == 0:
            return self._empty()
        sr_in = sample_rate if sample_rate is not None else self._model_sr
        if self._input_sr is None:
            self._input_sr = sr_in
        elif self._input_sr != sr_in:
            raise ValueError(
                f"Sample rate changed from {self._input_sr} to {sr_in} between "
                "process() calls.  Call reset() before processing a new stream."
            )
        chunk_model = ensure_sample_rate(chunks, sr_in, self._model_sr)
        self._pending = np.concatenate([self._pending, chunk_model], axis=1)
        hop = self._hop_size
        if not self._primed:
            if self._pending.shape[1] < self._win_len:
                return self._empty()
            self._streams.prime(np.ascontiguousarray(self._pending[:, :hop]))
            self._pending = self._pending[:, hop:]
            self._primed = True
        k = self._pending.shape[1] // hop
        if k == 0:
            return self._empty()
        enhanced_model_sr = self._streams.process(np.ascontiguousarray(self._pending[:, : k * hop])).reshape(self._n, -1)
        self._pending = self._pending[:, k * hop:]
        if sr_in != self._model_sr:
            return ensure_sample_rate(enhanced_model_sr, self._model_sr, sr_in)
        return enhanced_model_sr

    def flush(self) -> np.ndarray:
        """Drain the last partial window by zero-padding to a full frame (reference stream.py:167-200;
        like the reference it feeds the padding at the model rate, so it is meant for native-rate
        streams)."""
        remainder = self._buffered()
        if remainder == 0:
            return self._empty()
        sr_in = self._input_sr or self._model_sr
        pad = np.zeros((self._n, self._win_len - remainder), dtype=np.float32)
        if pad.shape[1] == 0:
            return self._empty()
        out = self.process(pad, sample_rate=self._model_sr)         # as the reference: raises on a resampled stream
        trimmed = out[:, : min(self._hop_size, out.shape[1])]
        if sr_in != self._model_sr:
            trimmed = ensure_sample_rate(trimmed, self._model_sr, sr_in)
        return np.ascontiguousarray(trimmed, dtype=np.float32)


class StreamEnhancer:
    """Process audio chunk-by-chunk while preserving RNN state across calls.

    Args:
        model: Model name (default: ``"dpdfnet2"``).
        onnx_path: Optional weight-file path (or ``"synthetic:<seed>"``); overrides *model* lookup.
        verbose: Kept for signature compatibility.
    """

    def __init__(self, model: str = DEFAULT_MODEL, onnx_path: Optional[Union[str, Path]] = None,
                 verbose: bool = False) -> None:
        self._g = StreamGroup(1, model=model, onnx_path=onnx_path, verbose=verbose)

    @staticmethod
    def group(n_streams: int, model: str = DEFAULT_MODEL, onnx_path: Optional[Union[str, Path]] = None,
              verbose: bool = False) -> StreamGroup:
        """n concurrent streams behind one object: one device call per hop for all of them (`StreamGroup`)."""
        return StreamGroup(n_streams, model=model, onnx_path=onnx_path, verbose=verbose)

    def reset(self) -> None:
        """Reset RNN state and internal buffers (reference stream.py:62-72)."""
        self._g.reset()

    def process(self, chunk: np.ndarray, sample_rate: Optional[int] = None) -> np.ndarray:
        """Enhance a chunk; returns enhanced float32 mono samples, possibly empty
        (reference stream.py:74-165)."""
        chunk = to_mono(np.asarray(chunk, dtype=np.float32))
        if chunk.size == 0:
            return np.zeros(0, dtype=np.float32)
        return self._g.process(chunk[None, :], sample_rate).reshape(-1)

    def flush(self) -> np.ndarray:
        """Drain the last partial window (reference stream.py:167-200)."""
        return self._g.flush().reshape(-1).astype(np.float32)
