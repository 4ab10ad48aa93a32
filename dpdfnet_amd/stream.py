"""StreamEnhancer: chunk-by-chunk enhancement with persistent state
(reference package/src/dpdfnet/stream.py:13-200).

Same buffering contract as the reference -- nothing is returned until one window (20 ms) has
arrived, then exactly one hop of output per hop of input, for arbitrary chunk sizes -- but the hot
loop (stream.py:116-156: window, rfft, session.run, irfft, overlap-add) runs on the GPU through
`dpdf_streams_process`, with the RNN state, the analysis tail and the overlap-add tail resident in
HBM.  All complete hops of a `process()` call are handled by ONE device call.

`StreamGroup` (also `StreamEnhancer.group(n)`) is the same object for n concurrent streams that advance in
lockstep -- BASELINE config 5 (64 live StreamEnhancer states): one device call per hop for ALL streams instead of
one launch sequence per stream.

`StreamPool` is for streams that do NOT advance in lockstep -- what the reference's objects are: independent, each fed
whenever its caller has a chunk (stream.py:13-72).  `pool.enhancer()` hands out objects with the StreamEnhancer interface
that share one device-resident stream set; hops that arrive within a short window (from several threads), or in one
`pool.process_many()` call, go to the GPU as ONE masked device call (`dpdf_streams_process_masked`), and
`save_state()` / `load_state()` move a stream between objects, pools or processes (resume = the explicit state vector)."""
from __future__ import annotations

import collections
import threading
from pathlib import Path
from typing import Dict, Iterable, List, Optional, Sequence, Tuple, Union

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

    @staticmethod
    def pool(n_slots: int, model: str = DEFAULT_MODEL, onnx_path: Optional[Union[str, Path]] = None,
             verbose: bool = False, window_s: float = 2e-4, **tuning) -> "StreamPool":
        """Up to n_slots INDEPENDENT streams (each a StreamEnhancer-shaped object from `pool.enhancer()`) whose hops are
        coalesced into shared device calls (`StreamPool`; tuning: its `regular_window_s`, `spin_s`)."""
        return StreamPool(n_slots, model=model, onnx_path=onnx_path, verbose=verbose, window_s=window_s, **tuning)

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


class StreamPool:
    """Independent streams behind one device-resident stream set.

    The queue, the leader election and the coalescing window live in the library (`dpdf_streams_submit*`, include/dpdfnet_hip.h):
    a member's `process()` is one C call that puts its hops into the open round's pinned input block, and returns with its row of
    the round's result; the first submitter of a round issues the ONE masked device call for everybody in it.

    Args:
        n_slots: how many streams the pool can hold at once.
        model / onnx_path / verbose: as for `StreamEnhancer`.
        window_s: how long the first submitter of a round waits for other threads' hops before it issues the device call for
            everyone queued (0: no waiting, still coalesces what is already queued).  It only waits while more than one host
            thread has been feeding the pool within the last second, and never longer than until every stream in use has queued.
        regular_window_s: the same for streams that rode in the previous round (their callers feed the pool hop after hop and
            are on their way back: firing without them costs everybody a second device call).
        spin_s: for this long a round's leader and the callers waiting for its result poll instead of sleeping (a wake-up
            through the kernel costs tens of microseconds per thread and round; 0: always sleep).  Only threads inside a
            pool call poll, without the GIL.
    """

    def __init__(self, n_slots: int, model: str = DEFAULT_MODEL, onnx_path: Optional[Union[str, Path]] = None,
                 verbose: bool = False, window_s: float = 2e-4, regular_window_s: float = 2e-3, spin_s: float = 1e-3) -> None:
        if int(n_slots) < 1:
            raise ValueError(f"n_slots must be positive, got {n_slots}")
        self._n = int(n_slots)
        resolved = resolve_model(model=model, onnx_path=onnx_path, auto_download=True, verbose=verbose)
        import os
        device = int(os.environ.get("DPDFNET_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        self._runtime: RuntimeModel = build_runtime_model(resolved.onnx_path, resolved.info, device)
        self._model_sr: int = resolved.info.sample_rate
        self._win_len: int = infer_win_len(self._runtime.session, self._model_sr)
        self._hop_size: int = self._win_len // 2
        self._streams = self._runtime.session.open_streams(self._n)
        self._streams.pool_tune(float(window_s), float(regular_window_s), float(spin_s))
        self._free: List[int] = list(range(self._n - 1, -1, -1))
        # slots of members that were dropped without close(): their finaliser only appends here (it may run at any allocation,
        # also one made under `_lock`, so it must not take a lock); `_reap` hands them back
        self._dropped: "collections.deque[int]" = collections.deque()
        self._lock = threading.Lock()
        self._slots_cache: Dict[tuple, np.ndarray] = {}      # tuple of slot numbers -> the same as an int32 array (steady-state process_many)

    @property
    def n_slots(self) -> int:
        return self._n

    @property
    def device_calls(self) -> int:
        """Masked device calls issued so far (what the coalescing saves)."""
        return self._streams.pool_stats()[0]

    def _reap(self) -> None:
        while self._dropped:
            try:
                slot = self._dropped.popleft()
            except IndexError:
                return
            self._release(slot)

    def enhancer(self) -> "PooledStreamEnhancer":
        """A new independent stream (StreamEnhancer interface) in a free slot of the pool."""
        self._reap()
        with self._lock:
            if not self._free:
                raise RuntimeError(f"all {self._n} slots of the pool are in use")
            slot = self._free.pop()
        self._streams.reset(slot)
        self._streams.slot_use(slot, True)
        return PooledStreamEnhancer(self, slot)

    def _release(self, slot: int) -> None:
        self._streams.slot_use(slot, False)      # a round's leader no longer waits for this stream
        with self._lock:
            self._free.append(slot)

    # ------------------------------------------------------------------
    def _run(self, slot: int, pcm: np.ndarray, k: int) -> np.ndarray:
        """Called by a member from its own thread: k hops for its slot; returns when the round they rode in is done."""
        if self._dropped:
            self._reap()
        return self._streams.submit_wait(slot, pcm, k)

    def process_many(self, items: Iterable[Tuple["PooledStreamEnhancer", np.ndarray]],
                     sample_rate: Optional[int] = None, wait: bool = True) -> List[np.ndarray]:
        """[(enhancer, chunk), ...] -> [enhanced, ...]: every member's complete hops in ONE submission (chunks may have
        different sizes; each result is exactly what `enhancer.process(chunk)` returns).  Other threads' submissions that arrive
        within the window share the device call(s).  wait=False: do not wait for other threads at all (a caller that knows it
        is the only one feeding the pool)."""
        if type(items) is not list:
            items = list(items)
        n = len(items)
        if self._dropped:
            self._reap()
        hop = self._hop_size
        # steady state of live streams: every member primed with nothing buffered, every chunk a float32 vector of the same whole
        # number of hops at the model rate -- the chunks ARE the round's input: one gather, one C call, results handed out as rows
        # (this is host code that runs, under the GIL, between a round's result and the caller's next submission: kept short)
        if n and (sample_rate is None or sample_rate == self._model_sr):
            first = items[0][1]
            if type(first) is np.ndarray and first.ndim == 1 and first.dtype == np.float32 and first.shape[0] and first.shape[0] % hop == 0:
                shape = first.shape
                ok = True
                key = []
                for enh, c in items:
                    if not (enh._steady and enh._pool is self and type(c) is np.ndarray and c.dtype == np.float32 and c.shape == shape):
                        ok = False
                        break
                    key.append(enh._slot)
                if ok:
                    key = tuple(key)                                   # (slot numbers, not the members: the cache must not keep them alive)
                    slots = self._slots_cache.get(key)
                    if slots is None:
                        if len(set(key)) != n:
                            raise ValueError("an enhancer appears twice in one process_many() call")
                        slots = np.array(key, dtype=np.int32)
                        if len(self._slots_cache) > 64:
                            self._slots_cache.clear()
                        self._slots_cache[key] = slots
                    out = self._streams.submit_block(slots, np.concatenate([c for _, c in items]), shape[0] // hop, not wait)
                    return list(out.reshape(n, shape[0]))
        # validate EVERY item before any member's buffer is touched: a bad later item must not leave earlier members with hops
        # taken out of their buffers and never sent to the device
        seen = set()
        for enh, chunk in items:
            if enh._pool is not self:
                raise ValueError("enhancer belongs to another pool")
            if id(enh) in seen:
                raise ValueError("an enhancer appears twice in one process_many() call")
            seen.add(id(enh))
            enh._check_rate(chunk, sample_rate)
        staged = [enh._stage(chunk, sample_rate) for enh, chunk in items]
        live = [(i, g) for i, g in enumerate(staged) if g is not None]
        res: Dict[int, np.ndarray] = {}
        if live:
            outs = self._streams.submit_many([items[i][0]._slot for i, _ in live], [g[0] for _, g in live], [g[1] for _, g in live], not wait)
            for (i, _), o in zip(live, outs):
                res[i] = o
        return [items[i][0]._finish(res[i]) if i in res else np.zeros(0, dtype=np.float32) for i in range(n)]


class PooledStreamEnhancer:
    """One independent stream of a `StreamPool`: the StreamEnhancer interface (process / flush / reset, reference
    stream.py:62-200) plus save_state / load_state."""

    def __init__(self, pool: StreamPool, slot: int) -> None:
        self._pool, self._slot = pool, slot
        self._pending = np.zeros(0, dtype=np.float32)
        self._primed = False
        self._input_sr: Optional[int] = None
        self._closed = False
        self._steady = False       # primed, nothing buffered, fed at the model rate, open: a chunk of whole hops is the device input as it is

    def _update_steady(self) -> None:
        self._steady = (self._primed and self._pending.shape[0] == 0 and self._input_sr == self._pool._model_sr and not self._closed)

    def close(self) -> None:
        """Give the slot back to the pool."""
        if not self._closed:
            self._closed = True
            self._steady = False
            self._pool._release(self._slot)

    def __enter__(self) -> "PooledStreamEnhancer":
        return self

    def __exit__(self, *_exc) -> None:
        self.close()

    def __del__(self):
        # a dropped member gives its slot back -- but a finaliser can run at any allocation, in any thread, also under the pool's
        # lock: it only leaves a note (deque.append is atomic); the pool's next call hands the slot back (StreamPool._reap)
        try:
            if not self._closed:
                self._closed = True
                self._steady = False
                self._pool._dropped.append(self._slot)
        except Exception:
            pass

    def _check_rate(self, chunk: np.ndarray, sample_rate: Optional[int]) -> None:
        """The checks of `_stage` that can fail, without consuming anything."""
        if self._closed:
            raise RuntimeError("this pool member was closed")
        if np.asarray(chunk).size == 0:
            return
        sr_in = sample_rate if sample_rate is not None else self._pool._model_sr
        if self._input_sr is not None and self._input_sr != sr_in:
            raise ValueError(
                f"Sample rate changed from {self._input_sr} to {sr_in} between "
                "process() calls.  Call reset() before processing a new stream."
            )

    def reset(self) -> None:
        self._pool._streams.reset(self._slot)
        self._pending = np.zeros(0, dtype=np.float32)
        self._primed = False
        self._input_sr = None
        self._steady = False

    # ---- the host half of process(): buffering exactly as the reference (stream.py:74-115) ----
    def _stage(self, chunk: np.ndarray, sample_rate: Optional[int]):
        """Buffer the chunk; returns (pcm of k whole hops, k) when there is something to run, else None."""
        p = self._pool
        # steady state of a live stream: primed, nothing buffered, a whole number of hops at the model rate in a float32 vector --
        # the chunk IS the device call's input (no concatenate / slice / copy: ~1 us instead of ~5 per member and hop)
        if (self._steady and type(chunk) is np.ndarray and chunk.dtype == np.float32 and chunk.ndim == 1
                and chunk.flags.c_contiguous and chunk.shape[0] and chunk.shape[0] % p._hop_size == 0
                and (sample_rate is None or sample_rate == p._model_sr)):
            return chunk, chunk.shape[0] // p._hop_size
        if self._closed:
            raise RuntimeError("this pool member was closed")
        chunk = to_mono(np.asarray(chunk, dtype=np.float32))
        if chunk.size == 0:
            return None
        sr_in = sample_rate if sample_rate is not None else p._model_sr
        if self._input_sr is None:
            self._input_sr = sr_in
        elif self._input_sr != sr_in:
            raise ValueError(
                f"Sample rate changed from {self._input_sr} to {sr_in} between "
                "process() calls.  Call reset() before processing a new stream."
            )
        self._pending = np.concatenate([self._pending, ensure_sample_rate(chunk, sr_in, p._model_sr)])
        hop = p._hop_size
        if not self._primed:
            if self._pending.shape[0] < p._win_len:
                self._steady = False
                return None
            p._streams.prime_one(self._slot, self._pending[:hop])
            self._pending = self._pending[hop:]
            self._primed = True
        k = self._pending.shape[0] // hop
        if k == 0:
            self._update_steady()
            return None
        pcm = np.ascontiguousarray(self._pending[: k * hop])
        self._pending = self._pending[k * hop:]
        self._update_steady()
        return pcm, k

    def _finish(self, enhanced_model_sr: np.ndarray) -> np.ndarray:
        p = self._pool
        if self._input_sr is not None and self._input_sr != p._model_sr:
            return ensure_sample_rate(enhanced_model_sr, p._model_sr, self._input_sr)
        return enhanced_model_sr

    def process(self, chunk: np.ndarray, sample_rate: Optional[int] = None) -> np.ndarray:
        """Enhance a chunk (reference stream.py:74-165).  Hops of other pool members queued within the pool's window
        ride in the same device call."""
        got = self._stage(chunk, sample_rate)
        if got is None:
            return np.zeros(0, dtype=np.float32)
        return self._finish(self._pool._run(self._slot, got[0], got[1]))

    def flush(self) -> np.ndarray:
        """Drain the last partial window by zero-padding to a full frame (reference stream.py:167-200)."""
        p = self._pool
        remainder = (p._hop_size if self._primed else 0) + int(self._pending.shape[0])
        if remainder == 0 or p._win_len - remainder == 0:
            return np.zeros(0, dtype=np.float32)
        sr_in = self._input_sr or p._model_sr
        out = self.process(np.zeros(p._win_len - remainder, dtype=np.float32), sample_rate=p._model_sr)
        trimmed = out[: min(p._hop_size, out.shape[0])]
        if sr_in != p._model_sr:
            trimmed = ensure_sample_rate(trimmed, p._model_sr, sr_in)
        return np.ascontiguousarray(trimmed, dtype=np.float32)

    # ---- resume = the explicit state vector (onnx_backend.py:52-78) ----
    def save_state(self) -> Dict[str, np.ndarray]:
        """Everything that defines the stream: the reference-layout model state, the analysis and overlap-add buffers
        (stream.py:62-72) and the samples not yet consumed."""
        st = self._pool._streams
        in_tail, ola_tail = st.get_tails(self._slot)
        return {"state": st.get_state(self._slot), "in_tail": in_tail, "ola_tail": ola_tail, "pending": self._pending.copy(),
                "primed": np.asarray(self._primed), "input_sr": np.asarray(-1 if self._input_sr is None else self._input_sr)}

    def load_state(self, saved: Dict[str, np.ndarray]) -> None:
        st = self._pool._streams
        st.reset(self._slot)
        primed = bool(np.asarray(saved["primed"]))
        st.set_state(self._slot, saved["state"], saved["in_tail"] if primed else None, saved["ola_tail"])
        self._pending = np.asarray(saved["pending"], dtype=np.float32).copy()
        self._primed = primed
        sr = int(np.asarray(saved["input_sr"]))
        self._input_sr = None if sr < 0 else sr
        self._update_steady()
