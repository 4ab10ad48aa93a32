"""Public offline API (reference package/src/dpdfnet/api.py:16-280).

`enhance()` keeps the reference signature and semantics; the per-frame `session.run` loop of the
reference (api.py:96-104) is replaced by ONE call into the HIP engine (`dpdf_enhance_batch`) that
runs STFT, all frames, attenuation limit and iSTFT on the GPU.  `enhance_batch()` is the
extension that makes the GPU worthwhile: many clips per call."""
from __future__ import annotations

import wave
from pathlib import Path
from typing import Any, Callable, Dict, List, Optional, Sequence, Union

import numpy as np

from .models import DEFAULT_MODEL, available_model_entries, resolve_model
from .runtime import build_runtime_model, infer_win_len

ProgressCb = Optional[Callable[[int, int], None]]


def available_models() -> List[Dict[str, Any]]:
    return available_model_entries()


def download(model: Optional[str] = None, *, force: bool = False, quiet: bool = False, verbose: bool = False):
    """The reference downloads ONNX files from Hugging Face (api.py:22-48).  This build has no
    network path: place weight files in DPDFNET_MODEL_DIR instead."""
    if quiet and verbose:
        raise ValueError("quiet=True and verbose=True are mutually exclusive.")
    raise RuntimeError(
        "download() is unavailable in the MI355X build (no network access): copy <model>.npz/.safetensors/.pth "
        "into DPDFNET_MODEL_DIR or pass onnx_path=<weight file>."
    )


def _device_from_env() -> int:
    import os
    return int(os.environ.get("DPDFNET_DEVICE", os.environ.get("LOCAL_RANK", "0")))


def enhance(
    audio: np.ndarray,
    sample_rate: int,
    *,
    model: str = DEFAULT_MODEL,
    onnx_path: Optional[Union[str, Path]] = None,
    attn_limit_db: Optional[float] = None,
    verbose: bool = False,
    progress_callback: ProgressCb = None,
) -> np.ndarray:
    """Enhance one clip; returns float32 mono of the input length (reference api.py:51-113).
    `onnx_path` is the weight-file path (or "synthetic:<seed>")."""
    from .audio import to_mono

    waveform = to_mono(np.asarray(audio, dtype=np.float32))
    resolved = resolve_model(model=model, onnx_path=onnx_path, auto_download=True, verbose=verbose)
    runtime = build_runtime_model(resolved.onnx_path, resolved.info, _device_from_env())
    return _enhance_with_runtime(waveform, sample_rate, runtime=runtime, model_sample_rate=resolved.info.sample_rate,
                                 attn_limit_db=attn_limit_db, progress_callback=progress_callback)


def _enhance_with_runtime(
    audio: np.ndarray,
    sample_rate: int,
    *,
    runtime,
    model_sample_rate: int,
    attn_limit_db: Optional[float] = None,
    progress_callback: ProgressCb = None,
) -> np.ndarray:
    """Like `enhance()` with a pre-built runtime (reference api.py:116-169)."""
    from .audio import ensure_sample_rate, fit_length, to_mono, validate_attn_limit_db

    waveform = to_mono(np.asarray(audio, dtype=np.float32))
    sr_in = int(sample_rate)
    attn = validate_attn_limit_db(attn_limit_db)
    wav_model = ensure_sample_rate(waveform, sr_in, model_sample_rate)
    win_len = infer_win_len(runtime.session, model_sample_rate)
    total_frames = 1 + (wav_model.shape[0] + win_len) // (win_len // 2)   # api.py:88 pad + centre STFT
    if progress_callback is not None:
        progress_callback(0, total_frames)
    if wav_model.shape[0] == 0:
        return waveform.copy()
    if progress_callback is None or not hasattr(runtime.session, "progress"):
        enhanced_model_sr = runtime.session.enhance_batch(wav_model[None, :], attn)[0]
        if progress_callback is not None:
            for t in range(total_frames):
                progress_callback(t + 1, total_frames)
    else:
        # The (done, total) protocol of api.py:94-104 with REAL progress: the engine call runs on a worker thread (it releases
        # the GIL) and publishes the frames whose output is complete, once per time chunk; this thread polls that figure and
        # reports every finished frame, in order, on the caller's thread.
        import threading
        import time
        box: Dict[str, Any] = {}

        def _call() -> None:
            try:
                box["out"] = runtime.session.enhance_batch(wav_model[None, :], attn)[0]
            except Exception as exc:                       # re-raised below, on the caller's thread
                box["err"] = exc

        th = threading.Thread(target=_call, name="dpdfnet-enhance")
        th.start()
        reported = 0
        try:
            import inspect
            owned = "owner" in inspect.signature(runtime.session.progress).parameters
        except (TypeError, ValueError):
            owned = False
        try:
            while th.is_alive():
                th.join(0.002)
                # only the frames of OUR worker's call: 0 until it is inside the engine (and again once it has left)
                done = min(int(runtime.session.progress(owner=th.ident) if owned else runtime.session.progress()), total_frames)
                while reported < done:
                    reported += 1
                    progress_callback(reported, total_frames)
        finally:
            th.join()           # a callback that raises must not leave the engine call running unobserved
        if "err" in box:
            raise box["err"]
        while reported < total_frames:
            reported += 1
            progress_callback(reported, total_frames)
        enhanced_model_sr = box["out"]
    enhanced = ensure_sample_rate(enhanced_model_sr, model_sample_rate, sr_in)
    return fit_length(enhanced, waveform.shape[0]).astype(np.float32, copy=False)


# ----------------------------------------------------------------------------------------------
# batched execution: length buckets, ragged engine calls, one handle + host thread per GPU
# ----------------------------------------------------------------------------------------------
RAGGED_MIN_FILL = 0.8                                  # shortest clip of a bucket >= 0.8 x its longest: <= 20 % padding frames
MAX_BATCH_SAMPLES = 64 * 1024 * 1024                   # per engine call (model-rate samples incl. padding); spectra < 1 GB


def _length_buckets(lengths: Sequence[int], max_samples: int = MAX_BATCH_SAMPLES) -> List[List[int]]:
    """Indices of the non-empty clips, longest first, cut into buckets whose clips are within RAGGED_MIN_FILL of the
    bucket's longest and whose padded size stays under `max_samples`.  A directory of arbitrary lengths becomes a handful
    of engine calls (the reference runs one file per host thread, cli.py:249-259)."""
    order = sorted((i for i, n in enumerate(lengths) if n > 0), key=lambda i: (-lengths[i], i))
    buckets: List[List[int]] = []
    cur: List[int] = []
    for i in order:
        if cur and (lengths[i] < RAGGED_MIN_FILL * lengths[cur[0]] or (len(cur) + 1) * lengths[cur[0]] > max_samples):
            buckets.append(cur)
            cur = []
        cur.append(i)
    if cur:
        buckets.append(cur)
    return buckets


def _run_bucket(session, clips: List[np.ndarray], attn: Optional[float]) -> List[np.ndarray]:
    """One engine call for a bucket.  The clips go in as they lie (row pointers, `dpdf_enhance_batch_rows`): stacking 256 ten-second
    clips into one block and slicing the result apart costs more host time than the GPU needs for the batch."""
    return session.enhance_batch_ragged(clips, attn)


def _runtimes_for(resolved, devices: Optional[Sequence[int]]) -> list:
    """One runtime (engine handle) per entry of `devices`; the same GPU listed twice gets two independent handles."""
    devs = [_device_from_env()] if not devices else [int(d) for d in devices]
    seen: Dict[int, int] = {}
    rts = []
    for d in devs:
        k = seen.get(d, 0)
        seen[d] = k + 1
        rts.append(build_runtime_model(resolved.onnx_path, resolved.info, d) if k == 0
                   else build_runtime_model(resolved.onnx_path, resolved.info, d, replica=k))
    return rts


def _enhance_model_rate_clips(runtimes: list, model_clips: List[np.ndarray], attn: Optional[float],
                              on_done: Optional[Callable[[int, np.ndarray], None]] = None) -> List[Optional[np.ndarray]]:
    """Model-rate mono clips -> enhanced clips (None for empty ones).  Buckets by length; every bucket is sharded
    contiguously over the handles (`shard_range`, the sharding bench.py uses across ranks) and each handle's share runs
    on its own host thread (the engine call releases the GIL), so N GPUs work concurrently with no exchange between
    them: utterances are independent."""
    from .multi_gpu import shard_range
    lengths = [int(c.shape[0]) for c in model_clips]
    out: List[Optional[np.ndarray]] = [None] * len(model_clips)
    nd = len(runtimes)
    work: List[List[List[int]]] = [[] for _ in range(nd)]
    for bucket in _length_buckets(lengths):
        for k in range(nd):
            lo, hi = shard_range(len(bucket), nd, k)
            if hi > lo:
                work[k].append(bucket[lo:hi])
    errors: List[Exception] = []
    import threading
    done_lock = threading.Lock()          # on_done (file writes, the caller's file_callback) runs on the workers: one at a time

    def _worker(k: int) -> None:
        try:
            for idxs in work[k]:
                res = _run_bucket(runtimes[k].session, [model_clips[i] for i in idxs], attn)
                for i, r in zip(idxs, res):
                    out[i] = r
                    if on_done is not None:
                        with done_lock:
                            on_done(i, r)
        except Exception as exc:  # surfaced on the calling thread (KeyboardInterrupt / SystemExit are not swallowed)
            errors.append(exc)

    if nd == 1:
        _worker(0)
    else:
        ts = [threading.Thread(target=_worker, args=(k,), name=f"dpdfnet-gpu{runtimes[k].device}") for k in range(nd) if work[k]]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
    if errors:
        raise errors[0]
    return out


def enhance_batch(
    audio: Union[np.ndarray, Sequence[np.ndarray]],
    sample_rate: int,
    *,
    model: str = DEFAULT_MODEL,
    onnx_path: Optional[Union[str, Path]] = None,
    attn_limit_db: Optional[float] = None,
    verbose: bool = False,
    devices: Optional[Sequence[int]] = None,
) -> List[np.ndarray]:
    """Enhance many mono clips on the GPU(s).  Clips may differ in length: they are bucketed by length and each bucket
    is ONE ragged engine call (`dpdf_enhance_batch_ragged`) in which every clip keeps its own tail semantics (SURVEY
    appendix A.4), so each result equals `enhance()` on that clip alone.  `devices=[0, 1, ...]`: one engine handle and
    host thread per listed GPU, clips sharded contiguously across them (no exchange: utterances are independent)."""
    from .audio import ensure_sample_rate, fit_length, to_mono, validate_attn_limit_db

    clips = [to_mono(np.asarray(a, dtype=np.float32)) for a in (audio if not isinstance(audio, np.ndarray) or audio.ndim != 2 else list(audio))]
    attn = validate_attn_limit_db(attn_limit_db)
    resolved = resolve_model(model=model, onnx_path=onnx_path, auto_download=True, verbose=verbose)
    runtimes = _runtimes_for(resolved, devices)
    msr, sr_in = resolved.info.sample_rate, int(sample_rate)
    dev0 = runtimes[0].device                      # resampling runs on a GPU the call was given, not on DPDFNET_DEVICE
    model_clips = [ensure_sample_rate(c, sr_in, msr, dev0) for c in clips]
    res = _enhance_model_rate_clips(runtimes, model_clips, attn)
    out: List[np.ndarray] = []
    for c, r in zip(clips, res):
        if r is None:
            out.append(c.copy())
        else:
            out.append(fit_length(ensure_sample_rate(r, msr, sr_in, dev0), c.shape[0]).astype(np.float32, copy=False))
    return out


# ----------------------------------------------------------------------------------------------
# file wrappers (reference api.py:172-280).  soundfile is optional; PCM16 WAV via the stdlib otherwise.
# ----------------------------------------------------------------------------------------------
def _read_audio(path: Path):
    try:
        import soundfile as sf  # type: ignore
        audio, sr = sf.read(str(path), always_2d=False)
        return np.asarray(audio, dtype=np.float32), int(sr)
    except ImportError:
        pass
    if path.suffix.lower() != ".wav":
        raise ValueError(f"Unsupported audio format {path.suffix!r} for file: {path} (only .wav without soundfile)")
    with wave.open(str(path), "rb") as w:
        sr, ch, sw, n = w.getframerate(), w.getnchannels(), w.getsampwidth(), w.getnframes()
        raw = w.readframes(n)
    if sw == 2:
        x = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif sw == 4:
        x = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
    elif sw == 1:
        x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    else:
        raise ValueError(f"Unsupported WAV sample width {sw} in {path}")
    if ch > 1:
        x = x.reshape(-1, ch)
    return x, int(sr)


def _probe_audio(path: Path):
    """(sample_rate, frames) from the file header only -- lets enhance_dir bucket a directory by length without
    decoding it into RAM first."""
    try:
        import soundfile as sf  # type: ignore
        info = sf.info(str(path))
        return int(info.samplerate), int(info.frames)
    except ImportError:
        pass
    if path.suffix.lower() != ".wav":
        raise ValueError(f"Unsupported audio format {path.suffix!r} for file: {path} (only .wav without soundfile)")
    with wave.open(str(path), "rb") as w:
        return int(w.getframerate()), int(w.getnframes())


def _write_pcm16(path: Path, audio: np.ndarray, sr: int) -> None:
    from .audio import pcm16_safe
    path.parent.mkdir(parents=True, exist_ok=True)
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(int(sr))
        w.writeframes(pcm16_safe(audio).astype("<i2").tobytes())


def enhance_file(
    input_path: Union[str, Path],
    output_path: Optional[Union[str, Path]] = None,
    *,
    model: str = DEFAULT_MODEL,
    onnx_path: Optional[Union[str, Path]] = None,
    attn_limit_db: Optional[float] = None,
    verbose: bool = False,
    progress_callback: ProgressCb = None,
) -> Path:
    in_path = Path(input_path).expanduser().resolve()
    if not in_path.is_file():
        raise FileNotFoundError(f"Input file not found: {in_path}")
    audio, sr = _read_audio(in_path)
    enhanced = enhance(audio=audio, sample_rate=int(sr), model=model, onnx_path=onnx_path,
                       attn_limit_db=attn_limit_db, verbose=verbose, progress_callback=progress_callback)
    out_path = in_path.with_name(f"{in_path.stem}_enhanced.wav") if output_path is None else Path(output_path).expanduser().resolve()
    _write_pcm16(out_path, enhanced, int(sr))
    return out_path


# directory batch (reference cli.py:222-311 `_run_enhance_dir`): the reference fans files out over a CPU thread pool,
# one ORT session per thread; here the files are read, brought to the model rate, bucketed by length and each bucket
# goes through the GPU as ONE ragged call (`dpdf_enhance_batch_ragged`).  Same discovery rules, output naming and
# error aggregation.  Lengths come from the file headers; files are decoded, enhanced and written bucket by bucket.
SUPPORTED_EXTENSIONS = frozenset({".wav"})            # stdlib reader; soundfile (optional) widens this at run time


def backend_resample_len(n: int, sr_in: int, sr_out: int) -> int:
    """Length of `n` samples after ensure_sample_rate (dpdf_resample_len: ceil(n * sr_out / sr_in))."""
    return n if sr_in == sr_out else -(-int(n) * int(sr_out) // int(sr_in))


def _supported_extensions() -> frozenset:
    try:
        import soundfile as sf  # type: ignore
        return frozenset("." + k.lower() for k in sf.available_formats()) | SUPPORTED_EXTENSIONS
    except ImportError:
        return SUPPORTED_EXTENSIONS


def enhance_dir(
    input_dir: Union[str, Path],
    output_dir: Union[str, Path],
    *,
    model: str = DEFAULT_MODEL,
    onnx_path: Optional[Union[str, Path]] = None,
    attn_limit_db: Optional[float] = None,
    verbose: bool = False,
    file_callback: Optional[Callable[[Path, Path], None]] = None,
    devices: Optional[Sequence[int]] = None,
) -> List[Path]:
    """Enhance every supported audio file of `input_dir` into `output_dir/<stem>_enhanced.wav`.
    Returns the written paths in the sorted order of the inputs."""
    from .audio import ensure_sample_rate, fit_length, to_mono, validate_attn_limit_db

    in_dir, out_dir = Path(input_dir).expanduser().resolve(), Path(output_dir).expanduser().resolve()
    if not in_dir.is_dir():
        raise FileNotFoundError(f"Input directory not found: {in_dir}")
    exts = _supported_extensions()
    files = sorted(p for p in in_dir.iterdir() if p.is_file() and p.suffix.lower() in exts)
    if not files:
        raise FileNotFoundError(f"No supported audio files found in {in_dir}\nSupported extensions: {', '.join(sorted(exts))}")
    attn = validate_attn_limit_db(attn_limit_db)
    resolved = resolve_model(model=model, onnx_path=onnx_path, auto_download=True, verbose=verbose)
    runtimes = _runtimes_for(resolved, devices)
    msr = resolved.info.sample_rate
    out_dir.mkdir(parents=True, exist_ok=True)

    errors: List = []
    # pass 1: headers only -> lengths at the model rate -> buckets.  pass 2: decode, enhance and write bucket by bucket, so
    # that the decoded audio held in RAM is bounded by one bucket (MAX_BATCH_SAMPLES), not by the directory.
    good: List[Path] = []
    est_len: List[int] = []
    for p in files:
        try:
            sr, n = _probe_audio(p)
            good.append(p)
            est_len.append(int(backend_resample_len(n, sr, msr)))
        except Exception as exc:  # one bad file must not stop the directory (cli.py:296-305)
            errors.append((p, exc))

    written: Dict[Path, Path] = {}
    import threading
    wlock = threading.Lock()
    buckets = _length_buckets(est_len)
    empties = [i for i, n in enumerate(est_len) if n == 0]
    for bucket in buckets + ([empties] if empties else []):
        paths = [good[i] for i in bucket]
        monos: List[np.ndarray] = []
        rates: List[int] = []
        model_clips: List[np.ndarray] = []
        kept: List[Path] = []
        for p in paths:
            try:
                audio, sr = _read_audio(p)
                mono = to_mono(audio)
                mc = ensure_sample_rate(mono, int(sr), msr, runtimes[0].device)
                kept.append(p); monos.append(mono); rates.append(int(sr)); model_clips.append(mc)
            except Exception as exc:
                errors.append((p, exc))

        def _emit(i: int, enhanced_model_sr: np.ndarray, kept=kept, monos=monos, rates=rates) -> None:
            p, mono, sr = kept[i], monos[i], rates[i]
            y = fit_length(ensure_sample_rate(enhanced_model_sr, msr, sr, runtimes[0].device), mono.shape[0])
            dst = out_dir / f"{p.stem}_enhanced.wav"
            _write_pcm16(dst, y, sr)
            with wlock:
                written[p] = dst
            if file_callback is not None:           # _emit itself runs under _enhance_model_rate_clips' lock: one callback at a time
                file_callback(p, dst)

        try:
            res = _enhance_model_rate_clips(runtimes, model_clips, attn, on_done=_emit)
            for i, r in enumerate(res):
                if r is None:                   # empty file: the reference returns the (empty) waveform itself
                    _emit(i, monos[i].copy())
        except Exception as exc:
            errors.extend((p, exc) for p in kept if p not in written)
    if errors:
        msgs = "\n".join(f"  {p}: {e}" for p, e in errors)
        raise RuntimeError(f"Errors during processing:\n{msgs}")
    return [written[p] for p in files]
