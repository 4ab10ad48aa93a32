"""Host-side audio helpers of the enhancement path (reference package/src/dpdfnet/audio.py).

The STFT / iSTFT / attenuation-limit of the OFFLINE path run on the GPU inside
`dpdf_enhance_batch`; what stays on the host is the cheap glue with reference-pinned semantics."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np

ATTN_LIMIT_NOISY_FRAME_OFFSET = 4   # reference audio.py:8 (2 frames feature look-ahead + 2 frames DF look-ahead); the blend
                                    # itself (audio.py:50-76) runs in the GPU deep-filter kernel (csrc/misc_kernels.h df_apply_kernel)


def to_mono(audio: np.ndarray) -> np.ndarray:
    """[samples] or [samples, channels] -> float32 mono (reference audio.py:11-17)."""
    x = np.asarray(audio, dtype=np.float32)
    if x.ndim == 1:
        return x
    if x.ndim != 2:
        raise ValueError(f"Expected mono/stereo audio, got shape {x.shape}")
    return np.mean(x, axis=1, dtype=np.float32)


def ensure_sample_rate(audio: np.ndarray, sample_rate: int, target_sample_rate: int, device: Optional[int] = None) -> np.ndarray:
    """Identity when the rates match (the on-path case, reference audio.py:20-22).  Otherwise the
    device polyphase resampler `dpdf_resample` (Kaiser-windowed sinc, the scheme of
    scipy.signal.resample_poly) -- the reference delegates to librosa/soxr_hq, whose source is not
    in the repository: parity for mismatched rates is UNPINNED.  No host fallback:
    without the HIP extension / a GPU this raises like every other engine call."""
    x = np.asarray(audio, dtype=np.float32)
    if int(sample_rate) == int(target_sample_rate):
        return x
    if x.size == 0:
        return x
    from . import backend
    if device is None:
        import os
        device = int(os.environ.get("DPDFNET_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    return backend.resample(x, int(sample_rate), int(target_sample_rate), device)


def fit_length(audio: np.ndarray, target_len: int) -> np.ndarray:
    """Truncate or zero-pad to `target_len` (reference audio.py:30-38)."""
    x = np.asarray(audio, dtype=np.float32).reshape(-1)
    if x.shape[0] == target_len:
        return x
    if x.shape[0] > target_len:
        return x[:target_len]
    out = np.zeros(target_len, dtype=np.float32)
    out[: x.shape[0]] = x
    return out


def validate_attn_limit_db(attn_limit_db: Optional[float]) -> Optional[float]:
    """None / +inf = off; negative or NaN is an error (reference audio.py:41-47)."""
    if attn_limit_db is None:
        return None
    value = float(attn_limit_db)
    if np.isnan(value) or value < 0.0:
        raise ValueError("attn_limit_db must be non-negative, infinity, or None.")
    return value


def apply_attn_limit(spec_noisy: np.ndarray, spec_enh: np.ndarray, attn_limit_db: Optional[float]) -> np.ndarray:
    """Attenuation limit on [B, T, F, 2] spectra, on the host (reference audio.py:50-76): out = a * noisy[t - 4] + (1 - a) * enh[t]
    with a = 10^(-dB / 20); frames before the 4-frame offset blend against zero.  The engine's own offline path applies the
    same blend inside the deep-filter kernel; this function is for callers that hold spectra themselves -- the
    reference's enhance() loop run over `ort_shim`, spectral post-processing."""
    value = validate_attn_limit_db(attn_limit_db)
    enh = np.asarray(spec_enh, dtype=np.float32)
    if value is None:
        return enh
    noisy = np.asarray(spec_noisy, dtype=np.float32)
    if noisy.shape != enh.shape:
        raise ValueError(f"spec_noisy and spec_enh must have matching shapes, got {noisy.shape} and {enh.shape}.")
    k = ATTN_LIMIT_NOISY_FRAME_OFFSET
    delayed = np.zeros_like(noisy)
    if noisy.shape[1] > k:
        delayed[:, k:] = noisy[:, :-k]
    a = float(10.0 ** (-value / 20.0))
    return np.ascontiguousarray(a * delayed + (1.0 - a) * enh, dtype=np.float32)


def pcm16_safe(audio: np.ndarray) -> np.ndarray:
    """Clip to [-1,1] and scale to int16 (reference audio.py:79-81)."""
    x = np.clip(np.asarray(audio, dtype=np.float32), -1.0, 1.0)
    return (x * 32767.0).astype(np.int16)


def vorbis_window(window_len: int) -> np.ndarray:
    """sin(pi/2 sin^2(pi (n+1/2)/N)) (reference audio.py:84-88); power-complementary at 50 % overlap."""
    n = np.arange(window_len)
    s = np.sin(0.5 * np.pi * (n + 0.5) / (window_len / 2))
    return np.sin(0.5 * np.pi * s * s).astype(np.float32)


@dataclass(frozen=True)
class StftConfig:
    win_len: int
    hop_size: int
    window: np.ndarray


def make_stft_config(win_len: int) -> StftConfig:
    """hop = win/2, vorbis window (reference audio.py:98-101)."""
    return StftConfig(win_len=int(win_len), hop_size=int(win_len) // 2, window=vorbis_window(int(win_len)))


def preprocess_waveform(waveform: np.ndarray, cfg: StftConfig) -> np.ndarray:
    """Host form of the analysis STFT for callers of the reference's helper (reference audio.py:104-117: librosa.stft,
    centre / reflect padding, unnormalised): waveform [n] -> spectrum [1, T, F, 2] with T = n // hop + 1.  NOT on the product
    path -- `enhance()` runs the same transform on the device inside `dpdf_enhance_batch` (csrc/gemm_rows.h `StftA`); this is
    the numpy statement of it for code that works on spectra (the ORT-shaped session of `ort_shim.py`, tests)."""
    x = np.asarray(waveform, dtype=np.float32).reshape(-1)
    win, hop = int(cfg.win_len), int(cfg.hop_size)
    if x.shape[0] <= win // 2:
        raise ValueError(f"waveform of {x.shape[0]} samples is too short for reflect padding of {win // 2}")
    xp = np.pad(x, (win // 2, win // 2), mode="reflect")
    n_frames = 1 + (xp.shape[0] - win) // hop
    idx = hop * np.arange(n_frames)[:, None] + np.arange(win)[None, :]
    spec = np.fft.rfft(xp[idx] * np.asarray(cfg.window, dtype=np.float32)[None, :], axis=-1).astype(np.complex64)
    return np.stack([spec.real, spec.imag], axis=-1).astype(np.float32)[None, ...]


def postprocess_spec(spec_e: np.ndarray, cfg: StftConfig) -> np.ndarray:
    """Host form of the synthesis (reference audio.py:120-136: librosa.istft centre mode, then drop the first 2 win samples
    -- the model's two-frame lookahead + the analysis centre -- and pad as many zeros behind).  spec_e [1, T, F, 2] ->
    waveform [hop (T - 1)].  The vorbis window is power-complementary at 50 % overlap, so librosa's window-sum-square
    normalisation is 1 over the kept range and is applied for exactness only where it differs from 1."""
    s = np.asarray(spec_e[0], dtype=np.float32)
    win, hop = int(cfg.win_len), int(cfg.hop_size)
    w = np.asarray(cfg.window, dtype=np.float32)
    frames = np.fft.irfft(s[..., 0] + 1j * s[..., 1], n=win, axis=-1).astype(np.float32) * w[None, :]
    T = frames.shape[0]
    y = np.zeros(hop * (T - 1) + win, dtype=np.float32)
    wss = np.zeros_like(y)
    for t in range(T):
        y[t * hop:t * hop + win] += frames[t]
        wss[t * hop:t * hop + win] += w * w
    ok = wss > np.finfo(np.float32).tiny
    y[ok] /= wss[ok]
    y = y[win // 2:win // 2 + hop * (T - 1)]
    return np.concatenate([y[win * 2:], np.zeros(win * 2, dtype=np.float32)], axis=0)
