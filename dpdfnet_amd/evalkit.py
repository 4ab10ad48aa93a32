"""Waveform-level parity / quality report.

PESQ and STOI need the `pesq` / `pystoi` packages (reference pesq_stoi_sisnr_calc.py:11-12), which
are optional here (`stoi_np` restates STOI from its paper for boxes without pystoi; PESQ has no such stand-in); SI-SNR and the cross-correlation alignment are plain NumPy/SciPy restatements of
the reference's definitions (pesq_stoi_sisnr_calc.py:16-27, 101-146) and are what the parity report
uses: `waveform_report(ours, reference_output)`."""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np


def si_snr(ref: np.ndarray, est: np.ndarray, eps: float = 1e-8) -> float:
    """Scale-invariant SNR in dB of `est` against `ref`, DC removed (not symmetric)."""
    r = np.asarray(ref, dtype=np.float64) - np.mean(ref)
    e = np.asarray(est, dtype=np.float64) - np.mean(est)
    alpha = float(np.dot(e, r)) / (float(np.sum(r * r)) + eps)
    target = alpha * r
    noise = e - target
    return float(10.0 * np.log10((np.sum(target * target) + eps) / (np.sum(noise * noise) + eps)))


def align_by_xcorr_trim(a: np.ndarray, b: np.ndarray) -> Tuple[np.ndarray, np.ndarray, int]:
    """Align two 1-D signals on the peak of their full cross-correlation and trim to the overlap.
    Returns (a_aligned, b_aligned, lag) with lag > 0 meaning `a` lags `b`."""
    from scipy.signal import correlate, correlation_lags

    a = np.asarray(a, dtype=np.float32).reshape(-1)
    b = np.asarray(b, dtype=np.float32).reshape(-1)
    long_is_a = len(a) >= len(b)
    lng, sht = (a, b) if long_is_a else (b, a)
    corr = correlate(lng, sht, mode="full", method="fft")
    lag = int(correlation_lags(len(lng), len(sht), mode="full")[int(np.argmax(corr))])
    l0, s0 = (lag, 0) if lag >= 0 else (0, -lag)
    n = min(len(lng) - l0, len(sht) - s0)
    if n <= 0:
        n = min(len(a), len(b))
        return a[:n].copy(), b[:n].copy(), 0
    la, sa = lng[l0:l0 + n], sht[s0:s0 + n]
    if long_is_a:
        return la.copy(), sa.copy(), lag
    return sa.copy(), la.copy(), -lag


def stoi_np(clean: np.ndarray, degraded: np.ndarray, sample_rate: int) -> float:
    """Short-time objective intelligibility (Taal, Hendriks, Heusdens, Jensen: "An algorithm for intelligibility prediction of
    time-frequency weighted noisy speech", IEEE TASL 2011) in plain NumPy / SciPy, for boxes without `pystoi` (the package the
    reference's pesq_stoi_sisnr_calc.py:149-153 calls): 10 kHz, 256-sample Hann frames at 50 % overlap, frames more than 40 dB
    below the loudest clean frame dropped, 15 one-third-octave bands from 150 Hz, 30-frame segments, clipping at -15 dB SDR, mean
    correlation.  Restated from the paper; NOT pinned against pystoi (absent here) -- it serves to show that two nearly identical
    signals score identically, which needs the same function on both, not a particular implementation."""
    from math import gcd
    from scipy.signal import resample_poly

    fs, n_frame, nfft, n_band, f_min, n_seg, beta, dyn = 10000, 256, 512, 15, 150.0, 30, -15.0, 40.0
    x = np.asarray(clean, dtype=np.float64).reshape(-1)
    y = np.asarray(degraded, dtype=np.float64).reshape(-1)
    n = min(len(x), len(y)); x, y = x[:n], y[:n]
    if sample_rate != fs:
        g = gcd(fs, int(sample_rate))
        x = resample_poly(x, fs // g, int(sample_rate) // g); y = resample_poly(y, fs // g, int(sample_rate) // g)
    w = np.hanning(n_frame + 2)[1:-1]
    hop = n_frame // 2
    nfr = (len(x) - n_frame) // hop + 1
    if nfr < n_seg:
        raise ValueError("signal too short for STOI (needs >= 30 frames of 25.6 ms after silence removal)")
    idx = np.arange(n_frame)[None, :] + hop * np.arange(nfr)[:, None]
    xf, yf = x[idx] * w, y[idx] * w
    en = 20.0 * np.log10(np.linalg.norm(xf, axis=1) + 1e-12)
    keep = en > en.max() - dyn
    xf, yf = xf[keep], yf[keep]
    if len(xf) < n_seg + 1:
        raise ValueError("signal too short for STOI after silence removal")
    # the kept (windowed) frames overlap-added back to two waveforms, which are then framed and windowed again
    def ola(fr):
        out = np.zeros((len(fr) - 1) * hop + n_frame)
        for i, v in enumerate(fr):
            out[i * hop: i * hop + n_frame] += v
        return out
    x, y = ola(xf), ola(yf)
    nfr = (len(x) - n_frame) // hop + 1
    idx = np.arange(n_frame)[None, :] + hop * np.arange(nfr)[:, None]
    xf, yf = x[idx] * w, y[idx] * w
    X = np.abs(np.fft.rfft(xf, nfft, axis=1)) ** 2
    Y = np.abs(np.fft.rfft(yf, nfft, axis=1)) ** 2
    f = np.linspace(0.0, fs, nfft + 1)[: nfft // 2 + 1]
    k = np.arange(n_band)
    lo, hi = f_min * 2.0 ** ((2 * k - 1) / 6.0), f_min * 2.0 ** ((2 * k + 1) / 6.0)
    obm = np.zeros((n_band, len(f)))
    for i in range(n_band):
        obm[i, int(np.argmin((f - lo[i]) ** 2)): int(np.argmin((f - hi[i]) ** 2))] = 1.0
    xt, yt = np.sqrt(X @ obm.T), np.sqrt(Y @ obm.T)          # [frames][bands]
    clip = 1.0 + 10.0 ** (-beta / 20.0)
    tot, cnt = 0.0, 0
    for m in range(n_seg, len(xt) + 1):
        xs, ys = xt[m - n_seg:m].T, yt[m - n_seg:m].T       # [bands][30]
        ys = ys * (np.linalg.norm(xs, axis=1, keepdims=True) / (np.linalg.norm(ys, axis=1, keepdims=True) + 1e-12))
        ys = np.minimum(ys, xs * clip)
        xs = xs - xs.mean(axis=1, keepdims=True); ys = ys - ys.mean(axis=1, keepdims=True)
        xs = xs / (np.linalg.norm(xs, axis=1, keepdims=True) + 1e-12); ys = ys / (np.linalg.norm(ys, axis=1, keepdims=True) + 1e-12)
        tot += float(np.sum(xs * ys)); cnt += n_band
    return tot / cnt


def waveform_report(ours: np.ndarray, reference: np.ndarray, sample_rate: int = 16000) -> Dict[str, Optional[float]]:
    """RMS error, SI-SNR of ours vs the reference output, and (when the libraries exist) the PESQ / STOI
    of `ours` measured against `reference` -- the quantities BASELINE.json's quality gate names."""
    o = np.asarray(ours, dtype=np.float64).reshape(-1)
    r = np.asarray(reference, dtype=np.float64).reshape(-1)
    n = min(len(o), len(r))
    o, r = o[:n], r[:n]
    rep: Dict[str, Optional[float]] = {
        "rms_error": float(np.sqrt(np.mean((o - r) ** 2))) if n else 0.0,
        "rms_reference": float(np.sqrt(np.mean(r ** 2))) if n else 0.0,
        "si_snr_db": si_snr(r, o) if n else None,
        "pesq_wb": None,
        "stoi": None,
    }
    try:  # optional, absent in the build image
        from pesq import pesq  # type: ignore
        from pystoi.stoi import stoi  # type: ignore
        rep["pesq_wb"] = float(pesq(sample_rate, r.astype(np.float32), o.astype(np.float32), "wb"))
        rep["stoi"] = float(stoi(r, o, sample_rate, extended=False))
    except Exception:
        pass
    if rep["stoi"] is None and n:
        try:
            rep["stoi"] = stoi_np(r, o, sample_rate); rep["stoi_impl"] = "dpdfnet_amd.evalkit.stoi_np (pystoi absent)"
        except ValueError:
            pass
    return rep
