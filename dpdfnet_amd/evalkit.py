"""Waveform-level parity / quality report.

PESQ and STOI need the `pesq` / `pystoi` packages (reference pesq_stoi_sisnr_calc.py:11-12), which
are optional here; SI-SNR and the cross-correlation alignment are plain NumPy/SciPy restatements of
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
    return rep
