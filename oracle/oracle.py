"""ctypes binding of the CPU oracle (oracle/libdpdf_oracle.so).  TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never from the
product package (dpdfnet_amd/ must not import this module; tests/test_layout.py enforces it).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from pathlib import Path
from typing import List, Optional

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "libdpdf_oracle.so"
_lib: Optional[ctypes.CDLL] = None


class DpdfCfg(ctypes.Structure):
    _fields_ = [("sample_rate", ctypes.c_int), ("nb", ctypes.c_int)]


def build(force: bool = False) -> Path:
    src = _HERE / "dpdf_oracle.c"
    hdr = _HERE.parent / "include" / "dpdf_manifest.h"
    stale = (not _LIB_PATH.exists()) or any(
        f.stat().st_mtime > _LIB_PATH.stat().st_mtime for f in (src, hdr, hdr.with_name("dpdf_norm_init.h"), _HERE / "dpdf_oracle.h")
    )
    if force or stale:
        subprocess.run(["make", "-C", str(_HERE), "-B", "libdpdf_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        try:
            L = ctypes.CDLL(str(_LIB_PATH))
        except OSError:
            build(force=True)
            L = ctypes.CDLL(str(_LIB_PATH))
        fp = ctypes.POINTER(ctypes.c_float)
        L.dpdf_oracle_create.restype = ctypes.c_void_p
        L.dpdf_oracle_create.argtypes = [ctypes.POINTER(DpdfCfg), fp, ctypes.c_size_t]
        L.dpdf_oracle_destroy.argtypes = [ctypes.c_void_p]
        L.dpdf_oracle_weight_count.restype = ctypes.c_size_t
        L.dpdf_oracle_weight_count.argtypes = [ctypes.POINTER(DpdfCfg)]
        for fn in ("dpdf_oracle_state_size", "dpdf_oracle_win_len", "dpdf_oracle_freq_bins"):
            getattr(L, fn).restype = ctypes.c_int
            getattr(L, fn).argtypes = [ctypes.c_void_p]
        L.dpdf_oracle_set_norm_init.argtypes = [ctypes.c_void_p, fp, fp]
        L.dpdf_oracle_initial_state.argtypes = [ctypes.c_void_p, fp]
        L.dpdf_oracle_frame.argtypes = [ctypes.c_void_p, fp, fp, fp, fp]
        L.dpdf_oracle_probe.restype = ctypes.c_int
        L.dpdf_oracle_probe.argtypes = [ctypes.c_void_p, ctypes.c_char_p, fp, ctypes.c_int]
        L.dpdf_oracle_num_frames.restype = ctypes.c_int
        L.dpdf_oracle_num_frames.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.dpdf_oracle_stft.argtypes = [ctypes.c_void_p, fp, ctypes.c_int, fp]
        L.dpdf_oracle_istft.argtypes = [ctypes.c_void_p, fp, ctypes.c_int, fp, ctypes.c_int]
        L.dpdf_oracle_attn_limit.argtypes = [fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_float]
        L.dpdf_oracle_resample.argtypes = [fp, ctypes.c_long, ctypes.c_int, ctypes.c_int, fp, ctypes.c_long]
        L.dpdf_oracle_resample.restype = ctypes.c_long
        L.dpdf_oracle_enhance.argtypes = [ctypes.c_void_p, fp, ctypes.c_int, ctypes.c_float, fp]
        L.dpdf_oracle_erb_widths.restype = ctypes.c_int
        L.dpdf_oracle_erb_widths.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.c_int]
        L.dpdf_oracle_window.argtypes = [ctypes.c_void_p, fp]
        L.dpdf_oracle_manifest_text.restype = ctypes.c_size_t
        L.dpdf_oracle_manifest_text.argtypes = [ctypes.POINTER(DpdfCfg), ctypes.c_char_p, ctypes.c_size_t]
        _lib = L
    return _lib


def _fp(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def manifest_text(sample_rate: int, nb: int) -> str:
    cfg = DpdfCfg(sample_rate, nb)
    n = lib().dpdf_oracle_manifest_text(ctypes.byref(cfg), None, 0)
    buf = ctypes.create_string_buffer(n + 1)
    lib().dpdf_oracle_manifest_text(ctypes.byref(cfg), buf, n + 1)
    return buf.value.decode("utf-8")


class Oracle:
    """One oracle model instance (single-threaded; create one per thread)."""

    def __init__(self, sample_rate: int, nb: int, blob: np.ndarray,
                 erb_norm_init: Optional[np.ndarray] = None, spec_norm_init: Optional[np.ndarray] = None):
        self.cfg = DpdfCfg(int(sample_rate), int(nb))
        blob = np.ascontiguousarray(blob, dtype=np.float32)
        self._h = lib().dpdf_oracle_create(ctypes.byref(self.cfg), _fp(blob), blob.size)
        if not self._h:
            raise ValueError("dpdf_oracle_create failed (bad cfg or blob size)")
        self.state_size = lib().dpdf_oracle_state_size(self._h)
        self.win_len = lib().dpdf_oracle_win_len(self._h)
        self.hop = self.win_len // 2
        self.freq_bins = lib().dpdf_oracle_freq_bins(self._h)
        if erb_norm_init is not None or spec_norm_init is not None:
            e = None if erb_norm_init is None else np.ascontiguousarray(erb_norm_init, dtype=np.float32)
            s = None if spec_norm_init is None else np.ascontiguousarray(spec_norm_init, dtype=np.float32)
            lib().dpdf_oracle_set_norm_init(self._h, None if e is None else _fp(e), None if s is None else _fp(s))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.dpdf_oracle_destroy(h)

    def initial_state(self) -> np.ndarray:
        st = np.zeros(self.state_size, dtype=np.float32)
        lib().dpdf_oracle_initial_state(self._h, _fp(st))
        return st

    def frame(self, spec: np.ndarray, state: np.ndarray):
        spec = np.ascontiguousarray(spec, dtype=np.float32).reshape(self.freq_bins, 2)
        state = np.ascontiguousarray(state, dtype=np.float32)
        out = np.empty_like(spec)
        st = np.empty_like(state)
        lib().dpdf_oracle_frame(self._h, _fp(spec), _fp(state), _fp(out), _fp(st))
        return out, st

    def run_frames(self, spec: np.ndarray, state: Optional[np.ndarray] = None):
        """spec [T,F,2] -> (spec_e [T,F,2], state_out)."""
        spec = np.ascontiguousarray(spec, dtype=np.float32)
        st = self.initial_state() if state is None else np.array(state, dtype=np.float32)
        out = np.empty_like(spec)
        for t in range(spec.shape[0]):
            o, st = self.frame(spec[t], st)
            out[t] = o
        return out, st

    def probe(self, name: str) -> np.ndarray:
        buf = np.zeros(64 * 481, dtype=np.float32)
        n = lib().dpdf_oracle_probe(self._h, name.encode(), _fp(buf), buf.size)
        return buf[:n].copy()

    def num_frames(self, n: int) -> int:
        return lib().dpdf_oracle_num_frames(self._h, int(n))

    def stft(self, wav: np.ndarray) -> np.ndarray:
        wav = np.ascontiguousarray(wav, dtype=np.float32)
        T = self.num_frames(wav.size)
        spec = np.empty((T, self.freq_bins, 2), dtype=np.float32)
        lib().dpdf_oracle_stft(self._h, _fp(wav), wav.size, _fp(spec))
        return spec

    def istft(self, spec: np.ndarray, n: int) -> np.ndarray:
        spec = np.ascontiguousarray(spec, dtype=np.float32)
        out = np.empty(n, dtype=np.float32)
        lib().dpdf_oracle_istft(self._h, _fp(spec), spec.shape[0], _fp(out), n)
        return out

    @staticmethod
    def attn_limit(noisy: np.ndarray, enh: np.ndarray, db: float) -> np.ndarray:
        noisy = np.ascontiguousarray(noisy, dtype=np.float32)
        enh = np.array(enh, dtype=np.float32)
        lib().dpdf_oracle_attn_limit(_fp(noisy), _fp(enh), enh.shape[0], enh.shape[1], float(db))
        return enh

    @staticmethod
    def resample(x: np.ndarray, sr_in: int, sr_out: int) -> np.ndarray:
        """ensure_sample_rate for mismatched rates (resample_poly restatement; see dpdf_oracle.h)."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        if int(sr_in) == int(sr_out) or x.size == 0:
            return x
        n_out = -(-x.size * int(sr_out) // int(sr_in))
        out = np.empty(n_out, dtype=np.float32)
        got = lib().dpdf_oracle_resample(_fp(x), x.size, int(sr_in), int(sr_out), _fp(out), n_out)
        assert got == n_out, (got, n_out)
        return out

    def enhance(self, wav: np.ndarray, attn_limit_db: Optional[float] = None) -> np.ndarray:
        wav = np.ascontiguousarray(wav, dtype=np.float32)
        out = np.empty_like(wav)
        db = float("nan") if attn_limit_db is None else float(attn_limit_db)
        lib().dpdf_oracle_enhance(self._h, _fp(wav), wav.size, db, _fp(out))
        return out

    def stream(self, wav: np.ndarray) -> np.ndarray:
        """The reference StreamEnhancer's arithmetic (package/src/dpdfnet/stream.py:74-200) around this oracle's frame function:
        causal float32 frame * window, float64 rfft rounded to float32 (numpy 1.26.4's np.fft.rfft, the reference's pin), frame
        function, irfft * window overlap-add, flush() with a zero-padded last window.  Pinned by tests/golden/stream_*.npz."""
        win, hop = self.win_len, self.hop
        w = self.window()
        x = np.asarray(wav, dtype=np.float32)
        n_full = (len(x) - win) // hop + 1 if len(x) >= win else 0
        left = x[n_full * hop:]
        if len(left):                                   # flush(): pad the remainder to one full window, keep len(left) samples (<= hop)
            x = np.concatenate([x[: n_full * hop], np.pad(left, (0, win - len(left)))])
        st = self.initial_state()
        ola = np.zeros(win, dtype=np.float32)
        out = []
        n_frames = (len(x) - win) // hop + 1 if len(x) >= win else 0
        for t in range(n_frames):
            fr = (x[t * hop: t * hop + win] * w).astype(np.float32)
            c = np.fft.rfft(fr.astype(np.float64), n=win)
            spec = np.stack([c.real.astype(np.float32), c.imag.astype(np.float32)], axis=-1)
            se, st = self.frame(spec, st)
            y = (np.fft.irfft(se[:, 0].astype(np.float64) + 1j * se[:, 1].astype(np.float64), n=win) * w).astype(np.float32)
            ola += y
            out.append(ola[:hop].copy())
            ola[: win - hop] = ola[hop:]
            ola[win - hop:] = 0.0
        y = np.concatenate(out) if out else np.zeros(0, np.float32)
        if len(left):
            keep = min(len(left), hop)
            y = np.concatenate([y[: n_full * hop], y[n_full * hop: n_full * hop + keep]])
        return y

    def erb_widths(self) -> List[int]:
        w = (ctypes.c_int * 64)()
        n = lib().dpdf_oracle_erb_widths(self._h, w, 64)
        return [int(w[i]) for i in range(n)]

    def window(self) -> np.ndarray:
        w = np.empty(self.win_len, dtype=np.float32)
        lib().dpdf_oracle_window(self._h, _fp(w))
        return w
