"""Shared test helpers (test infrastructure; may import oracle/)."""
from __future__ import annotations

import json
from pathlib import Path
from typing import Optional

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
GOLDEN = ROOT / "tests" / "golden"
MODEL_TAGS = ["16k_nb0", "16k_nb1", "16k_nb2", "16k_nb4", "16k_nb8", "48k_nb1", "48k_nb2", "48k_nb8",
              # weight-robustness goldens: saturating GRU gates / BatchNorm var ~ eps + LayerNorm gains x 5 (weights.stress_blob)
              "16k_nb2_hot", "16k_nb2_stiff", "48k_nb1_hot", "48k_nb1_stiff"]


def rms(x) -> float:
    return float(np.sqrt(np.mean(np.square(np.asarray(x, dtype=np.float64)))))


def synth_clip(n: int, sr: int, seed: int) -> np.ndarray:
    rng = np.random.default_rng(seed)
    t = np.arange(n, dtype=np.float64) / sr
    f0 = rng.uniform(100.0, 1000.0)
    x = 0.05 * rng.standard_normal(n) + 0.1 * np.sin(2 * np.pi * f0 * t) * (1.0 + np.sin(2 * np.pi * 3.0 * t))
    return np.clip(x, -1.0, 1.0).astype(np.float32)


def load_golden(tag: str):
    g = np.load(GOLDEN / f"model_{tag}.npz")
    meta = json.loads(bytes(g["meta_json"]).decode())
    return g, meta


def norm_inits(sr: int):
    """The reference's initial norm states as captured in constants.npz (to ASSERT against, not to inject)."""
    C = np.load(GOLDEN / "constants.npz")
    k = "16k" if sr == 16000 else "48k"
    return C[f"erb_norm_init_{k}"], C[f"spec_norm_init_{k}"]


def golden_blob(meta) -> np.ndarray:
    from oracle import oracle as orc
    from dpdfnet_amd.weights import parse_manifest_text, stress_blob, synth_blob
    entries = parse_manifest_text(orc.manifest_text(meta["sample_rate"], meta["nb"]))
    if meta.get("stress"):
        return stress_blob(entries, meta["seed"], meta["stress"])
    return synth_blob(entries, meta["seed"])


def make_oracle(meta, blob):
    """The oracle on its OWN default initial state (no tables injected): include/dpdf_norm_init.h is what is tested."""
    from oracle import oracle as orc
    return orc.Oracle(meta["sample_rate"], meta["nb"], blob)


# ----- numpy passthrough doubles for the engine (host-logic tests without a GPU) ---------------
def _window(win: int) -> np.ndarray:
    n = np.arange(win)
    s = np.sin(0.5 * np.pi * (n + 0.5) / (win / 2))
    return np.sin(0.5 * np.pi * s * s).astype(np.float32)


class PassthroughStreams:
    """Causal STFT -> identity frame function -> overlap-add (what the reference's passthrough
    session makes StreamEnhancer compute, tests/test_package_behaviors.py:421-446)."""

    def __init__(self, win: int, n: int = 1, zero: bool = False):
        self.win, self.hop, self.n, self.zero = win, win // 2, n, zero
        self.w = _window(win)
        self.reset(-1)

    def reset(self, _stream: int = -1):
        if _stream is not None and _stream >= 0 and hasattr(self, "tail"):
            self.reset_one(_stream)
            return
        self.tail = np.zeros((self.n, self.hop), dtype=np.float32)
        self.ola = np.zeros((self.n, self.hop), dtype=np.float32)

    def prime(self, pcm):
        self.tail = np.asarray(pcm, dtype=np.float32).reshape(self.n, self.hop).copy()

    def reset_one(self, i: int):
        self.tail[i] = 0; self.ola[i] = 0

    def prime_one(self, i: int, pcm):
        self.tail[i] = np.asarray(pcm, dtype=np.float32).reshape(self.hop)

    def get_state(self, i: int):
        return np.zeros(1, dtype=np.float32)

    def get_tails(self, i: int):
        return self.tail[i].copy(), self.ola[i].copy()

    def set_state(self, i: int, state=None, in_tail=None, ola_tail=None):
        if in_tail is not None:
            self.tail[i] = np.asarray(in_tail, dtype=np.float32)
        if ola_tail is not None:
            self.ola[i] = np.asarray(ola_tail, dtype=np.float32)

    # ---- the library's native pool entry points (dpdf_streams_submit*), restated for host-logic tests: one masked call per hop count
    # of a submission, no cross-thread coalescing (that is the library's job and is tested on the GPU) ----
    def pool_config(self, window_s):
        self._pool_calls = getattr(self, "_pool_calls", 0)

    def pool_tune(self, window_s, regular_window_s, spin_s):
        self._pool_calls = getattr(self, "_pool_calls", 0)

    def slot_use(self, slot, in_use):
        pass

    def pool_stats(self):
        return getattr(self, "_pool_calls", 0), getattr(self, "_pool_calls", 0)

    def submit_many(self, slots, rows, ks, no_window=False):
        import threading
        lock = self.__dict__.setdefault("_pool_lock", threading.Lock())
        outs = [None] * len(slots)
        with lock:
            for k in sorted(set(int(k) for k in ks)):
                idx = [i for i in range(len(slots)) if int(ks[i]) == k]
                pcm = np.zeros((self.n, k * self.hop), dtype=np.float32)
                active = np.zeros(self.n, dtype=bool)
                for i in idx:
                    pcm[slots[i]] = rows[i]; active[slots[i]] = True
                res = self.process_masked(pcm, active)
                self._pool_calls = getattr(self, "_pool_calls", 0) + 1
                for i in idx:
                    outs[i] = res[slots[i]].copy()
        return outs

    def submit_wait(self, slot, pcm, k, no_window=False):
        return self.submit_many([slot], [pcm], [k], no_window)[0]

    def submit_block(self, slots, block, k, no_window=False):
        block = np.asarray(block, dtype=np.float32).reshape(len(slots), -1)
        return np.stack(self.submit_many([int(x) for x in slots], list(block), [k] * len(slots), no_window))

    def process_masked(self, pcm, active):
        """Only the streams with active[i] advance (engine: dpdf_streams_process_masked)."""
        pcm = np.asarray(pcm, dtype=np.float32).reshape(self.n, -1)
        active = np.asarray(active).astype(bool)
        tail, ola = self.tail.copy(), self.ola.copy()
        out = self.process(pcm)
        for i in range(self.n):
            if not active[i]:
                self.tail[i], self.ola[i] = tail[i], ola[i]
                out[i] = 0
        return out

    def process(self, pcm):
        pcm = np.asarray(pcm, dtype=np.float32).reshape(self.n, -1)
        k = pcm.shape[1] // self.hop
        x = np.concatenate([self.tail, pcm], axis=1)
        out = np.zeros_like(pcm)
        for j in range(k):
            fr = x[:, j * self.hop: j * self.hop + self.win] * self.w
            spec = np.fft.rfft(fr, axis=1)
            if self.zero:
                spec = spec * 0
            y = (np.fft.irfft(spec, n=self.win, axis=1) * self.w).astype(np.float32)
            out[:, j * self.hop:(j + 1) * self.hop] = self.ola + y[:, :self.hop]
            self.ola = y[:, self.hop:].copy()
        self.tail = x[:, k * self.hop:].copy()
        return out


class PassthroughSession:
    """Stand-in for backend.HipModel: identity frame function, reference host pipeline in numpy."""

    def __init__(self, win: int = 320, sample_rate: int = 16000, zero: bool = False):
        self.win_len, self.hop, self.sample_rate, self.zero = win, win // 2, sample_rate, zero
        self.freq_bins = win // 2 + 1
        self.state_size = 1
        self.calls = []
        self.ragged_calls = []

    def initial_state(self):
        return np.zeros(1, dtype=np.float32)

    def open_streams(self, n: int):
        return PassthroughStreams(self.win_len, n, self.zero)

    def enhance_batch_ragged(self, clips, attn_limit_db=None):
        """One engine call for clips of different lengths; each clip is processed with its own tail semantics."""
        self.ragged_calls.append([int(len(c)) for c in clips])
        return [self.enhance_batch(np.asarray(c, np.float32)[None], attn_limit_db, _record=False)[0] for c in clips]

    def enhance_batch(self, wav, attn_limit_db=None, _record=True):
        wav = np.asarray(wav, dtype=np.float32)
        if _record:
            self.calls.append((wav.shape, attn_limit_db))
        win, hop, w = self.win_len, self.hop, _window(self.win_len)
        out = np.zeros_like(wav)
        for b in range(wav.shape[0]):
            x = np.pad(wav[b], (0, win))
            xp = np.pad(x, (hop, hop), mode="reflect")
            T = 1 + x.shape[0] // hop
            frames = np.stack([xp[t * hop: t * hop + win] * w for t in range(T)])
            spec = np.fft.rfft(frames, axis=1)
            spec_e = spec * 0 if self.zero else spec.copy()
            if attn_limit_db is not None and np.isfinite(attn_limit_db):
                a = 10.0 ** (-float(attn_limit_db) / 20.0)
                shifted = np.zeros_like(spec)
                shifted[4:] = spec[:-4]
                spec_e = a * shifted + (1 - a) * spec_e
            y = np.zeros(win + hop * (T - 1), dtype=np.float64)
            wss = np.zeros_like(y)
            fr = np.fft.irfft(spec_e, n=win, axis=1) * w
            for t in range(T):
                y[t * hop: t * hop + win] += fr[t]
                wss[t * hop: t * hop + win] += w.astype(np.float64) ** 2
            y = np.where(wss > 1e-30, y / np.maximum(wss, 1e-30), y)[hop: hop + hop * (T - 1)]
            y = np.concatenate([y[2 * win:], np.zeros(2 * win)])
            n = wav.shape[1]
            out[b, :min(n, y.shape[0])] = y[:n]
        return out


# ----- spectrally sparse 48 kHz goldens (tests/golden/make_golden.py: make_sparse_fixtures) -----------------------------
SPARSE_TAGS = ["48k_nb1_sparse", "48k_nb8_sparse"]
# classes the review's bar names "< 1e-4 RMS" for, and the two torture classes (nothing but rounding residue off the signal's bins)
SPARSE_MAIN = ("bl_f32", "bl_i16", "sil_sig")
SPARSE_TORTURE = ("dc", "square")


def oracle_stream(o, wav: np.ndarray) -> np.ndarray:
    """The reference StreamEnhancer's arithmetic around the ORACLE's frame function (oracle/oracle.py: Oracle.stream)."""
    return o.stream(wav)
