"""ctypes binding of libdpdfnet_hip.so -- the thin C-ABI layer between the Python host code and the
hand-written gfx950 kernels (include/dpdfnet_hip.h).

This module is the counterpart of the reference's runtime seam
(reference package/src/dpdfnet/onnx_backend.py:11-107: ``RuntimeModel``, ``build_runtime_model``,
``infer_win_len``, ``load_initial_state_from_metadata``).  There is deliberately NO CPU fallback:
if the shared library or a GPU is missing, loading/creating fails loudly.
"""
from __future__ import annotations

import contextlib
import collections
import ctypes
import math
import os
import threading
import weakref
from dataclasses import dataclass
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np

from . import weights as _weights

_PKG_DIR = Path(__file__).resolve().parent
_LIB_NAME = "libdpdfnet_hip.so"
_lib: Optional[ctypes.CDLL] = None
_lib_lock = threading.Lock()

DPDF_HOST_PTRS = 0
DPDF_DEVICE_PTRS = 1


class DpdfCfg(ctypes.Structure):
    _fields_ = [("sample_rate", ctypes.c_int), ("nb", ctypes.c_int)]


class DpdfDims(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in
                ("sr", "nb", "win", "hop", "F", "E", "Ec", "D", "C", "H", "O", "s1", "s2", "s3",
                 "F1", "F2", "F3", "Fd", "emb", "is48")] + [("wnorm", ctypes.c_float), ("state_size", ctypes.c_int)]


EXPORTED_SYMBOLS = (
    "dpdf_abi_version", "dpdf_last_error", "dpdf_device_count", "dpdf_weight_count", "dpdf_manifest_text",
    "dpdf_query_dims", "dpdf_create", "dpdf_destroy", "dpdf_set_norm_init", "dpdf_state_size",
    "dpdf_initial_state", "dpdf_win_len", "dpdf_hop", "dpdf_freq_bins", "dpdf_sample_rate",
    "dpdf_run_frames", "dpdf_enhance_batch", "dpdf_num_frames", "dpdf_streams_create",
    "dpdf_streams_destroy", "dpdf_streams_reset", "dpdf_streams_prime", "dpdf_streams_process",
    "dpdf_streams_get_state", "dpdf_sync", "dpdf_profile_enable", "dpdf_profile_report",
    "dpdf_set_chunk_frames", "dpdf_set_overlap", "dpdf_set_fuse_dprnn", "dpdf_debug_fetch",
    "dpdf_resample_len", "dpdf_resample", "dpdf_enhance_batch_ragged", "dpdf_debug_raise_device_error",
    "dpdf_set_option", "dpdf_streams_process_masked", "dpdf_streams_set_state", "dpdf_streams_get_tails",
    "dpdf_streams_prime_one", "dpdf_streams_is_primed", "dpdf_recovery_count", "dpdf_progress",
    "dpdf_enhance_batch_rows", "dpdf_streams_pool_config", "dpdf_streams_pool_tune", "dpdf_streams_slot_use", "dpdf_streams_submit_wait",
    "dpdf_streams_submit_many", "dpdf_streams_submit_block", "dpdf_streams_pool_stats", "dpdf_streams_pool_timing",
)


def lib_path() -> Path:
    override = os.environ.get("DPDFNET_HIP_LIB")       # tools/: timing ablation builds of the same library
    return Path(override) if override else _PKG_DIR / _LIB_NAME


def _elf_dynamic(path: Path):
    """(SONAME, [NEEDED...]) of a 64-bit little-endian ELF shared object, read from its dynamic section (no external tools)."""
    import struct
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"\x7fELF" or data[4] != 2 or data[5] != 1:
        raise ValueError("not a 64-bit little-endian ELF file")
    shoff, = struct.unpack_from("<Q", data, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", data, 0x3A)
    secs = [struct.unpack_from("<IIQQQQIIQQ", data, shoff + i * shentsize) for i in range(shnum)]
    soname, needed = None, []
    for sec in secs:
        if sec[1] != 6:                      # SHT_DYNAMIC
            continue
        strtab = secs[sec[6]]                # sh_link: its string table
        for off in range(sec[4], sec[4] + sec[5], 16):
            tag, val = struct.unpack_from("<qQ", data, off)
            if tag == 0:
                break
            if tag in (1, 14):               # DT_NEEDED, DT_SONAME
                beg = strtab[4] + val
                name = data[beg:data.index(b"\0", beg)].decode()
                if tag == 1:
                    needed.append(name)
                else:
                    soname = name
    return soname, needed


_preload_note = ""       # what the loader did about torch's bundled HIP runtime (one line; `preload_note()`)


def preload_note() -> str:
    return _preload_note


def _preload_torch_hip_runtime(engine: Path) -> None:
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own `libamdhip64` / `libhsa-runtime64` under torch/lib, and
    whichever copy of a soname is loaded first serves everybody: with the system copy first (this library loaded before torch) a
    later `torch.cuda` initialisation finds "No HIP GPUs"; with torch's copy first both work (bench.py's order).  So when torch is
    installed but not imported yet, its bundled runtime is loaded here first -- by path, torch itself is NOT imported -- but ONLY
    if BOTH bundled sonames are exactly the ones the engine library was linked against (read from the ELF headers): a wheel
    whose HSA soname differs while its HIP soname matches would bind the system HIP to the wheel's older HSA, and is left alone
    (the engine then runs on the system runtime; import torch first if both are needed).  DPDFNET_TORCH_HIP_PRELOAD=0 never
    preloads, =1 forces it; every outcome is kept in `preload_note()` and printed when DPDFNET_VERBOSE is set."""
    global _preload_note
    import sys

    def note(msg: str) -> None:
        global _preload_note
        _preload_note = msg
        if os.environ.get("DPDFNET_VERBOSE", "") not in ("", "0"):
            print(f"[dpdfnet_amd] {msg}", file=sys.stderr)

    mode = os.environ.get("DPDFNET_TORCH_HIP_PRELOAD", "")
    if os.environ.get("DPDFNET_NO_TORCH_HIP_PRELOAD", "") not in ("", "0"):      # earlier spelling of the opt-out
        mode = "0"
    if mode == "0":
        return note("torch HIP runtime preload: off (DPDFNET_TORCH_HIP_PRELOAD=0)")
    if "torch" in sys.modules:
        return note("torch HIP runtime preload: not needed (torch is imported already: its runtime is the process's runtime)")
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return note("torch HIP runtime preload: torch is not installed")
        libdir = Path(spec.origin).parent / "lib"
        files = [libdir / "libhsa-runtime64.so", libdir / "libamdhip64.so"]
        if not all(f.is_file() for f in files):
            return note("torch HIP runtime preload: this torch build bundles no HIP runtime")
        if mode != "1":
            _, needed = _elf_dynamic(engine)
            want_hip = next((n for n in needed if n.startswith("libamdhip64.so")), None)
            hip_soname, _ = _elf_dynamic(files[1])
            hsa_soname, _ = _elf_dynamic(files[0])
            sys_hsa = None          # the HSA soname the SYSTEM HIP runtime wants (what the engine would get without the preload)
            for d in ("/opt/rocm/lib", "/opt/rocm/lib64"):
                cand = Path(d) / (want_hip or "libamdhip64.so")
                if cand.is_file():
                    sys_hsa = next((n for n in _elf_dynamic(cand.resolve())[1] if n.startswith("libhsa-runtime64.so")), None)
                    break
            if hip_soname != want_hip or (sys_hsa is not None and hsa_soname != sys_hsa):
                return note(f"torch HIP runtime preload: SKIPPED, sonames differ (engine needs {want_hip}, system HSA {sys_hsa}; torch bundles "
                            f"{hip_soname} / {hsa_soname}): the engine runs on the system ROCm; import torch BEFORE dpdfnet_amd if both are used")
        for f in files:
            ctypes.CDLL(str(f), mode=ctypes.RTLD_GLOBAL)
        note(f"torch HIP runtime preload: loaded {files[1]} + {files[0].name} (same sonames as the engine's ROCm) so that a later `import torch` "
             "shares the process's one HIP runtime")
    except Exception as exc:          # best effort: the engine itself runs on either copy
        note(f"torch HIP runtime preload: failed ({type(exc).__name__}: {exc}); the engine runs on the system ROCm")


def load_library() -> ctypes.CDLL:
    """Load the HIP extension.  Raises RuntimeError (never falls back) when it is missing."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        p = lib_path()
        if not p.is_file():
            raise RuntimeError(
                f"MI355X HIP extension not built: {p} is missing. "
                "Run `python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc)."
            )
        _preload_torch_hip_runtime(p)
        try:
            L = ctypes.CDLL(str(p))
        except OSError as exc:
            raise RuntimeError(f"Failed to load the MI355X HIP extension {p}: {exc}") from exc
        fp = ctypes.POINTER(ctypes.c_float)
        vp = ctypes.c_void_p
        cfgp = ctypes.POINTER(DpdfCfg)
        L.dpdf_abi_version.restype = ctypes.c_int
        L.dpdf_last_error.restype = ctypes.c_char_p
        L.dpdf_device_count.restype = ctypes.c_int
        L.dpdf_weight_count.restype = ctypes.c_size_t
        L.dpdf_weight_count.argtypes = [cfgp]
        L.dpdf_manifest_text.restype = ctypes.c_size_t
        L.dpdf_manifest_text.argtypes = [cfgp, ctypes.c_char_p, ctypes.c_size_t]
        L.dpdf_query_dims.argtypes = [cfgp, ctypes.POINTER(DpdfDims)]
        L.dpdf_create.argtypes = [cfgp, fp, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(vp)]
        L.dpdf_destroy.argtypes = [vp]
        L.dpdf_destroy.restype = None
        L.dpdf_set_norm_init.argtypes = [vp, fp, ctypes.c_int, fp, ctypes.c_int]
        for fn in ("dpdf_state_size", "dpdf_win_len", "dpdf_hop", "dpdf_freq_bins", "dpdf_sample_rate", "dpdf_sync"):
            getattr(L, fn).argtypes = [vp]
        L.dpdf_initial_state.argtypes = [vp, fp]
        L.dpdf_run_frames.argtypes = [vp, vp, ctypes.c_int, ctypes.c_int, vp, vp, ctypes.c_int]
        L.dpdf_enhance_batch.argtypes = [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_float, vp, ctypes.c_int]
        L.dpdf_num_frames.argtypes = [vp, ctypes.c_int]
        L.dpdf_enhance_batch_ragged.argtypes = [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_float, vp, ctypes.c_int]
        L.dpdf_enhance_batch_rows.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                              ctypes.POINTER(vp), ctypes.c_int]
        L.dpdf_debug_raise_device_error.argtypes = [vp]
        L.dpdf_set_option.argtypes = [vp, ctypes.c_char_p, ctypes.c_int]
        L.dpdf_streams_create.argtypes = [vp, ctypes.c_int, ctypes.POINTER(vp)]
        L.dpdf_streams_destroy.argtypes = [vp]
        L.dpdf_streams_destroy.restype = None
        L.dpdf_streams_reset.argtypes = [vp, ctypes.c_int]
        L.dpdf_streams_prime.argtypes = [vp, vp, ctypes.c_int]
        L.dpdf_streams_process.argtypes = [vp, vp, ctypes.c_int, vp, ctypes.c_int]
        L.dpdf_streams_get_state.argtypes = [vp, ctypes.c_int, fp]
        L.dpdf_streams_process_masked.argtypes = [vp, vp, ctypes.c_int, vp, ctypes.c_char_p, ctypes.c_int]
        L.dpdf_streams_set_state.argtypes = [vp, ctypes.c_int, vp, vp, vp]
        L.dpdf_streams_get_tails.argtypes = [vp, ctypes.c_int, vp, vp]
        L.dpdf_streams_prime_one.argtypes = [vp, ctypes.c_int, vp]
        L.dpdf_streams_is_primed.argtypes = [vp, ctypes.c_int]
        L.dpdf_progress.argtypes = [vp]
        L.dpdf_streams_pool_config.argtypes = [vp, ctypes.c_double]
        L.dpdf_streams_pool_tune.argtypes = [vp, ctypes.c_double, ctypes.c_double, ctypes.c_double]
        L.dpdf_streams_slot_use.argtypes = [vp, ctypes.c_int, ctypes.c_int]
        L.dpdf_streams_submit_wait.argtypes = [vp, ctypes.c_int, vp, ctypes.c_int, vp, ctypes.c_int]
        L.dpdf_streams_submit_many.argtypes = [vp, ctypes.c_int, vp, vp, vp, vp, ctypes.c_int]
        L.dpdf_streams_submit_block.argtypes = [vp, ctypes.c_int, vp, vp, ctypes.c_int, vp, ctypes.c_int]
        L.dpdf_streams_pool_timing.argtypes = [vp, ctypes.POINTER(ctypes.c_double)]
        L.dpdf_streams_pool_stats.argtypes = [vp, ctypes.POINTER(ctypes.c_long), ctypes.POINTER(ctypes.c_long)]
        L.dpdf_recovery_count.argtypes = [vp]
        L.dpdf_recovery_count.restype = ctypes.c_long
        L.dpdf_profile_enable.argtypes = [vp, ctypes.c_int]
        L.dpdf_profile_report.restype = ctypes.c_size_t
        L.dpdf_profile_report.argtypes = [vp, ctypes.c_char_p, ctypes.c_size_t]
        L.dpdf_set_chunk_frames.argtypes = [vp, ctypes.c_int]
        L.dpdf_set_overlap.argtypes = [vp, ctypes.c_int]
        L.dpdf_set_fuse_dprnn.argtypes = [vp, ctypes.c_int]
        L.dpdf_debug_fetch.restype = ctypes.c_long
        L.dpdf_debug_fetch.argtypes = [vp, ctypes.c_char_p, fp, ctypes.c_long]
        L.dpdf_resample_len.restype = ctypes.c_long
        L.dpdf_resample_len.argtypes = [ctypes.c_long, ctypes.c_int, ctypes.c_int]
        L.dpdf_resample.argtypes = [ctypes.c_int, vp, ctypes.c_int, ctypes.c_long, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int]
        _lib = L
        return L


class _HostBlockPool:
    """Recycled host memory for the outputs of the batch calls.

    A fresh 164 MB result block costs the kernel ~40 000 page faults (zeroed pages) while the library scatters into it and a
    page-table teardown when it is dropped: 5-8 ms per 256 x 10 s call, as much as the whole H2D / D2H pipeline leaves exposed.
    Blocks are therefore leased: `take(n)` hands out a flat float32 array of n elements on a block from the free list (or a new
    one); the results of a call are row views of it; when the LAST view of a lease is garbage collected the block returns to the
    free list (a ctypes array is the views' common base object; its weakref finaliser gives the block back).  Consequence: one
    kept result keeps its whole block alive (copy what you keep beyond the next call if that matters).  DPDFNET_OUTPUT_POOL_MB
    (default 512: three result blocks of a 256 x 10 s call) bounds the idle memory kept, and a request that no idle block fits evicts idle
    blocks of other sizes down to that bound before it allocates; 0 turns leasing off (every result its own fresh array)."""

    def __init__(self) -> None:
        self._lock = threading.Lock()
        self._free: List[np.ndarray] = []
        # blocks handed back by finalisers.  A finaliser can run at ANY allocation -- including one made by this class while it
        # holds `_lock` (cyclic GC) -- so it must never take the lock: it only appends here (deque.append is atomic), and take()
        # moves the blocks to the free list under the lock.
        self._returned: "collections.deque[np.ndarray]" = collections.deque()
        self.limit_bytes = int(float(os.environ.get("DPDFNET_OUTPUT_POOL_MB", "512")) * (1 << 20))
        self.leases = 0          # statistics (tests)
        self.reused = 0

    def _give_back(self, raw: np.ndarray) -> None:
        self._returned.append(raw)

    def _drain_returned(self) -> None:
        """(under `_lock`) returned blocks -> free list, up to the idle-memory bound."""
        held = sum(b.nbytes for b in self._free)
        while True:
            try:
                raw = self._returned.popleft()
            except IndexError:
                return
            if held + raw.nbytes <= self.limit_bytes:
                self._free.append(raw)
                held += raw.nbytes

    def idle_blocks(self) -> int:
        """Blocks currently kept for reuse (returned ones included)."""
        with self._lock:
            self._drain_returned()
            return len(self._free)

    def take(self, n: int) -> np.ndarray:
        n = int(n)
        if self.limit_bytes <= 0 or n * 4 < (1 << 20):          # small results: plain arrays
            return np.empty(n, dtype=np.float32)
        raw = None
        with self._lock:
            self.leases += 1
            self._drain_returned()
            best = -1
            for i, b in enumerate(self._free):
                if n <= b.size <= 2 * n + 1024 and (best < 0 or b.size < self._free[best].size):
                    best = i
            if best >= 0:
                raw = self._free.pop(best)
                self.reused += 1
            else:
                # a new size class: idle blocks of other sizes do not pile up beside it (largest first)
                self._free.sort(key=lambda b: b.nbytes)
                while self._free and sum(b.nbytes for b in self._free) + n * 4 > self.limit_bytes:
                    self._free.pop()
        if raw is None:
            raw = np.empty(n, dtype=np.float32)
        lease = (ctypes.c_float * raw.size).from_buffer(raw)          # shares raw's memory and keeps raw alive
        weakref.finalize(lease, self._give_back, raw)
        return np.frombuffer(lease, dtype=np.float32)[:n]           # .base (and every view's base) is `lease`


_out_pool = _HostBlockPool()


def _fp(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _check(rc: int) -> None:
    """Map C status codes to the reference's exception types."""
    if rc == 0:
        return
    msg = load_library().dpdf_last_error().decode("utf-8", "replace")
    if rc == -2:
        raise RuntimeError(msg)
    raise ValueError(msg)


def manifest(sample_rate: int, nb: int) -> List[_weights.ManifestEntry]:
    L = load_library()
    cfg = DpdfCfg(int(sample_rate), int(nb))
    n = L.dpdf_manifest_text(ctypes.byref(cfg), None, 0)
    if n == 0:
        raise ValueError(f"Unsupported model config sample_rate={sample_rate} nb={nb}")
    buf = ctypes.create_string_buffer(n + 1)
    L.dpdf_manifest_text(ctypes.byref(cfg), buf, n + 1)
    return _weights.parse_manifest_text(buf.value.decode("utf-8"))


def query_dims(sample_rate: int, nb: int) -> DpdfDims:
    d = DpdfDims()
    cfg = DpdfCfg(int(sample_rate), int(nb))
    _check(load_library().dpdf_query_dims(ctypes.byref(cfg), ctypes.byref(d)))
    return d


def device_count() -> int:
    return int(load_library().dpdf_device_count())


def resample(x: np.ndarray, sr_in: int, sr_out: int, device: int = 0) -> np.ndarray:
    """Device polyphase resampler (`dpdf_resample`): [n] or [B, n] float32 at sr_in -> same rank at sr_out."""
    L = load_library()
    a = np.ascontiguousarray(x, dtype=np.float32)
    one = a.ndim == 1
    a2 = a[None, :] if one else a
    if a2.ndim != 2:
        raise ValueError(f"Expected [n] or [B, n] audio, got shape {a.shape}")
    B, n = a2.shape
    n_out = int(L.dpdf_resample_len(n, int(sr_in), int(sr_out)))
    if n_out < 0:
        raise ValueError(f"bad resample arguments n={n} {sr_in}->{sr_out}")
    out = np.empty((B, n_out), dtype=np.float32)
    if B and n:
        _check(L.dpdf_resample(int(device), a2.ctypes.data, B, n, int(sr_in), int(sr_out), out.ctypes.data, DPDF_HOST_PTRS))
    return out[0] if one else out


class HipModel:
    """A frame-function handle on one GPU (the analogue of an ORT ``InferenceSession``)."""

    def __init__(self, sample_rate: int, nb: int, blob: np.ndarray, device: int = 0,
                 erb_norm_init: Optional[np.ndarray] = None, spec_norm_init: Optional[np.ndarray] = None):
        self._L = load_library()
        self.cfg = DpdfCfg(int(sample_rate), int(nb))
        blob = np.ascontiguousarray(blob, dtype=np.float32).reshape(-1)
        h = ctypes.c_void_p()
        _check(self._L.dpdf_create(ctypes.byref(self.cfg), _fp(blob), blob.size, int(device), ctypes.byref(h)))
        self._h = h
        self._call_lock = threading.Lock()      # one offline host call at a time per handle (the C side serialises them anyway) ...
        self._call_owner: Optional[int] = None  # ... so that `progress(owner=...)` can tell WHOSE call the counter belongs to
        self.device = int(device)
        self.sample_rate = int(sample_rate)
        self.nb = int(nb)
        self.state_size = int(self._L.dpdf_state_size(h))
        self.win_len = int(self._L.dpdf_win_len(h))
        self.hop = int(self._L.dpdf_hop(h))
        self.freq_bins = int(self._L.dpdf_freq_bins(h))
        self.dims = query_dims(sample_rate, nb)
        if erb_norm_init is not None or spec_norm_init is not None:
            e = None if erb_norm_init is None else np.ascontiguousarray(erb_norm_init, dtype=np.float32)
            s = None if spec_norm_init is None else np.ascontiguousarray(spec_norm_init, dtype=np.float32)
            _check(self._L.dpdf_set_norm_init(h, None if e is None else _fp(e), 0 if e is None else e.size,
                                              None if s is None else _fp(s), 0 if s is None else s.size))

    def close(self) -> None:
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._L.dpdf_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- reference RuntimeModel surface ------------------------------------------------------
    def initial_state(self) -> np.ndarray:
        st = np.zeros(self.state_size, dtype=np.float32)
        _check(self._L.dpdf_initial_state(self._h, _fp(st)))
        return st

    def num_frames(self, n_samples: int) -> int:
        return int(self._L.dpdf_num_frames(self._h, int(n_samples)))

    def run_frames(self, spec: np.ndarray, state: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        """spec [B,T,F,2] (or [T,F,2]), state [B,S] (or [S]) -> (spec_e, state_out)."""
        spec = np.ascontiguousarray(spec, dtype=np.float32)
        squeeze = spec.ndim == 3
        if squeeze:
            spec = spec[None]
        if spec.ndim != 4 or spec.shape[2] != self.freq_bins or spec.shape[3] != 2:
            raise ValueError(f"spec must be [B,T,{self.freq_bins},2], got {spec.shape}")
        B, T = spec.shape[0], spec.shape[1]
        st = np.array(state, dtype=np.float32).reshape(B, -1)
        if st.shape[1] != self.state_size:
            raise ValueError(f"state must have {self.state_size} values per stream, got {st.shape[1]}")
        st = np.ascontiguousarray(st)
        out = np.empty_like(spec)
        _check(self._L.dpdf_run_frames(self._h, spec.ctypes.data, B, T, st.ctypes.data, out.ctypes.data, DPDF_HOST_PTRS))
        if squeeze:
            return out[0], st[0]
        return out, st

    @contextlib.contextmanager
    def _offline_call(self):
        with self._call_lock:
            self._call_owner = threading.get_ident()
            try:
                yield
            finally:
                self._call_owner = None

    def enhance_batch(self, wav: np.ndarray, attn_limit_db: Optional[float] = None) -> np.ndarray:
        """wav [B,N] float32 at the model rate -> enhanced [B,N]."""
        wav = np.ascontiguousarray(wav, dtype=np.float32)
        if wav.ndim != 2:
            raise ValueError(f"wav must be [B,N], got {wav.shape}")
        out = _out_pool.take(wav.size).reshape(wav.shape)      # recycled host memory (no first-touch page faults; _HostBlockPool)
        db = float("nan") if attn_limit_db is None else float(attn_limit_db)
        with self._offline_call():
            _check(self._L.dpdf_enhance_batch(self._h, wav.ctypes.data, wav.shape[0], wav.shape[1], db, out.ctypes.data, DPDF_HOST_PTRS))
        return out

    def enhance_batch_ragged(self, clips, attn_limit_db: Optional[float] = None) -> List[np.ndarray]:
        """Clips of different (or equal) lengths, each a 1-D float32 array at the model rate in its OWN buffer, in ONE engine
        call (`dpdf_enhance_batch_rows`): the library reads every clip where it lies and writes every result into its own fresh
        array -- no [B, n_max] block is assembled or taken apart here, and the H2D / D2H of the PCM is pipelined under the
        compute inside the library.  Each result equals `enhance_batch` on that clip alone."""
        clips = [np.ascontiguousarray(c, dtype=np.float32).reshape(-1) for c in clips]
        if not clips:
            return []
        B = len(clips)
        lens = np.array([c.shape[0] for c in clips], dtype=np.int32)
        n_max = int(lens.max())
        offs = np.concatenate([[0], np.cumsum(lens, dtype=np.int64)])
        block = _out_pool.take(int(offs[-1]))                  # one leased block, the results are its row views (_HostBlockPool)
        outs = [block[int(offs[i]): int(offs[i + 1])] for i in range(B)]
        if n_max == 0:
            return outs
        live = [i for i in range(B) if lens[i] > 0]          # zero-length clips stay out of the call (their result is empty)
        rows_in = (ctypes.c_void_p * len(live))(*[clips[i].ctypes.data for i in live])
        rows_out = (ctypes.c_void_p * len(live))(*[outs[i].ctypes.data for i in live])
        ll = np.ascontiguousarray(lens[live])
        db = float("nan") if attn_limit_db is None else float(attn_limit_db)
        with self._offline_call():
            _check(self._L.dpdf_enhance_batch_rows(self._h, rows_in, ll.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), len(live), n_max, db,
                                                   rows_out, DPDF_HOST_PTRS))
        return outs

    def enhance_batch_device(self, wav_ptr: int, B: int, N: int, out_ptr: int, attn_limit_db: Optional[float] = None) -> None:
        """Device-pointer form (HBM-resident input/output; asynchronous on the model's stream)."""
        db = float("nan") if attn_limit_db is None else float(attn_limit_db)
        _check(self._L.dpdf_enhance_batch(self._h, wav_ptr, int(B), int(N), db, out_ptr, DPDF_DEVICE_PTRS))

    def open_streams(self, n_streams: int) -> "HipStreams":
        return HipStreams(self, n_streams)

    def sync(self) -> None:
        _check(self._L.dpdf_sync(self._h))

    def progress(self, owner: Optional[int] = None) -> int:
        """Frames of the offline call in flight whose output is complete (lock-free; poll from another thread).
        `owner` = ident of the thread that makes the call being watched: 0 is returned until THAT thread is inside its
        engine call (another thread's call on a shared handle, or the previous call's count, is never reported)."""
        if owner is not None and self._call_owner != owner:
            return 0
        return int(self._L.dpdf_progress(self._h))

    @property
    def recovery_count(self) -> int:
        """Host-pointer calls that were re-run on the non-spinning GRU-256 kernels after a cluster exchange timed out."""
        return int(self._L.dpdf_recovery_count(self._h))

    def debug_raise_device_error(self) -> None:
        """Test hook: make the next synchronisation point report the device-side failure path."""
        _check(self._L.dpdf_debug_raise_device_error(self._h))

    def set_chunk_frames(self, frames: int) -> None:
        _check(self._L.dpdf_set_chunk_frames(self._h, int(frames)))

    def set_overlap(self, mask: int) -> None:
        _check(self._L.dpdf_set_overlap(self._h, int(mask)))

    def set_fuse_dprnn(self, mode) -> None:
        """True/"always": fused GRU-64 epilogues for every chunk; False/"never": separate GEMM
        kernels; "auto" (engine default): fused only when the chunk fills the chip."""
        code = {"never": 0, "auto": 1, "always": 2, False: 0, True: 2}[mode]
        _check(self._L.dpdf_set_fuse_dprnn(self._h, code))

    def set_option(self, name: str, value: int) -> None:
        """Named engine switch (`dpdf_set_option`): every name, default and owner test is in docs/OPTIONS.md."""
        _check(self._L.dpdf_set_option(self._h, name.encode(), int(value)))

    def debug_fetch(self, name: str) -> np.ndarray:
        n = self._L.dpdf_debug_fetch(self._h, name.encode(), None, 0)
        if n < 0:
            raise ValueError(f"unknown debug tensor {name!r}")
        buf = np.empty(n, dtype=np.float32)
        self._L.dpdf_debug_fetch(self._h, name.encode(), _fp(buf), n)
        return buf

    def profile(self, on: bool) -> None:
        _check(self._L.dpdf_profile_enable(self._h, 1 if on else 0))

    def profile_report(self) -> Dict[str, Tuple[float, int]]:
        n = self._L.dpdf_profile_report(self._h, None, 0)
        buf = ctypes.create_string_buffer(n + 1)
        self._L.dpdf_profile_report(self._h, buf, n + 1)
        rep: Dict[str, Tuple[float, int]] = {}
        for line in buf.value.decode().splitlines():
            name, ms, calls = line.split(" ")
            rep[name] = (float(ms), int(calls))
        return rep


class HipStreams:
    """S device-resident streams (state + analysis/OLA tails in HBM)."""

    def __init__(self, model: HipModel, n_streams: int):
        self.model = model
        self.n = int(n_streams)
        h = ctypes.c_void_p()
        _check(model._L.dpdf_streams_create(model._h, self.n, ctypes.byref(h)))
        self._h = h

    def close(self) -> None:
        h, self._h = getattr(self, "_h", None), None
        if h and self.model._h:
            self.model._L.dpdf_streams_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self, stream: int = -1) -> None:
        _check(self.model._L.dpdf_streams_reset(self._h, int(stream)))

    def prime(self, pcm: np.ndarray) -> None:
        pcm = np.ascontiguousarray(pcm, dtype=np.float32).reshape(self.n, self.model.hop)
        _check(self.model._L.dpdf_streams_prime(self._h, pcm.ctypes.data, DPDF_HOST_PTRS))

    def process(self, pcm: np.ndarray) -> np.ndarray:
        pcm = np.ascontiguousarray(pcm, dtype=np.float32).reshape(self.n, -1)
        if pcm.shape[1] % self.model.hop:
            raise ValueError("streams.process needs a whole number of hops per stream")
        n_hops = pcm.shape[1] // self.model.hop
        out = np.empty_like(pcm)
        _check(self.model._L.dpdf_streams_process(self._h, pcm.ctypes.data, n_hops, out.ctypes.data, DPDF_HOST_PTRS))
        return out

    def process_masked(self, pcm: np.ndarray, active) -> np.ndarray:
        """One device call that advances only the streams with active[i] true; rows of the others are returned as zeros
        and their state / tails are untouched (`dpdf_streams_process_masked`)."""
        pcm = np.ascontiguousarray(pcm, dtype=np.float32).reshape(self.n, -1)
        if pcm.shape[1] % self.model.hop:
            raise ValueError("streams.process needs a whole number of hops per stream")
        act = np.ascontiguousarray(np.asarray(active).astype(bool).astype(np.uint8)).reshape(self.n)
        out = np.zeros_like(pcm)
        _check(self.model._L.dpdf_streams_process_masked(self._h, pcm.ctypes.data, pcm.shape[1] // self.model.hop, out.ctypes.data,
                                                         act.tobytes(), DPDF_HOST_PTRS))
        return out

    # ---- native coalescing of independent submitters (dpdf_streams_submit*): the queue, the leader and the window live in the library ----
    def pool_config(self, window_s: float) -> None:
        _check(self.model._L.dpdf_streams_pool_config(self._h, float(window_s)))

    def pool_tune(self, window_s: float, regular_window_s: float, spin_s: float) -> None:
        _check(self.model._L.dpdf_streams_pool_tune(self._h, float(window_s), float(regular_window_s), float(spin_s)))

    def slot_use(self, slot: int, in_use: bool) -> None:
        _check(self.model._L.dpdf_streams_slot_use(self._h, int(slot), 1 if in_use else 0))

    def pool_stats(self) -> Tuple[int, int]:
        a, b = ctypes.c_long(0), ctypes.c_long(0)
        _check(self.model._L.dpdf_streams_pool_stats(self._h, ctypes.byref(a), ctypes.byref(b)))
        return int(a.value), int(b.value)

    def pool_timing(self) -> Tuple[float, float, float]:
        """Seconds summed over all rounds: leaders waiting for other submitters, inside device calls, between device calls."""
        a = (ctypes.c_double * 3)()
        _check(self.model._L.dpdf_streams_pool_timing(self._h, a))
        return float(a[0]), float(a[1]), float(a[2])

    def _hops(self, pcm, k: int, what: str) -> np.ndarray:
        """float32, contiguous, exactly k whole hops: the library reads k * hop floats from the pointer, whatever the array holds."""
        k = int(k)
        if k <= 0:
            raise ValueError(f"{what}: hop count must be positive, got {k}")
        a = np.ascontiguousarray(pcm, dtype=np.float32).reshape(-1)
        if a.size != k * self.model.hop:
            raise ValueError(f"{what}: expected {k} x {self.model.hop} = {k * self.model.hop} samples, got {a.size}")
        return a

    def submit_wait(self, slot: int, pcm: np.ndarray, k: int, no_window: bool = False) -> np.ndarray:
        """k whole hops for one slot -> its k * hop enhanced samples; rides in the round other threads' submissions for this stream
        set are in (ONE masked device call per round)."""
        pcm = self._hops(pcm, k, "submit_wait")
        out = np.empty(int(k) * self.model.hop, dtype=np.float32)
        _check(self.model._L.dpdf_streams_submit_wait(self._h, int(slot), pcm.ctypes.data, int(k), out.ctypes.data, 1 if no_window else 0))
        return out

    def submit_block(self, slots: np.ndarray, block: np.ndarray, k: int, no_window: bool = False) -> np.ndarray:
        """n = len(slots) requests of k hops each, rows of `block` [n, k * hop] -> [n, k * hop]."""
        slots = np.ascontiguousarray(slots, dtype=np.int32).reshape(-1)
        n = int(slots.shape[0])
        block = self._hops(block, int(k) * n, "submit_block").reshape(n, int(k) * self.model.hop) if n else np.zeros((0, int(k) * self.model.hop), np.float32)
        out = np.empty(block.shape, dtype=np.float32)
        if n == 0:
            return out
        rc = self.model._L.dpdf_streams_submit_block(self._h, n, slots.ctypes.data, block.ctypes.data, int(k), out.ctypes.data,
                                                     1 if no_window else 0)
        if rc:
            _check(rc)
        return out

    def submit_many(self, slots: Sequence[int], rows: Sequence[np.ndarray], ks: Sequence[int], no_window: bool = False) -> List[np.ndarray]:
        """Requests with different hop counts: rows[i] holds ks[i] whole hops for slots[i]."""
        n = len(slots)
        if len(rows) != n or len(ks) != n:
            raise ValueError(f"submit_many: {n} slots, {len(rows)} rows, {len(ks)} hop counts")
        hop = self.model.hop
        rows = [self._hops(r, k, f"submit_many[{i}]") for i, (r, k) in enumerate(zip(rows, ks))]
        outs = [np.empty(int(k) * hop, dtype=np.float32) for k in ks]
        sl = (ctypes.c_int * n)(*[int(x) for x in slots]); kk = (ctypes.c_int * n)(*[int(k) for k in ks])
        ip = (ctypes.c_void_p * n)(*[r.ctypes.data for r in rows]); op = (ctypes.c_void_p * n)(*[o.ctypes.data for o in outs])
        _check(self.model._L.dpdf_streams_submit_many(self._h, n, sl, ip, kk, op, 1 if no_window else 0))
        return outs

    def prime_one(self, stream: int, pcm_hop: np.ndarray) -> None:
        pcm_hop = np.ascontiguousarray(pcm_hop, dtype=np.float32).reshape(self.model.hop)
        _check(self.model._L.dpdf_streams_prime_one(self._h, int(stream), pcm_hop.ctypes.data))

    def is_primed(self, stream: int) -> bool:
        return bool(self.model._L.dpdf_streams_is_primed(self._h, int(stream)))

    def get_state(self, stream: int) -> np.ndarray:
        st = np.empty(self.model.state_size, dtype=np.float32)
        _check(self.model._L.dpdf_streams_get_state(self._h, int(stream), _fp(st)))
        return st

    def get_tails(self, stream: int):
        """(analysis tail, overlap-add tail) of one stream, hop floats each (reference stream.py:62-72 buffers)."""
        a = np.empty(self.model.hop, dtype=np.float32); b = np.empty(self.model.hop, dtype=np.float32)
        _check(self.model._L.dpdf_streams_get_tails(self._h, int(stream), a.ctypes.data, b.ctypes.data))
        return a, b

    def set_state(self, stream: int, state=None, in_tail=None, ola_tail=None) -> None:
        """Resume a stream from a saved reference-layout state vector and (optionally) its two hop-sized buffers."""
        def ptr(x, n):
            if x is None:
                return None, None
            x = np.ascontiguousarray(x, dtype=np.float32).reshape(n)
            return x, x.ctypes.data
        ks, ps = ptr(state, self.model.state_size); ki, pi = ptr(in_tail, self.model.hop); ko, po = ptr(ola_tail, self.model.hop)
        _check(self.model._L.dpdf_streams_set_state(self._h, int(stream), ps, pi, po))
