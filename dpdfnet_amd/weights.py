"""Weight handling for the MI355X DPDFNet engine.

* the canonical flat fp32 weight blob (layout = include/dpdf_manifest.h),
* packing a PyTorch ``state_dict`` (offline-twin *or* streaming key names, grouped linears in
  "loop" *or* "einsum" form) into that blob -- the counterpart of the reference's
  ``correct_state_dict`` + ``convert_grouped_linear_to_einsum``
  (reference onnx_model/dpdfnet.py:876-888, onnx_model/layers.py:1053-1080,
  onnx_model/export_dpdfnet_to_onnx.py:86-111),
* a portable seeded synthetic-weight generator (no checkpoints exist offline; both the
  golden-vector script and the tests regenerate identical weights from the seed).
"""
from __future__ import annotations

import re
import zlib
from dataclasses import dataclass
from pathlib import Path
from typing import Dict, List, Mapping, Optional, Tuple, Union

import numpy as np

MODEL_CONFIGS: Dict[str, Tuple[int, int]] = {
    # name -> (sample_rate, dprnn_num_blocks)   (reference package/src/dpdfnet/models.py:26-69,
    # onnx_model/export_dpdfnet_to_onnx.py:86-100)
    "baseline": (16000, 0),
    "dpdfnet2": (16000, 2),
    "dpdfnet4": (16000, 4),
    "dpdfnet8": (16000, 8),
    "dpdfnet2_48khz_hr": (48000, 2),
    "dpdfnet8_48khz_hr": (48000, 8),
}


@dataclass(frozen=True)
class ManifestEntry:
    name: str
    offset: int
    count: int
    shape: Tuple[int, ...]


def parse_manifest_text(text: str) -> List[ManifestEntry]:
    out: List[ManifestEntry] = []
    for line in text.strip().splitlines():
        name, off, cnt, shp = line.split(" ")
        out.append(ManifestEntry(name, int(off), int(cnt), tuple(int(s) for s in shp.split(","))))
    return out


def manifest_total(entries: List[ManifestEntry]) -> int:
    return entries[-1].offset + entries[-1].count if entries else 0


# ----------------------------------------------------------------------------------------------
# synthetic weights
# ----------------------------------------------------------------------------------------------
def _rng_for(name: str, seed: int) -> np.random.Generator:
    return np.random.default_rng([int(seed) & 0x7FFFFFFF, zlib.crc32(name.encode("utf-8"))])


def synth_tensor(name: str, shape: Tuple[int, ...], seed: int) -> np.ndarray:
    """Deterministic pseudo-checkpoint tensor for manifest entry `name`.

    Scales are chosen so that signals neither vanish nor blow up through the random network and
    every recurrence stays contractive (|W_hh| rows well inside the unit ball)."""
    rng = _rng_for(name, seed)
    n = int(np.prod(shape))
    u = lambda lo, hi: rng.uniform(lo, hi, size=n).astype(np.float32).reshape(shape)
    if name.endswith(".running_var"):
        return u(0.6, 1.4)
    if name.endswith(".running_mean"):
        return u(-0.15, 0.15)
    is_1d = len(shape) == 1
    if is_1d and name.endswith(".weight"):           # BN gamma / LayerNorm gamma
        return u(0.8, 1.2)
    if is_1d and name.endswith(".bias") or re.search(r"\.bias(_ih|_hh)", name):
        return u(-0.1, 0.1)
    # matrices / kernels
    if "weight_hh" in name:
        fan_in, gain = shape[1], 0.9
    elif "weight_ih" in name:
        fan_in, gain = shape[1], 1.2
    elif len(shape) == 4:                            # conv [out, in/g, kt, kf]
        fan_in = shape[1] * shape[2] * shape[3]
        gain = 1.4
        if shape[1] == 1 and shape[2] == 1 and shape[3] == 1:   # pathway per-channel scale
            return (u(0.5, 1.2) * np.where(rng.uniform(size=shape) < 0.15, -1.0, 1.0)).astype(np.float32)
    elif len(shape) == 3:                            # grouped linear [G, Og, Ig]
        fan_in, gain = shape[2], 1.3
    else:                                            # nn.Linear [out, in]
        fan_in, gain = shape[1], 1.2
    a = gain * np.sqrt(3.0 / fan_in)
    return u(-a, a)


def synth_blob(entries: List[ManifestEntry], seed: int) -> np.ndarray:
    blob = np.zeros(manifest_total(entries), dtype=np.float32)
    for e in entries:
        blob[e.offset:e.offset + e.count] = synth_tensor(e.name, e.shape, seed).reshape(-1)
    return blob


STRESS_KINDS = ("hot", "stiff")


def stress_blob(entries: List[ManifestEntry], seed: int, kind: str) -> np.ndarray:
    """Seeded synthetic weights pushed into the corners a trained checkpoint can sit in (the plain `synth_blob` keeps every
    recurrence contractive and every normalisation near unit scale):
      "hot":   every GRU matrix and bias, fc_intra and fc_inter x 3 -- gates deep in saturation (sigmoid / tanh tails, where
               the engine's exp2 / rcp forms and folded scales could part from the reference's);
      "stiff": every third BatchNorm channel gets a running_var of 1e-4 .. 5e-4 (gamma rescaled so that the layer's output
               scale stays where it was: it is the 1 / sqrt(var + eps) fold at var ~ eps that is exercised, not an
               exploding network), LayerNorm gains x 5.
    Used by tests/golden/make_golden.py (reference side) and by the parity tests."""
    if kind not in STRESS_KINDS:
        raise ValueError(f"unknown stress kind '{kind}'")
    blob = synth_blob(entries, seed)
    by_name = {e.name: e for e in entries}
    for e in entries:
        v = blob[e.offset:e.offset + e.count]
        if kind == "hot":
            if re.search(r"weight_ih|weight_hh|bias_ih|bias_hh|fc_intra\.|fc_inter\.", e.name):
                v *= np.float32(3.0)
        else:
            if e.name.endswith(".running_var"):
                idx = np.arange(e.count)
                sel = idx % 3 == 0
                new = (1e-4 * (1 + idx % 5)).astype(np.float32)
                gam = by_name[e.name[: -len("running_var")] + "weight"]
                g = blob[gam.offset:gam.offset + gam.count]
                ratio = np.sqrt((new.astype(np.float64) + 1e-5) / (v.astype(np.float64) + 1e-5)).astype(np.float32)
                g[sel] *= ratio[sel]
                v[sel] = new[sel]
            elif re.search(r"ln_(intra|inter)\.weight$", e.name):
                v *= np.float32(5.0)
    return blob


# ----------------------------------------------------------------------------------------------
# state_dict <-> blob
# ----------------------------------------------------------------------------------------------
def streaming_key(k: str) -> str:
    """Offline-twin checkpoint key -> streaming-module key (reference onnx_model/dpdfnet.py:876-888)."""
    if "grucell" in k:                       # already a streaming-module key
        return k
    if "inter_gru" in k:
        return k.replace("_l0", "").replace("inter_gru.", "inter_gru.grucell.")
    if "gru.gru" in k:
        layer = k[-1]
        return k[:-3].replace(".gru.", f".gru.{layer}.grucell.")
    return k


def _to_np(v) -> np.ndarray:
    if hasattr(v, "detach"):
        v = v.detach().cpu().numpy()
    return np.asarray(v, dtype=np.float32)


def pack_state_dict(entries: List[ManifestEntry], state_dict: Mapping[str, object]) -> np.ndarray:
    """Flatten a checkpoint into the canonical blob.  Accepts offline or streaming key names and
    grouped linears as ``<p>.layers.{g}.weight [Og,Ig]`` (loop) or ``<p>.weight [G,Ig,Og]`` (einsum)."""
    sd = {streaming_key(k): v for k, v in state_dict.items()}
    blob = np.zeros(manifest_total(entries), dtype=np.float32)
    for e in entries:
        if e.name in sd and tuple(_to_np(sd[e.name]).shape) == e.shape:
            arr = _to_np(sd[e.name])
        elif len(e.shape) == 3 and e.name.endswith(".weight"):
            prefix = e.name[: -len(".weight")]
            G = e.shape[0]
            if f"{prefix}.layers.0.weight" in sd:
                arr = np.stack([_to_np(sd[f"{prefix}.layers.{g}.weight"]) for g in range(G)], axis=0)
            elif e.name in sd:                       # einsum storage [G, Ig, Og] (reference onnx_model/layers.py:976-1018, 1053-1080)
                arr = _to_np(sd[e.name])
                if arr.ndim != 3 or (arr.shape[0], arr.shape[2], arr.shape[1]) != e.shape:
                    raise ValueError(f"tensor {e.name}: expected {e.shape} or its einsum form "
                                     f"{(e.shape[0], e.shape[2], e.shape[1])}, got {arr.shape}")
                arr = np.transpose(arr, (0, 2, 1))
            else:
                raise KeyError(f"checkpoint lacks grouped-linear weight for {prefix}")
        elif e.name.endswith(".bias") and f"{e.name[:-5]}.layers.0.bias" in sd:
            prefix = e.name[:-5]
            G = 0
            while f"{prefix}.layers.{G}.bias" in sd:
                G += 1
            arr = np.concatenate([_to_np(sd[f"{prefix}.layers.{g}.bias"]) for g in range(G)], axis=0)
        elif e.name in sd:
            arr = _to_np(sd[e.name])
        else:
            raise KeyError(f"checkpoint lacks tensor {e.name}")
        if tuple(arr.shape) != e.shape:
            # the only reshape accepted is PyTorch's storage of the same tensor with unit axes dropped / added
            squeeze = lambda shp: tuple(d for d in shp if d != 1)
            if int(arr.size) != e.count or squeeze(tuple(arr.shape)) != squeeze(e.shape):
                raise ValueError(f"tensor {e.name}: expected {e.shape}, got {arr.shape}")
        blob[e.offset:e.offset + e.count] = np.ascontiguousarray(arr, dtype=np.float32).reshape(-1)
    return blob


def unpack_to_streaming_state_dict(entries: List[ManifestEntry], blob: np.ndarray) -> Dict[str, np.ndarray]:
    """Blob -> streaming-module ``state_dict`` (grouped linears in the reference's default "loop"
    form).  Used by the golden-vector script to load OUR synthetic weights into the reference."""
    sd: Dict[str, np.ndarray] = {}
    gl_prefixes = {e.name[: -len(".weight")] for e in entries if len(e.shape) == 3}
    for e in entries:
        arr = blob[e.offset:e.offset + e.count].reshape(e.shape)
        prefix = e.name.rsplit(".", 1)[0]
        if prefix in gl_prefixes:
            if e.name.endswith(".weight"):
                for g in range(e.shape[0]):
                    sd[f"{prefix}.layers.{g}.weight"] = arr[g].copy()
            else:
                G = next(x.shape[0] for x in entries if x.name == prefix + ".weight")
                og = e.count // G
                for g in range(G):
                    sd[f"{prefix}.layers.{g}.bias"] = arr[g * og:(g + 1) * og].copy()
        else:
            sd[e.name] = arr.copy()
    return sd


# ----------------------------------------------------------------------------------------------
# weight files
# ----------------------------------------------------------------------------------------------
def load_weight_file(path: Union[str, Path], entries: List[ManifestEntry]) -> Tuple[np.ndarray, Dict[str, np.ndarray]]:
    """Read a weight file into (blob, extras).  Formats: ``.npz`` holding either ``blob`` or named
    tensors, ``.safetensors``, or a PyTorch ``.pth`` state_dict (PyTorch is used ONLY here, for
    deserialisation).  ``extras`` may carry ``erb_norm_init`` / ``spec_norm_init`` -- the values the
    reference embeds as ONNX metadata (export_dpdfnet_to_onnx.py:59-83)."""
    p = Path(path).expanduser().resolve()
    if not p.is_file() or p.stat().st_size == 0:
        raise FileNotFoundError(f"Model weight file not found or empty: {p}")
    extras: Dict[str, np.ndarray] = {}
    suffix = p.suffix.lower()
    if suffix == ".npz":
        with np.load(str(p)) as z:
            keys = set(z.files)
            for k in ("erb_norm_init", "spec_norm_init"):
                if k in keys:
                    extras[k] = np.asarray(z[k], dtype=np.float32)
            if "blob" in keys:
                blob = np.asarray(z["blob"], dtype=np.float32).reshape(-1)
                if blob.size != manifest_total(entries):
                    raise ValueError(f"{p}: blob has {blob.size} floats, model needs {manifest_total(entries)}")
                return blob, extras
            sd = {k: z[k] for k in keys}
        return pack_state_dict(entries, sd), extras
    if suffix == ".safetensors":
        from safetensors.numpy import load_file
        sd = load_file(str(p))
        for k in ("erb_norm_init", "spec_norm_init"):
            if k in sd:
                extras[k] = np.asarray(sd.pop(k), dtype=np.float32)
        return pack_state_dict(entries, sd), extras
    if suffix in (".pth", ".pt", ".ckpt"):
        import os
        import torch
        try:
            sd = torch.load(str(p), map_location="cpu", weights_only=True)
        except Exception as exc:
            # The safe loader refuses anything but plain tensors/containers.  Full unpickling executes code from the
            # file, so it is never a silent fallback: explicit opt-in only.
            if os.environ.get("DPDFNET_ALLOW_UNSAFE_PICKLE") != "1":
                raise ValueError(
                    f"{p}: not a plain tensor state_dict (torch.load(weights_only=True) refused it: {exc}). Re-save it as "
                    "a state_dict / .safetensors, or set DPDFNET_ALLOW_UNSAFE_PICKLE=1 if you trust the file.") from exc
            sd = torch.load(str(p), map_location="cpu", weights_only=False)
        if isinstance(sd, dict) and "state_dict" in sd and isinstance(sd["state_dict"], dict):
            sd = sd["state_dict"]
        if not isinstance(sd, dict):
            raise ValueError(f"{p}: expected a state_dict mapping, got {type(sd).__name__}")
        return pack_state_dict(entries, sd), extras
    raise ValueError(f"Unsupported weight file format {suffix!r}: {p}")


def save_blob(path: Union[str, Path], blob: np.ndarray, **extras: np.ndarray) -> Path:
    p = Path(path)
    p.parent.mkdir(parents=True, exist_ok=True)
    np.savez(str(p), blob=np.asarray(blob, dtype=np.float32), **extras)
    return p
