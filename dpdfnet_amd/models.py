"""Model registry and weight-file resolution (reference package/src/dpdfnet/models.py:26-69, 369-407).

Network download is out of scope here (no egress): weights are looked up
as ``<name>.npz|.safetensors|.pth`` in ``DPDFNET_MODEL_DIR`` or the cache dir.  For tests, smoke
and benchmarks ``onnx_path="synthetic:<seed>"`` selects the portable seeded weight generator."""
from __future__ import annotations

import os
import sys
from dataclasses import asdict, dataclass
from pathlib import Path
from typing import Any, Dict, List, Optional, Union

from .weights import MODEL_CONFIGS


@dataclass(frozen=True)
class ModelInfo:
    name: str
    sample_rate: int
    frame_ms: float
    description: str
    onnx_filename: str          # kept for drop-in compatibility: stem + ".onnx"
    dprnn_num_blocks: int = 0

    @property
    def stem(self) -> str:
        return Path(self.onnx_filename).stem


_DESCRIPTIONS = {
    "baseline": "Fastest and lowest-compute baseline model.",
    "dpdfnet2": "Balanced quality/speed DPDFNet-2 model.",
    "dpdfnet4": "Higher quality DPDFNet-4 model.",
    "dpdfnet8": "Highest quality 16 kHz DPDFNet-8 model.",
    "dpdfnet2_48khz_hr": "High-resolution 48 kHz DPDFNet-2 model.",
    "dpdfnet8_48khz_hr": "High-resolution 48 kHz DPDFNet-8 model.",
}

MODEL_REGISTRY: Dict[str, ModelInfo] = {
    name: ModelInfo(name=name, sample_rate=sr, frame_ms=20.0, description=_DESCRIPTIONS[name],
                    onnx_filename=f"{name}.onnx", dprnn_num_blocks=nb)
    for name, (sr, nb) in MODEL_CONFIGS.items()
}

DEFAULT_MODEL = "dpdfnet2"
WEIGHT_SUFFIXES = (".npz", ".safetensors", ".pth", ".pt")
SYNTHETIC_PREFIX = "synthetic:"


@dataclass(frozen=True)
class ResolvedModel:
    info: ModelInfo
    onnx_path: Union[Path, str]   # weight file path, or "synthetic:<seed>"


def get_cache_dir() -> Path:
    override = os.environ.get("DPDFNET_CACHE_DIR")
    if override:
        return Path(override).expanduser().resolve()
    if sys.platform == "darwin":
        return (Path.home() / "Library" / "Caches" / "dpdfnet").resolve()
    xdg = os.environ.get("XDG_CACHE_HOME")
    return (Path(xdg) / "dpdfnet" if xdg else Path.home() / ".cache" / "dpdfnet").resolve()


def _candidate_model_dirs() -> List[Path]:
    env_dir = os.environ.get("DPDFNET_MODEL_DIR")
    if env_dir:
        return [Path(env_dir).expanduser().resolve()]
    return [(get_cache_dir() / "models").resolve()]


def supported_models() -> List[str]:
    return sorted(MODEL_REGISTRY)


def get_model_info(model: str) -> ModelInfo:
    try:
        return MODEL_REGISTRY[model]
    except KeyError as exc:
        supported = ", ".join(supported_models())
        raise ValueError(f"Unsupported model '{model}'. Supported: {supported}") from exc


def _is_valid_file(path: Path) -> bool:
    try:
        return path.is_file() and path.stat().st_size > 0
    except OSError:
        return False


def _find_weights(dirs: List[Path], stem: str) -> Optional[Path]:
    for d in dirs:
        for sfx in WEIGHT_SUFFIXES:
            cand = d / f"{stem}{sfx}"
            if _is_valid_file(cand):
                return cand.resolve()
    return None


def resolve_model(*, model: str, onnx_path: Optional[Union[str, Path]] = None, auto_download: bool = True,
                  verbose: bool = False, notifier=None) -> ResolvedModel:
    info = get_model_info(model)
    if onnx_path is not None:
        if isinstance(onnx_path, str) and onnx_path.startswith(SYNTHETIC_PREFIX):
            return ResolvedModel(info=info, onnx_path=onnx_path)
        explicit = Path(onnx_path).expanduser().resolve()
        if not _is_valid_file(explicit):
            raise FileNotFoundError(f"Model weight file not found or empty: {explicit}")
        return ResolvedModel(info=info, onnx_path=explicit)
    dirs = _candidate_model_dirs()
    found = _find_weights(dirs, info.stem)
    if found is None:
        searched = [str(p) for p in dirs]
        raise FileNotFoundError(
            f"Could not resolve weights for '{info.name}'. Searched: {searched} for "
            f"{info.stem}{{{','.join(WEIGHT_SUFFIXES)}}}. Set DPDFNET_CACHE_DIR/DPDFNET_MODEL_DIR, or use the "
            "Python API onnx_path parameter (weight file path)."
        )
    return ResolvedModel(info=info, onnx_path=found)


def available_model_entries() -> List[Dict[str, Any]]:
    dirs = _candidate_model_dirs()
    cache_dir = (get_cache_dir() / "models").resolve()
    rows: List[Dict[str, Any]] = []
    for name in supported_models():
        info = MODEL_REGISTRY[name]
        found = _find_weights(dirs, info.stem)
        row = asdict(info)
        row["onnx_path"] = str(found) if found else None
        row["onnx_found"] = found is not None
        row["ready"] = found is not None
        row["cache_dir"] = str(cache_dir)
        row["cached"] = _find_weights([cache_dir], info.stem) is not None
        rows.append(row)
    return rows
