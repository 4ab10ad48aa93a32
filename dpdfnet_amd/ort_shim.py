"""The engine behind the reference's OWN runtime seam, unchanged callers.

The reference reaches its frame function through `RuntimeModel(session, init_state, in_spec_name, in_state_name,
out_spec_name, out_state_name)` built by `build_runtime_model(onnx_path)` (reference
package/src/dpdfnet/onnx_backend.py:11-18, 81-99) and calls `session.run([out_spec, out_state], {spec, state})` once per
frame (api.py:98-101, 154-157; stream.py:129-135).  `HipSession` quacks like the `ort.InferenceSession` those callers
use -- `run`, `get_inputs`, `get_outputs`, `get_modelmeta().custom_metadata_map` (the state-initialisation metadata of
export_dpdfnet_to_onnx.py:55-83, so the reference's `load_initial_state_from_metadata` works on it as is) -- and every
`run` is one `dpdf_run_frames(B=1, T=1)` through the C ABI.  A maintainer drops this module in for `onnx_backend`
(INTEGRATION.md section 1); `dpdfnet_amd.api` / `.stream` use the batched entry points instead, which is what the
GPU is for.
"""
from __future__ import annotations

from dataclasses import dataclass
from pathlib import Path
from types import SimpleNamespace
from typing import Dict, List, Optional, Sequence, Union

import numpy as np

from .models import SYNTHETIC_PREFIX, get_model_info

IN_SPEC, IN_STATE, OUT_SPEC, OUT_STATE = "spec", "state_in", "spec_e", "state_out"


@dataclass(frozen=True)
class RuntimeModel:            # field for field the reference's dataclass (onnx_backend.py:11-18)
    session: "HipSession"
    init_state: np.ndarray
    in_spec_name: str
    in_state_name: str
    out_spec_name: str
    out_state_name: str


class HipSession:
    """`ort.InferenceSession` look-alike over a `backend.HipModel` (one GPU frame-function handle)."""

    def __init__(self, hip_model) -> None:
        self._m = hip_model
        self.freq_bins = int(hip_model.freq_bins)
        self.state_size = int(hip_model.state_size)
        self.win_len = int(hip_model.win_len)

    # -- the four InferenceSession members the reference touches ---------------------------------
    def get_inputs(self) -> List[SimpleNamespace]:                     # infer_win_len reads [0].shape[-2]
        return [SimpleNamespace(name=IN_SPEC, shape=[1, 1, self.freq_bins, 2], type="tensor(float)"),
                SimpleNamespace(name=IN_STATE, shape=[self.state_size], type="tensor(float)")]

    def get_outputs(self) -> List[SimpleNamespace]:
        return [SimpleNamespace(name=OUT_SPEC, shape=[1, 1, self.freq_bins, 2], type="tensor(float)"),
                SimpleNamespace(name=OUT_STATE, shape=[self.state_size], type="tensor(float)")]

    def get_providers(self) -> List[str]:
        return ["DpdfnetHipExecutionProvider"]

    def get_modelmeta(self) -> SimpleNamespace:
        """The metadata keys the exporter embeds (export_dpdfnet_to_onnx.py:59-83): sizes and the two norm-state
        init vectors as comma-separated %.9g floats."""
        d = self._m.dims
        init = self._m.initial_state()
        e, s = init[: d.E], init[d.E: d.E + d.D]
        fmt = lambda v: ",".join(f"{float(x):.9g}" for x in v)
        return SimpleNamespace(custom_metadata_map={
            "state_size": str(self.state_size), "erb_norm_state_size": str(int(d.E)), "spec_norm_state_size": str(int(d.D)),
            "erb_norm_init": fmt(e), "spec_norm_init": fmt(s)})

    def run(self, output_names: Optional[Sequence[str]], feeds: Dict[str, np.ndarray]) -> List[np.ndarray]:
        """One frame for one stream: spec [1,1,F,2] + state [S] -> [spec_e [1,1,F,2], state_out [S]]
        (fresh arrays, like ORT; the caller's `state` is not modified)."""
        try:
            spec, state = feeds[IN_SPEC], feeds[IN_STATE]
        except KeyError as exc:
            raise ValueError(f"Missing input {exc} (expected {IN_SPEC!r} and {IN_STATE!r})") from exc
        spec = np.ascontiguousarray(spec, dtype=np.float32)
        if spec.shape != (1, 1, self.freq_bins, 2):
            raise ValueError(f"{IN_SPEC} must have shape (1, 1, {self.freq_bins}, 2), got {spec.shape}")
        out, st = self._m.run_frames(spec, np.asarray(state, dtype=np.float32).reshape(1, -1))
        res = {OUT_SPEC: out.reshape(1, 1, self.freq_bins, 2), OUT_STATE: st.reshape(-1)}
        names = list(output_names) if output_names else [OUT_SPEC, OUT_STATE]
        try:
            return [res[n] for n in names]
        except KeyError as exc:
            raise ValueError(f"Unknown output {exc}") from exc


def load_initial_state_from_metadata(session) -> np.ndarray:
    """Same contract as the reference (onnx_backend.py:52-78), reading the shim's metadata map."""
    if len(session.get_inputs()) < 2:
        raise ValueError("Expected streaming ONNX model with two inputs: (spec, state).")
    meta = session.get_modelmeta().custom_metadata_map
    try:
        n, ne, ns = int(meta["state_size"]), int(meta["erb_norm_state_size"]), int(meta["spec_norm_state_size"])
        e = np.array([float(x) for x in meta["erb_norm_init"].split(",")], dtype=np.float32)
        s = np.array([float(x) for x in meta["spec_norm_init"].split(",")], dtype=np.float32)
    except KeyError as exc:
        raise ValueError(f"ONNX model is missing required metadata key: {exc}. "
                         "Re-export the model to embed state initialisation metadata.") from exc
    init = np.zeros(n, dtype=np.float32)
    init[:ne] = e
    init[ne: ne + ns] = s
    return np.ascontiguousarray(init)


def build_runtime_model(onnx_path: Union[str, Path], model: str = "dpdfnet2", device: int = 0) -> RuntimeModel:
    """Drop-in for the reference's `build_runtime_model(onnx_path)`: `onnx_path` is the weight file (or
    "synthetic:<seed>"), `model` names the architecture (an .onnx file carries it; a weight blob does not)."""
    from . import backend, weights
    info = get_model_info(model)
    entries = backend.manifest(info.sample_rate, info.dprnn_num_blocks)
    extras: Dict[str, np.ndarray] = {}
    if isinstance(onnx_path, str) and onnx_path.startswith(SYNTHETIC_PREFIX):
        blob = weights.synth_blob(entries, int(onnx_path[len(SYNTHETIC_PREFIX):] or 0))
    else:
        blob, extras = weights.load_weight_file(onnx_path, entries)
    hip = backend.HipModel(info.sample_rate, info.dprnn_num_blocks, blob, device=device,
                           erb_norm_init=extras.get("erb_norm_init"), spec_norm_init=extras.get("spec_norm_init"))
    session = HipSession(hip)
    return RuntimeModel(session=session, init_state=load_initial_state_from_metadata(session), in_spec_name=IN_SPEC,
                        in_state_name=IN_STATE, out_spec_name=OUT_SPEC, out_state_name=OUT_STATE)


def infer_win_len(session, default_sr: int) -> int:
    """onnx_backend.py:102-107 verbatim semantics: (F - 1) * 2 from the static input shape."""
    shape = session.get_inputs()[0].shape
    fb = shape[-2] if len(shape) >= 2 else None
    if isinstance(fb, int) and fb > 1:
        return int((fb - 1) * 2)
    return int(round(default_sr * 0.02))
