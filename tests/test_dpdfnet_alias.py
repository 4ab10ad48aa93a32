"""`import dpdfnet` drop-in (reference package/src/dpdfnet/__init__.py:3-9; the submodules its tests reach into:
package/tests/test_package_behaviors.py:95-107).  CPU part: the alias resolves to the engine package's own module objects, and
the whole host-behaviour suite passes when it is written against the reference's import name.  GPU part: `dpdfnet.enhance()` and
`dpdfnet.StreamEnhancer` run on the HIP engine and agree with the oracle."""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


def test_alias_modules_are_the_engine_modules():
    import dpdfnet
    import dpdfnet_amd
    import dpdfnet.audio as audio_mod
    import dpdfnet.onnx_backend as backend_mod
    from dpdfnet import api, models, stream
    import dpdfnet_amd.api, dpdfnet_amd.audio, dpdfnet_amd.models, dpdfnet_amd.ort_shim, dpdfnet_amd.stream
    assert audio_mod is dpdfnet_amd.audio and backend_mod is dpdfnet_amd.ort_shim
    assert api is dpdfnet_amd.api and models is dpdfnet_amd.models and stream is dpdfnet_amd.stream
    # the reference's import surface (test_package_behaviors.py:17-24)
    for name in ("enhance", "enhance_file", "available_models", "download", "StreamEnhancer"):
        assert name in dpdfnet.__all__ and getattr(dpdfnet, name) is getattr(dpdfnet_amd, name)
    # the names the reference's tests monkeypatch exist under the reference's module names
    for name in ("to_mono", "ensure_sample_rate", "make_stft_config", "preprocess_waveform", "postprocess_spec", "fit_length"):
        assert callable(getattr(audio_mod, name)), name
    for name in ("build_runtime_model", "infer_win_len"):
        assert callable(getattr(backend_mod, name)), name
    assert callable(api.resolve_model)
    with pytest.raises(AttributeError, match="module 'dpdfnet' has no attribute"):
        dpdfnet.nope


def test_host_behaviour_suite_passes_under_the_reference_import_name(tmp_path):
    """tests/test_host_behaviors.py, every `dpdfnet_amd` spelled `dpdfnet`, run by pytest in a child process."""
    src = (ROOT / "tests" / "test_host_behaviors.py").read_text()
    assert "dpdfnet_amd" in src
    (tmp_path / "test_host_behaviors_as_dpdfnet.py").write_text(src.replace("dpdfnet_amd", "dpdfnet"))
    env = dict(os.environ, PYTHONPATH=str(ROOT) + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", "--rootdir", str(tmp_path),
                        str(tmp_path / "test_host_behaviors_as_dpdfnet.py")], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout


@pytest.mark.gpu
def test_import_dpdfnet_enhance_and_stream_run_on_the_engine():
    import dpdfnet
    from dpdfnet_amd import backend
    from dpdfnet_amd.weights import synth_blob
    from oracle import oracle as orc
    sr, nb, model = 16000, 2, "dpdfnet2"
    rng = np.random.default_rng(3)
    n = 12000
    x = (0.05 * rng.standard_normal(n) + 0.1 * np.sin(2 * np.pi * 330.0 * np.arange(n) / sr)).astype(np.float32)
    y = dpdfnet.enhance(x, sr, model=model, onnx_path="synthetic:1234")
    se = dpdfnet.StreamEnhancer(model=model, onnx_path="synthetic:1234")
    parts = [se.process(x[i:i + 1000], sr) for i in range(0, n, 1000)] + [se.flush()]
    ys = np.concatenate(parts)
    ref = orc.Oracle(sr, nb, synth_blob(backend.manifest(sr, nb), 1234)).enhance(x)
    assert y.shape == x.shape and np.sqrt(np.mean((y - ref) ** 2)) < 2e-6
    assert backend.load_library() is not None          # the product .so is what ran
    assert ys.shape[0] >= n - 2 * 320 and np.isfinite(ys).all()
