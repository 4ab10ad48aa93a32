"""N3 -- sample-rate conversion.  CPU: the oracle restatement against scipy.signal.resample_poly (the published
algorithm it follows; the reference's own librosa/soxr path is absent => parity with it is unpinned).  GPU: the HIP
polyphase kernel against the oracle through the C ABI, and `enhance()` at a foreign rate end to end."""
from fractions import Fraction

import numpy as np
import pytest

from oracle.oracle import Oracle

RATIOS = [(48000, 16000), (16000, 48000), (44100, 16000), (16000, 44100), (8000, 16000), (22050, 48000), (32000, 48000)]


def _sig(n, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(n, dtype=np.float64)
    return (0.05 * rng.standard_normal(n) + 0.2 * np.sin(0.01 * t)).astype(np.float32)


@pytest.mark.parametrize("sr_in,sr_out", RATIOS)
@pytest.mark.parametrize("n", [1, 7, 480, 4411, 16000])
def test_oracle_resample_matches_scipy(sr_in, sr_out, n):
    from scipy.signal import resample_poly
    x = _sig(n, n + sr_in)
    fr = Fraction(sr_out, sr_in)
    ref = resample_poly(x.astype(np.float64), fr.numerator, fr.denominator)
    got = Oracle.resample(x, sr_in, sr_out)
    assert got.shape == ref.shape == (-(-n * sr_out // sr_in),)
    np.testing.assert_allclose(got, ref, atol=3e-8, rtol=0)      # float32 rounding of the float64 sum


def test_oracle_resample_identity_and_dc_gain():
    x = _sig(1000, 1)
    assert Oracle.resample(x, 16000, 16000) is not None
    np.testing.assert_array_equal(Oracle.resample(x, 16000, 16000), x)
    dc = np.full(4800, 0.25, np.float32)
    y = Oracle.resample(dc, 48000, 16000)
    np.testing.assert_allclose(y[100:-100], 0.25, atol=2e-4)      # Kaiser(5) passband ripple; edges see the zero padding


def test_resample_len_is_host_arithmetic():
    from dpdfnet_amd import backend
    L = backend.load_library()
    assert L.dpdf_resample_len(48000, 48000, 16000) == 16000
    assert L.dpdf_resample_len(7, 48000, 16000) == 3
    assert L.dpdf_resample_len(777, 16000, 44100) == 2142
    assert L.dpdf_resample_len(0, 16000, 44100) == 0
    assert L.dpdf_resample_len(-1, 16000, 44100) == -1 and L.dpdf_resample_len(5, 0, 44100) == -1


@pytest.mark.gpu
@pytest.mark.parametrize("sr_in,sr_out", RATIOS)
def test_hip_resample_matches_oracle(sr_in, sr_out):
    from dpdfnet_amd import backend
    for n in (1, 7, 480, 4411, 48000):
        x = _sig(n, 3 * n + sr_out)
        got = backend.resample(x, sr_in, sr_out)
        ref = Oracle.resample(x, sr_in, sr_out)
        assert got.shape == ref.shape
        np.testing.assert_allclose(got, ref, atol=2e-6, rtol=0)   # fp32 taps + fp32 FMA chain of <= 62 terms, |x| < 0.5


@pytest.mark.gpu
def test_hip_resample_batch_rows_are_independent_and_deterministic():
    from dpdfnet_amd import backend
    x = np.stack([_sig(44100, s) for s in (1, 2, 1, 3)])
    y = backend.resample(x, 44100, 16000)
    assert y.shape == (4, 16000)
    np.testing.assert_array_equal(y[0], y[2])
    np.testing.assert_array_equal(y, backend.resample(x, 44100, 16000))
    np.testing.assert_array_equal(y[1], backend.resample(x[1], 44100, 16000))
    with pytest.raises(ValueError):
        backend.resample(x, 0, 16000)
    assert backend.resample(np.zeros((2, 0), np.float32), 48000, 16000).shape == (2, 0)


@pytest.mark.gpu
def test_enhance_at_a_foreign_rate_equals_oracle_pipeline():
    """enhance(audio, 48000) on a 16 kHz model: resample -> frame function -> resample back -> fit_length
    (reference api.py:51-113), every DSP stage on the device; oracle = the same chain on the CPU."""
    import dpdfnet_amd as pkg
    from dpdfnet_amd import backend
    from dpdfnet_amd.weights import synth_blob
    sr_in, msr, nb, seed = 48000, 16000, 2, 20260417
    x = _sig(24000, 11)
    out = pkg.enhance(x, sr_in, model=f"dpdfnet{nb}", onnx_path=f"synthetic:{seed}")
    blob = synth_blob(backend.manifest(msr, nb), seed)
    o = Oracle(msr, nb, blob)
    ref = Oracle.resample(o.enhance(Oracle.resample(x, sr_in, msr)), msr, sr_in)
    ref = ref[: x.size] if ref.size >= x.size else np.pad(ref, (0, x.size - ref.size))
    assert out.shape == x.shape and out.dtype == np.float32
    rms = float(np.sqrt(np.mean((out - ref) ** 2)))
    assert rms < 1e-4, rms                                        # north-star waveform budget; measured ~1e-7
