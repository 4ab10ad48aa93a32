"""GPU (-m gpu): a short randomised self-consistency soak (tools/stress.py): random model / stream count / frame count /
chunk length; the default execution shape (four streams, per-launch kernel selection, GRU-256 clusters of 4 or 8
workgroups, hoisted input GEMMs, fused epilogues) against the plain single-stream unfused form of the same engine,
plus run-to-run bit identity.  The fixed-size parity tests pin the numbers to the oracle; this one hunts ordering bugs
between streams and cluster workgroups at sizes nobody thought of."""
import importlib.util
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu


def test_random_shapes_default_pipeline_equals_plain_form():
    spec = importlib.util.spec_from_file_location("dpdf_stress", Path(__file__).resolve().parents[1] / "tools" / "stress.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rec = mod.run(budget=10.0, seed=20260417)
    assert not rec.get("FAIL"), rec
    assert rec["cases"] >= 5 and rec["worst_rel_err"] < 5e-5, rec          # (a count, not a speed test: a slow box must not fail parity)


def test_random_streaming_call_sequences_hop_forms_equal_plain_chain():
    """tools/stream_soak.py: the same random sequence of streaming calls (single hops, multi-hop calls, masked calls, resets,
    state save / restore through the host; 1..70 streams, four models) through an engine with every hop-only form on and one
    with all of them off: every output and the final states must agree."""
    spec = importlib.util.spec_from_file_location("dpdf_stream_soak", Path(__file__).resolve().parents[1] / "tools" / "stream_soak.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rec = mod.run(8.0, 20260929)
    assert not rec.get("FAIL"), rec
    assert rec["cases"] >= 3 and rec["worst_rms"] < 2e-6, rec


def test_random_host_calls_pipelined_equal_plain():
    """tools/host_pipe_soak.py: random batch shapes / chunk lengths / ragged lengths / attenuation limits through the pipelined
    host-pointer calls and through the plain ones of the same engine: bit-identical, in the block, ragged and row-pointer forms."""
    spec = importlib.util.spec_from_file_location("dpdf_host_pipe_soak", Path(__file__).resolve().parents[1] / "tools" / "host_pipe_soak.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rec = mod.run(10.0, 20260929)
    assert not rec.get("FAIL"), rec
    assert rec["cases"] >= 3 and rec["cases_with_pipelined_shape"] >= 1, rec
