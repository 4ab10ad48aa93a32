"""GPU (-m gpu): SHORT randomised self-consistency soaks (4 s each here; the 120 s runs of the same tools are banked per round under profiles/r*_soak.txt).
tools/stress.py: random model / stream count / frame count /
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
    rec = mod.run(budget=4.0, seed=20260417, min_cases=20)               # (a floor of cases, not a speed test: a slow box takes longer)
    assert not rec.get("FAIL"), rec
    assert rec["cases"] >= 20 and rec["worst_rel_err"] < 5e-5, rec


def test_random_streaming_call_sequences_hop_forms_equal_plain_chain():
    """tools/stream_soak.py: the same random sequence of streaming calls (single hops, multi-hop calls, masked calls, resets,
    state save / restore through the host; 1..70 streams, four models) through an engine with every hop-only form on and one
    with all of them off: every output and the final states must agree."""
    spec = importlib.util.spec_from_file_location("dpdf_stream_soak", Path(__file__).resolve().parents[1] / "tools" / "stream_soak.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rec = mod.run(4.0, 20260929, min_cases=10)
    assert not rec.get("FAIL"), rec
    assert rec["cases"] >= 10 and rec["worst_rms"] < 2e-6, rec


def test_random_host_calls_pipelined_equal_plain():
    """tools/host_pipe_soak.py: random batch shapes / chunk lengths / ragged lengths / attenuation limits through the pipelined
    host-pointer calls and through the plain ones of the same engine: bit-identical, in the block, ragged and row-pointer forms."""
    spec = importlib.util.spec_from_file_location("dpdf_host_pipe_soak", Path(__file__).resolve().parents[1] / "tools" / "host_pipe_soak.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rec = mod.run(4.0, 20260929, min_cases=10)
    assert not rec.get("FAIL"), rec
    assert rec["cases"] >= 10 and rec["cases_with_pipelined_shape"] >= 1, rec


def test_two_engines_hop_side_by_side_without_a_timeout():
    """Two engine handles (two host threads, 64 x 48 kHz dpdfnet8 streams and 40 x 16 kHz dpdfnet4 streams) hop at the same time:
    their one-launch DPRNN blocks (dprnn_hop_block.h: glue tiles wait for the scans of their OWN launch) and GRU-256 step kernels share
    the chip.  Forward progress must not depend on having it alone: every hop equals the same engine run by itself, bit for bit, and
    no call went through the time-out / recovery path."""
    import threading
    import numpy as np
    from dpdfnet_amd import backend as be
    from dpdfnet_amd.weights import synth_blob

    def session(sr, nb, S, hops, out, barrier=None):
        m = be.HipModel(sr, nb, synth_blob(be.manifest(sr, nb), 20260417), 0)
        st = be.HipStreams(m, S)
        r = np.random.default_rng(sr + nb)
        st.prime((0.05 * r.standard_normal((S, m.hop))).astype(np.float32))
        if barrier is not None:
            barrier.wait()
        res = [st.process((0.05 * r.standard_normal((S, m.hop))).astype(np.float32)).copy() for _ in range(hops)]
        out.append((res, int(m.recovery_count)))
        st.close(); m.close()

    cfgs = [(48000, 8, 64, 150), (16000, 4, 40, 300)]
    alone = []
    for c in cfgs:
        o = []; session(*c, o); alone.append(o[0])
    bar = threading.Barrier(2)
    outs = [[], []]
    ths = [threading.Thread(target=session, args=(*c, outs[i], bar)) for i, c in enumerate(cfgs)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    for i in range(2):
        res, rec = outs[i][0]
        assert rec == 0 and alone[i][1] == 0, (i, rec)
        for a_, b_ in zip(res, alone[i][0]):
            np.testing.assert_array_equal(a_, b_)


def test_hops_beside_torch_streams_in_the_same_process_never_time_out():
    """tools/torch_coexist_soak.py in a process of its own with GPU_MAX_HW_QUEUES=4 (HIP's default, not the package's 8): a second host
    thread keeps two torch streams busy (GEMMs + elementwise kernels) while single-hop streaming calls run; the hop's cross-stream
    counter join (hop_spin_join) shares hardware queues with them.  Zero recoveries, every hop bit-identical to the run alone."""
    import json, subprocess, sys
    root = Path(__file__).resolve().parents[1]
    for args in (["1500", "16000", "2", "1", "4"], ["600", "48000", "8", "64", "4"]):      # (the long runs: profiles/r6_torch_coexist_soak.txt, 100 000 hops)
        r = subprocess.run([sys.executable, str(root / "tools" / "torch_coexist_soak.py")] + args, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-1500:]
        rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert not rec["FAIL"] and rec["recoveries_beside_torch"] == 0 and rec["hops_differing_from_the_run_alone"] == 0, rec
        assert rec["torch_kernels_launched"] > 100, rec
