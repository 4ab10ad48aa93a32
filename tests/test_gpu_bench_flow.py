"""GPU (-m gpu): bench.py's contract, exercised for real: the N=1 line and -- with two ranks sharing this box's GPU over
gloo -- the N>1 control flow (rank env, sharding, barriers, max-over-ranks timing, gather to rank 0, one JSON line from
rank 0 only).  RCCL itself needs >= 2 GPUs and is what the driver's scaling run exercises."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline")


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _json_lines(stdout: str):
    return [json.loads(l) for l in stdout.splitlines() if l.startswith('{"metric"')]


def test_single_gpu_line_has_the_contract_fields():
    r = subprocess.run([sys.executable, "bench.py", "--clips", "16", "--steps", "1", "--warmup", "1", "--no-other-configs",
                        "--cpu-clip-seconds", "0.5", "--cpu-clips-per-thread", "1"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1
    d = lines[0]
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["unit"] == "frames/s" and d["higher_is_better"] is True and d["finite_output"] is True
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "mfma" and 0.0 < rf["frac"] < 1.0
    import re
    assert re.search(r"profiles/r\d+_pipelined_kernel_stats\.csv", rf["reproduce"])
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0
    assert cb["reference_runtime"]["kind"] == "ort"            # the optional onnxruntime leg: measured, or reported as absent with the reason
    # the timed step's own output against the oracle, on the line (16 clips here: slots 0 / 8 / 15)
    assert d["parity"]["ok"] is True and d["parity"]["rms_max"] < d["parity"]["tol"] and len(d["parity"]["clips"]) == 3
    assert d["value_hbm_resident"] == d["value"]
    # the H2D/D2H-inclusive figure rides on the same line and can only be slower than the HBM-resident one
    assert 0 < d["value_incl_pcie"] <= d["value"] * 1.05 and d["pcie"]["finite_output"] is True
    # ... measured through the library's own pipelined host-pointer call, bit-identical to the HBM-resident output; and the figure of
    # the public Python API (list of numpy clips in, list out) rides on the same line
    assert d["pcie"]["max_abs_diff_vs_hbm_resident_output"] == 0.0 and d["pcie"]["ms_per_step_unpipelined"] > 0
    assert d["value_public_api"] > 0 and d["public_api"]["finite_output"] is True
    # (with DPDF_GRU64_LIMBS in the environment the public API's own handle runs the opt-in limb kernels while the headline stays fp32 MFMA:
    # equal to rounding then, bit-identical otherwise)
    import os
    assert d["public_api"]["max_abs_diff_vs_hbm_resident_output"] <= (5e-6 if os.environ.get("DPDF_GRU64_LIMBS", "3") not in ("", "3") else 0.0)
    # RCCL has executed on this box: one-rank process group (backend nccl) in a child process, the collectives of the N > 1 path
    st = d["rccl_selftest"]
    # the child streams stage stamps and runs under a 60 s cap: a run that did not finish must SAY where it stopped (round-4 review:
    # a time-out may not simply pass)
    if st.get("rccl_init_ok") is False:
        assert st.get("hung_at") and isinstance(st.get("stages"), list), st
        pytest.fail(f"RCCL self-test did not finish on this box: hung at {st['hung_at']!r}; stages {st['stages']}")
    assert d["rccl_init_ok"] is True, st
    assert st["all_reduce_ok"] and st["barrier_ok"] and st["all_gather_object_ok"], st
    # the headline is the default engine (GRU-64 throughput kernels on bf16 limbs since round 6) and says so in `dtype`, with the float64-recurrence
    # error of both kernel families beside it; the fp32-MFMA kernels ride along as a block with its own parity (16 clips: the fused launches are not
    # selected at this size, so the block reports the timing and the parity only)
    assert d["dtype"].startswith("f32") and "bf16x3 limbs" in d["dtype"] and "ab_mode" not in d
    fe = d["float64_recurrence_error"]
    assert not fe["available"] or fe["limb_kernels"]["rms"] <= 1.2 * fe["fp32_mfma_kernels"]["rms"], fe       # not narrower than the fp32 MFMA (measured: 2.9e-8 vs 3.5e-8)
    lk = d["fp32_mfma_kernels"]
    assert lk["value"] > 0 and lk["parity"]["ok"] is True and "NOT the headline" in lk["what"] and lk["dtype"] == "f32"


def test_two_ranks_control_flow_over_gloo():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--backend", "gloo",
                        "--clips", "8", "--no-other-configs"], cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1                                   # rank 0 only
    d = lines[0]
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["finite_output"] is True
    assert "gather to rank 0" in d["config"]["sharding"] and "cpu_baseline" not in d
    T = d["config"]["frames_per_clip"]
    assert abs(d["value"] - 2 * 8 * T * d["steps"] / (d["ms_per_step"] * d["steps"] / 1e3)) < 1e-6 * d["value"]
    mg = d["multi_gpu"]                                      # the pre-flight block the 8-GPU run reports
    assert [r["rank"] for r in mg["rccl_ranks"]] == [0, 1] and len(mg["per_rank"]) == 2
    assert mg["collective_ok"] is True and mg["collective_error"] is None and mg["gathered_matches_rank_outputs"] is True
    assert mg["gathered_shape"][:2] == [2, 8]
    assert max(r["ms_per_step"] for r in mg["per_rank"]) <= d["ms_per_step"] * 1.001


def test_plain_invocation_with_gpus_2_launches_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher and no WORLD_SIZE in the environment (the shape of the driver's N=1 command
    with another N): the file starts its ranks itself and rank 0 prints the one line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--backend", "gloo", "--clips", "8",
                        "--no-other-configs"], cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1
    d = lines[0]
    assert d["n_gpus"] == 2 and d["finite_output"] is True
    mg = d["multi_gpu"]
    assert [q["rank"] for q in mg["rccl_ranks"]] == [0, 1] and mg["collective_ok"] is True and mg["gathered_matches_rank_outputs"] is True
    assert "starting 2 ranks" in r.stderr


def test_rccl_that_never_answers_falls_back_to_gloo_and_the_line_says_so():
    """Round-4 review item 2: at N > 1 `init_process_group("nccl")` runs under a watchdog; when it does not return (test hook
    DPDF_BENCH_FAKE_RCCL_HANG=1: the init sleeps instead) every rank re-executes itself on gloo with a host-staged gather -- rc 0, ONE line,
    the compute figure kept, `collective: FALLBACK gloo: <reason naming the stage>`."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update({"DPDF_BENCH_FAKE_RCCL_HANG": "1", "DPDF_BENCH_RCCL_WATCHDOG_S": "3"})
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--clips", "8", "--no-other-configs"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stderr[-2000:]
    d = lines[0]
    assert d["n_gpus"] == 2 and d["finite_output"] is True and d["value"] > 0
    assert "FALLBACK gloo" in d["config"]["sharding"] and "init_process_group" in d["config"]["sharding"]
    mg = d["multi_gpu"]
    assert "FALLBACK" in mg["backend"] and mg["fallback_reason"] and mg["gathered_matches_rank_outputs"] is True


def test_one_rank_rccl_self_test_that_hangs_names_its_stage_and_costs_only_its_cap():
    """N = 1: the RCCL self-test child "hangs" in its communicator init (DPDF_BENCH_FAKE_RCCL_HANG=1), the parent kills it at the cap
    (6 s here, 60 s by default) and the line carries `rccl_init_ok: false`, `hung_at` naming that stage and the stamps that did
    arrive -- the headline is measured all the same (round-4 review item 2a)."""
    env = dict(os.environ, DPDF_BENCH_FAKE_RCCL_HANG="1", DPDF_BENCH_SELFTEST_CAP_S="6")
    r = subprocess.run([sys.executable, "bench.py", "--clips", "16", "--steps", "1", "--warmup", "1", "--no-other-configs", "--no-cpu-baseline",
                        "--no-pcie", "--no-isolated"], cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_lines(r.stdout)[0]
    st = d["rccl_selftest"]
    assert d["rccl_init_ok"] is False and "timed out after 6 s" in st["error"], st
    assert "init_process_group" in st["hung_at"] and any("import torch" in x for x in st["stages"]), st
    assert d["value"] > 0 and d["parity"]["ok"] is True and 5.5 < st["wall_s"] < 30.0
