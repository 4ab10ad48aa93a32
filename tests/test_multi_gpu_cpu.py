"""CPU: the N>1 sharding path over gloo, world_size 2 (RCCL on the GPU box uses the same code)."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


def test_shard_range_partitions_exactly():
    from dpdfnet_amd.multi_gpu import shard_range, shard_sizes
    for n in (0, 1, 7, 256, 2048, 2049):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(shard_sizes(n, w)) - min(shard_sizes(n, w)) <= 1
    assert shard_range(2048, 8, 3) == (768, 1024)
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def _worker(rank: int, world: int, port: int, q):
    sys.path.insert(0, str(ROOT))
    import torch
    import torch.distributed as dist
    from dpdfnet_amd.multi_gpu import gather_ragged_to_root, gather_to_root, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        total, n = 5, 16
        full = np.arange(total * n, dtype=np.float32).reshape(total, n)
        lo, hi = shard_range(total, world, rank)
        local = full[lo:hi] * 2.0                         # "enhance" = x2 on this rank's clips
        got = gather_ragged_to_root(local, [shard_range(total, world, r)[1] - shard_range(total, world, r)[0] for r in range(world)], rank)
        eq = torch.full((3, n), float(rank))
        g2 = gather_to_root(eq, world, rank)
        dist.barrier()
        if rank == 0:
            q.put((np.array_equal(got, full * 2.0), [float(g2[r].mean()) for r in range(world)]))
    finally:
        dist.destroy_process_group()


def test_gather_world_size_2_gloo():
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok, means = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok and means == [0.0, 1.0]


def test_bench_plain_invocation_becomes_its_own_launcher(monkeypatch):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment re-launches itself under torch.distributed.run with N ranks
    on 127.0.0.1 and passes its own arguments through (bench.self_launch); under a launcher (WORLD_SIZE set) it does not."""
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    sys.path.insert(0, str(root))
    import bench
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return subprocess.CompletedProcess(cmd, 0)

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2", "--backend", "gloo"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert bench.self_launch(4) == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-6:] == ["--gpus", "4", "--steps", "2", "--backend", "gloo"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["MASTER_ADDR"] == "127.0.0.1" and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # main(): plain invocation with --gpus 4 -> SystemExit(rc of the launcher); with WORLD_SIZE set it goes on to the rank path
    with pytest.raises(SystemExit) as ei:
        bench.main()
    assert ei.value.code == 0 and seen["cmd"][1:3] == ["-m", "torch.distributed.run"]


def test_bench_names_the_banked_trace_of_its_own_mode():
    """bench.banked_trace(kind) is the file the roofline block tells a reader to compare with: the newest profiles/r<N>_<kind>_kernel_stats.csv,
    by EXACT name -- the A/B mode's `r6_fp32mfma_pipelined_kernel_stats.csv` is not the default engine's pipelined trace."""
    import re
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    sys.path.insert(0, str(root))
    import bench
    for kind in ("pipelined", "serial", "fp32mfma_pipelined", "fp32mfma_serial"):
        name = bench.banked_trace(kind)
        assert re.fullmatch(rf"profiles/r\d+_{kind}_kernel_stats\.csv", name), (kind, name)
        assert (root / name).is_file()
