"""Utterance sharding across the GPUs of one node.

The reference's only parallelism is one file per host thread with a private session
(reference package/src/dpdfnet/cli.py:249-259, 308-311).  Here the shard unit is the clip: each rank
(one process per GPU) owns a contiguous block of clips with its own model replica and state; no
collective touches the data path.  The single exchange step is the final gather of enhanced PCM
to rank 0 over RCCL/xGMI (backend "nccl" on ROCm); on CPU tests the same code runs over gloo.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np


def shard_range(n_items: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of items for `rank`; earlier ranks take the remainder."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError(f"bad rank/world_size {rank}/{world_size}")
    base, rem = divmod(int(n_items), world_size)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def shard_sizes(n_items: int, world_size: int) -> List[int]:
    return [shard_range(n_items, world_size, r)[1] - shard_range(n_items, world_size, r)[0] for r in range(world_size)]


def gather_to_root(local, world_size: int, rank: int, out=None):
    """Gather equally-shaped per-rank tensors [B_local, N] to rank 0 -> [world, B_local, N] (None elsewhere).

    Implemented as point-to-point sends into the root's buffer (root posts world-1 receives): over
    xGMI the 7 peers use the root's 7 direct links concurrently -- no ring, no all-reduce."""
    import torch
    import torch.distributed as dist

    if world_size == 1:
        return local.unsqueeze(0)
    if local.is_cuda and dist.get_backend() == "gloo":      # gloo moves host memory only: stage through pinned host memory
        host = torch.empty(tuple(local.shape), dtype=local.dtype, pin_memory=True)
        host.copy_(local, non_blocking=True)
        torch.cuda.current_stream(local.device).synchronize()
        got = gather_to_root(host, world_size, rank, None)
        if got is None:
            return None
        if out is None:
            return got.to(local.device)
        out.copy_(got)
        return out
    # one grouped launch (ncclGroupStart/End under batch_isend_irecv): the root's world-1 receives progress
    # concurrently instead of one peer after the other
    if rank == 0:
        if out is None:
            out = torch.empty((world_size,) + tuple(local.shape), dtype=local.dtype, device=local.device)
        out[0].copy_(local)
        ops = [dist.P2POp(dist.irecv, out[r], r) for r in range(1, world_size)]
        for q in dist.batch_isend_irecv(ops):
            q.wait()
        return out
    for q in dist.batch_isend_irecv([dist.P2POp(dist.isend, local.contiguous(), 0)]):
        q.wait()
    return None


def gather_ragged_to_root(local_rows: np.ndarray, counts: Sequence[int], rank: int):
    """CPU/NumPy variant for uneven shards (host-side `enhance_batch` across ranks): returns the
    concatenated [sum(counts), N] array on rank 0."""
    import torch
    import torch.distributed as dist

    world = len(counts)
    t = torch.from_numpy(np.ascontiguousarray(local_rows))
    if world == 1:
        return local_rows
    if rank == 0:
        parts = [t]
        for r in range(1, world):
            buf = torch.empty((counts[r],) + tuple(t.shape[1:]), dtype=t.dtype)
            dist.recv(buf, src=r)
            parts.append(buf)
        return torch.cat(parts, dim=0).numpy()
    dist.send(t, dst=0)
    return None
