#!/usr/bin/env python3
"""Randomised soak of the host-pointer batch calls (csrc/dpdf_model.hip enhance_impl): for random (model, clips, length, chunk
length, ragged lengths, attenuation limit, copy threads) the pipelined call -- per-chunk STFT / iSTFT, pinned staging ring, copy
threads, two-stage DFT -- must be BIT-identical to the plain call of the same engine (one upload, whole-batch transforms, one
download), in the block, ragged-block and row-pointer forms, and every clip's zero tail must be exact.  argv: seconds [seed]."""
import ctypes, json, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
from dpdfnet_amd import backend as be
from dpdfnet_amd.weights import synth_blob


def run(budget: float = 60.0, seed: int = 1, min_cases: int = 0) -> dict:
    rng = np.random.default_rng(seed)
    models = {}

    def model(sr, nb):
        if (sr, nb) not in models:
            models[(sr, nb)] = be.HipModel(sr, nb, synth_blob(be.manifest(sr, nb), 20260417), 0)
        return models[(sr, nb)]

    t_end = time.time() + budget
    n = piped_cases = 0
    while time.time() < t_end or n < min_cases:       # (min_cases: a fixed floor of cases however slow the box)
        sr, nb = [(16000, 2), (16000, 4), (48000, 2), (16000, 0)][rng.integers(4)]
        m = model(sr, nb)
        hop = m.hop
        B = int(rng.choice([1, 3, 8, 17, 40, 64, 97, 130]))
        T_target = int(rng.choice([3, 6, 9, 17, 40, 70, 130]))
        N = max(1, T_target * hop + int(rng.integers(-hop + 1, hop)))
        if B * (N // hop + 3) > 9000:
            B = max(1, 9000 // (N // hop + 3))
        chunk = int(rng.choice([0, 1, 2, 4, 5, 6, 8, 16, 33, 64]))
        attn = [None, 6.0, 0.0][rng.integers(3)]
        wav = (0.1 * rng.standard_normal((B, N))).astype(np.float32)
        lens = np.minimum(N, np.maximum(0, (N * rng.uniform(0.0, 1.2, B)).astype(np.int32))).astype(np.int32)
        lens[rng.integers(B)] = N
        db = float("nan") if attn is None else attn
        outs = {}
        for pipe in (1, 0):
            m.set_option("host_pipe", pipe)
            m.set_option("host_copy_threads", int(rng.choice([1, 2, 4])))
            m.set_chunk_frames(chunk)
            blk = m.enhance_batch(wav, attn)
            rag = np.full_like(wav, np.nan)
            be._check(m._L.dpdf_enhance_batch_ragged(m._h, wav.ctypes.data, B, N, lens.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), db, rag.ctypes.data, 0))
            rows = m.enhance_batch_ragged([wav[b, : lens[b]].copy() for b in range(B)], attn)
            outs[pipe] = (np.array(blk), rag, [np.array(r) for r in rows])
        case = {"sr": sr, "nb": nb, "B": B, "N": N, "chunk": chunk, "attn": attn}
        a, b_ = outs[1], outs[0]
        why = []
        if not np.array_equal(a[0], b_[0]): why.append("block: pipelined != plain")
        if not np.array_equal(a[1], b_[1]): why.append("ragged block: pipelined != plain")
        if not all(np.array_equal(x, y) for x, y in zip(a[2], b_[2])): why.append("rows: pipelined != plain")
        if not (np.isfinite(a[0]).all() and np.isfinite(a[1]).all()): why.append("not finite")
        same_batch = bool((lens > 0).all())           # (zero-length clips stay out of the row-pointer call: another batch size, other kernel forms)
        for b in range(B):
            d = np.abs(a[2][b] - a[1][b, : lens[b]]).max() if lens[b] else 0.0
            if (same_batch and d != 0.0) or d > 2e-5: why.append(f"rows vs ragged block, clip {b}: {d}")
            if a[1][b, lens[b]:].any(): why.append(f"ragged block: out[{b}][len:] not zero")
            if a[0][b, max(0, N - 2 * hop):].any(): why.append(f"zero tail of clip {b}")      # the reference's zero tail
        if why:
            return {"FAIL": True, **case, "why": why[:6]}
        n += 1
        piped_cases += int(B * (chunk if chunk > 0 else min(N // hop + 3, 256)) > 512)
    for m in models.values():
        m.close()
    return {"cases": n, "cases_with_pipelined_shape": piped_cases, "seconds": budget}


if __name__ == "__main__":
    rec = run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print(json.dumps(rec))
    sys.exit(1 if rec.get("FAIL") else 0)
