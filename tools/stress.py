#!/usr/bin/env python3
"""Randomised self-consistency soak: for random (model, streams, frames, chunk length) the default execution shape
(4 streams, per-launch kernel selection, 8-workgroup clusters, ...) must reproduce the plain one (single stream,
whole-sequence chunk, unfused kernels) of the SAME engine to rounding, and repeated runs must be bit-identical.
Catches ordering bugs between streams / cluster workgroups that a fixed test size can miss.  argv: seconds [seed]."""
import sys, time, json
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
            import os
            for kv in filter(None, os.environ.get("DPDF_STRESS_OPTS", "").split(",")):      # bisecting: DPDF_STRESS_OPTS=hop_fused=0,fuse_small=0
                models[(sr, nb)].set_option(kv.split("=")[0], int(kv.split("=")[1]))
        return models[(sr, nb)]

    t_end = time.time() + budget
    n = 0; worst = 0.0; worst_case = None
    while time.time() < t_end or n < min_cases:       # (min_cases: a fixed floor of cases however slow the box)
        sr, nb = [(16000, 2), (16000, 4), (16000, 8), (48000, 2), (48000, 8)][rng.integers(5)]
        m = model(sr, nb)
        B = int(rng.choice([1, 2, 3, 7, 16, 17, 33, 64, 65, 130, 257]))
        T = int(rng.choice([1, 2, 3, 5, 9, 31, 64, 100, 257]))
        if B * T > 20000: T = max(1, 20000 // B)
        F = m.freq_bins
        spec = (rng.standard_normal((B, T, F, 2)) * 3.0).astype(np.float32)
        st0 = np.tile(m.initial_state()[None, :], (B, 1))
        m.set_overlap(0); m.set_fuse_dprnn("never"); m.set_chunk_frames(-1); m.set_option("gru256_fused_x", 0); m.set_option("dec_seg", 0)
        ref, sref = m.run_frames(spec, st0.copy())
        m.set_overlap(27); m.set_fuse_dprnn("auto"); m.set_chunk_frames(int(rng.choice([0, 0, 1, 2, 7, 64]))); m.set_option("gru256_fused_x", 1); m.set_option("dec_seg", 2)
        out, s1 = m.run_frames(spec, st0.copy())
        out2, s2 = m.run_frames(spec, st0.copy())
        scale = float(np.abs(ref).max()) + 1e-12
        err = float(np.abs(out - ref).max()) / scale
        if err > worst: worst = err; worst_case = {"sr": sr, "nb": nb, "B": B, "T": T, "err": err}
        same = np.array_equal(out, out2) and np.array_equal(s1, s2)
        if not np.isfinite(out).all() or err > 5e-5 or not same or np.abs(s1 - sref).max() > 5e-4 * (1 + np.abs(sref).max()):
            rec = {"FAIL": True, "sr": sr, "nb": nb, "B": B, "T": T, "err": err, "repeatable": bool(same),
               "state_err": float(np.abs(s1 - sref).max())}
            return rec
        n += 1
    for m in models.values():
        m.close()
    return {"cases": n, "worst_rel_err": worst, "worst_case": worst_case, "seconds": budget}


if __name__ == "__main__":
    rec = run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print(json.dumps(rec))
    sys.exit(1 if rec.get("FAIL") else 0)
