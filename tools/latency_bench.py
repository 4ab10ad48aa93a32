#!/usr/bin/env python3
"""Small-batch latency of the offline path: `enhance()`-shaped calls with B clips of 10 s (B=1 is what a
single `dpdfnet.enhance(audio)` call costs).  Prints ms/call and the per-class serial breakdown."""
import sys, time, json
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
import torch
from dpdfnet_amd import backend as be
from dpdfnet_amd.weights import synth_blob

def run(sr, nb, B, seconds=10.0, reps=5, overlap=None, breakdown=True, chunk=None):
    m = be.HipModel(sr, nb, synth_blob(be.manifest(sr, nb), 20260417), 0)
    if overlap is not None: m.set_overlap(overlap)
    if chunk is not None: m.set_chunk_frames(chunk)
    N = int(seconds * sr)
    rng = np.random.default_rng(1)
    wav = torch.from_numpy((0.05 * rng.standard_normal((B, N))).astype(np.float32)).cuda()
    out = torch.empty_like(wav)
    for _ in range(2):
        m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None)
    m.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None)
    m.sync()
    ms = (time.perf_counter() - t0) / reps * 1e3
    T = m.num_frames(N)
    rec = {"chunk": chunk, "sr": sr, "nb": nb, "clips": B, "seconds": seconds, "overlap": overlap, "ms_per_call": round(ms, 2),
           "frames_per_s": round(B * T / ms * 1e3), "rtf": round(ms / 1e3 / seconds / B, 6)}
    if breakdown:
        m.set_overlap(0)
        m.profile(True)
        m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None)
        m.sync()
        prof = m.profile_report()
        m.profile(False)
        cls = {}
        for name, (ms_, n) in prof.items():
            cls[name] = round(ms_, 2)
        rec["serial_ms"] = dict(sorted(cls.items(), key=lambda kv: -kv[1])[:12])
    print(json.dumps(rec))

if __name__ == "__main__":
    if "--chunks" in sys.argv:
        for B in (1, 8, 32):
            for ch in (None, 512, 256, 128, 64):
                run(16000, 4, B, breakdown=False, chunk=ch)
        sys.exit(0)
    for B in (1, 8, 32):
        run(16000, 4, B)
    run(48000, 8, 1)
