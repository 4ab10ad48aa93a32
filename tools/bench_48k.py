#!/usr/bin/env python3
"""Offline throughput of the 48 kHz models (not a BASELINE batch config; sanity of the auto-selected kernel forms)."""
import sys, time, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from dpdfnet_amd import backend as be
from dpdfnet_amd.weights import synth_blob
for sr, nb, B in [(48000, 8, 64), (48000, 8, 256), (48000, 2, 256)]:
    m = be.HipModel(sr, nb, synth_blob(be.manifest(sr, nb), 20260417), 0)
    N = 10 * sr
    wav = torch.from_numpy((0.05 * np.random.default_rng(1).standard_normal((B, N))).astype(np.float32)).cuda()
    out = torch.empty_like(wav)
    m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None); m.sync()
    t0 = time.perf_counter()
    for _ in range(2): m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None)
    m.sync(); dt = (time.perf_counter() - t0) / 2
    T = m.num_frames(N)
    flop = {2: 45.68e6, 8: 136.51e6}[nb]
    print(json.dumps({"sr": sr, "nb": nb, "B": B, "ms": round(dt * 1e3, 1), "frames_per_s": round(B * T / dt), "tflops": round(B * T / dt * flop / 1e12, 1), "finite": bool(torch.isfinite(out).all())}))
    m.close()
