#!/usr/bin/env python3
"""Offline, 256 x 10 s: two-stage DFT (dft2stage.h) on / off, same process, interleaved; per-class serial times.
usage: python tools/dft2_ab.py [sr]"""
import sys, time, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from dpdfnet_amd import backend as be
from dpdfnet_amd.weights import synth_blob
sr, B = (int(sys.argv[1]) if len(sys.argv) > 1 else 48000), 256
N = 10 * sr
wav = torch.from_numpy((0.05 * np.random.default_rng(1).standard_normal((B, N))).astype(np.float32)).cuda()
out = torch.empty_like(wav)
for nb in ((2, 8) if sr == 48000 else (4,)):
    m = be.HipModel(sr, nb, synth_blob(be.manifest(sr, nb), 20260417), 0)
    T = m.num_frames(N)
    for rep in range(2):
        for dft2 in (1, 0):
            m.set_option("dft2", dft2)
            m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None); m.sync()
            t0 = time.perf_counter()
            for _ in range(2): m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None)
            m.sync(); dt = (time.perf_counter() - t0) / 2
            print(json.dumps({"nb": nb, "dft2": dft2, "ms": round(dt * 1e3, 2), "frames_per_s": round(B * T / dt)}), flush=True)
    for dft2 in (1, 0):
        m.set_option("dft2", dft2); m.set_overlap(0); m.profile(True)
        m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None); m.sync()
        rep = m.profile_report(); m.profile(False); m.set_overlap(27)
        print("serial classes dft2=%d:" % dft2, {k: round(v[0], 2) for k, v in rep.items() if k in ("stft", "istft")}, "total", round(sum(v[0] for v in rep.values()), 1), flush=True)
    m.close()
