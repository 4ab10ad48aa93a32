"""A/B of the tail chunk (dpdf_set_option "tail_frames") on big batches of clip lengths whose last chunk is long."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
import torch
from dpdfnet_amd import backend as be
from dpdfnet_amd.weights import synth_blob
m = be.HipModel(16000, 4, synth_blob(be.manifest(16000, 4), 20260417), 0)
B = 256
for secs in (7.0, 9.4, 5.7, 10.0):
    N = int(secs * 16000)
    wav = torch.from_numpy((0.05 * np.random.default_rng(1).standard_normal((B, N))).astype(np.float32)).cuda()
    out = torch.empty_like(wav)
    res = []
    for tail in (0, 32, 0, 32):
        m.set_option("tail_frames", tail)
        for _ in range(2): m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None)
        m.sync(); t0 = time.perf_counter()
        for _ in range(5): m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None)
        m.sync(); res.append((time.perf_counter() - t0) / 5 * 1e3)
    T = m.num_frames(N)
    print(f"{secs} s ({T} frames, last chunk {T % 192 or 192}): tail 0: {res[0]:.2f} / {res[2]:.2f} ms   tail 32: {res[1]:.2f} / {res[3]:.2f} ms")
