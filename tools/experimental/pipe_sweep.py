"""Small-batch latency: two-stage pipeline (overlap 27) vs sub-stage pipeline of stage 2 (overlap 59) over chunk lengths."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
import torch
from dpdfnet_amd import backend as be
from dpdfnet_amd.weights import synth_blob
sr, nb = 16000, 4
m = be.HipModel(sr, nb, synth_blob(be.manifest(sr, nb), 20260417), 0)
for B in (1, 8, 32, 64):
    N = 10 * sr
    wav = torch.from_numpy((0.05 * np.random.default_rng(1).standard_normal((B, N))).astype(np.float32)).cuda()
    out = torch.empty_like(wav)
    for ov in (27, 59):
        for ch in (64, 128, 256):
            m.set_overlap(ov); m.set_chunk_frames(ch)
            for _ in range(2): m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None)
            m.sync(); t0 = time.perf_counter()
            for _ in range(4): m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None)
            m.sync()
            print(f"B={B} overlap={ov} chunk={ch}: {(time.perf_counter()-t0)/4*1e3:.2f} ms")
