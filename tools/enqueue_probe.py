"""How long does the HOST take to enqueue one enhance call (device pointers, asynchronous) vs the GPU to finish it?"""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
import torch
from dpdfnet_amd import backend as be
from dpdfnet_amd.weights import synth_blob
for sr, nb, B in ((16000, 4, 1), (16000, 4, 8), (16000, 4, 256)):
    m = be.HipModel(sr, nb, synth_blob(be.manifest(sr, nb), 20260417), 0)
    N = int(10.0 * sr)
    wav = torch.from_numpy((0.05 * np.random.default_rng(1).standard_normal((B, N))).astype(np.float32)).cuda()
    out = torch.empty_like(wav)
    for _ in range(3): m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None)
    m.sync()
    enq, tot = [], []
    for _ in range(6):
        t0 = time.perf_counter(); m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None); t1 = time.perf_counter(); m.sync(); t2 = time.perf_counter()
        enq.append((t1 - t0) * 1e3); tot.append((t2 - t0) * 1e3)
    print(f"clips {B}: host enqueue {np.median(enq):.2f} ms, until done {np.median(tot):.2f} ms")
    m.close()
