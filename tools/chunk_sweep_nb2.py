"""Time-chunk length sweep for the shallow models (nb = 2): stage 2 (GRU-256 chain + decoder) weighs more against stage 1 there.
usage: python tools/chunk_sweep_nb2.py sr nb"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from dpdfnet_amd import backend as be
from dpdfnet_amd.weights import synth_blob
sr, nb, B = int(sys.argv[1]), int(sys.argv[2]), 256
m = be.HipModel(sr, nb, synth_blob(be.manifest(sr, nb), 20260417), 0)
N = 10 * sr
wav = torch.from_numpy((0.05 * np.random.default_rng(1).standard_normal((B, N))).astype(np.float32)).cuda()
out = torch.empty_like(wav)
for chunk in (0, 64, 96, 128, 160, 192, 256, 336):
    m.set_chunk_frames(chunk)
    m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None); m.sync()
    t0 = time.perf_counter()
    for _ in range(3): m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None)
    m.sync()
    print(f"sr {sr} nb {nb} chunk {chunk:4d}: {1e3 * (time.perf_counter() - t0) / 3:7.2f} ms/step", flush=True)
