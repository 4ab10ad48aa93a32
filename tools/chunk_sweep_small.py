import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from dpdfnet_amd import backend as be
from dpdfnet_amd.weights import synth_blob
for sr, nb, B in ((16000, 4, 1), (16000, 4, 8), (48000, 8, 1)):
    m = be.HipModel(sr, nb, synth_blob(be.manifest(sr, nb), 20260417), 0)
    N = 10 * sr
    wav = torch.from_numpy((0.05 * np.random.default_rng(1).standard_normal((B, N))).astype(np.float32)).cuda()
    out = torch.empty_like(wav)
    res = {}
    for chunk in (0, -1, 512, 128, 64):
        m.set_chunk_frames(chunk)
        for _ in range(2): m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None)
        m.sync()
        t0 = time.perf_counter()
        for _ in range(5): m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None)
        m.sync()
        res[chunk] = round(1e3 * (time.perf_counter() - t0) / 5, 2)
    print(sr, nb, B, res)
    m.close()
