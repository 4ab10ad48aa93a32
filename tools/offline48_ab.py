import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from dpdfnet_amd import backend as be
from dpdfnet_amd.weights import synth_blob
sr, nb, B = 48000, int(sys.argv[1]), 256
m = be.HipModel(sr, nb, synth_blob(be.manifest(sr, nb), 20260417), 0)
N = 10 * sr
wav = torch.from_numpy((0.05 * np.random.default_rng(1).standard_normal((B, N))).astype(np.float32)).cuda()
out = torch.empty_like(wav)
m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None); m.sync()
for rep in range(2):
    t0 = time.perf_counter()
    for _ in range(3): m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None)
    m.sync()
    print("nb", nb, "ms/step", round(1e3 * (time.perf_counter() - t0) / 3, 2), "checksum", float(out.double().abs().sum()))
