"""A/B of the GRU-256 small-launch forms (cluster8 vs cluster16) on latency-bound calls, same process, alternating."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from dpdfnet_amd import backend as be
from dpdfnet_amd.weights import synth_blob
for sr, nb, Bs in ((16000, 4, (1, 8, 16, 32, 64)), (48000, 8, (1, 16, 64))):
    m = be.HipModel(sr, nb, synth_blob(be.manifest(sr, nb), 20260417), 0)
    for B in Bs:
        N = 10 * sr
        wav = torch.from_numpy((0.05 * np.random.default_rng(1).standard_normal((B, N))).astype(np.float32)).cuda(); out = torch.empty_like(wav)
        res = {}
        for rep in range(2):
            for t in (0, 1, 2, 4):
                m.set_option("gru256_c16_tiles", t)
                for _ in range(2): m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None)
                m.sync(); t0 = time.perf_counter()
                for _ in range(3): m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None)
                m.sync(); res.setdefault(t, []).append((time.perf_counter() - t0) / 3 * 1e3)
        print(f"sr={sr} nb={nb} B={B}: " + "  ".join(f"c16<={t}: {min(v):.2f}" for t, v in res.items()))
    m.close()
