"""A/B of the stacked decoder-pair GRU-256 launch (dpdf_set_option "gru256_stack") on small batches."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
import torch
from dpdfnet_amd import backend as be
from dpdfnet_amd.weights import synth_blob

for sr, nb, B in ((16000, 4, 1), (16000, 4, 8), (16000, 4, 32), (48000, 8, 1)):
    m = be.HipModel(sr, nb, synth_blob(be.manifest(sr, nb), 20260417), 0)
    N = int(10.0 * sr)
    wav = torch.from_numpy((0.05 * np.random.default_rng(1).standard_normal((B, N))).astype(np.float32)).cuda()
    out = torch.empty_like(wav)
    res = []
    for on in (0, 1):
        m.set_option("gru256_stack", on)
        for _ in range(2): m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None)
        m.sync(); t0 = time.perf_counter()
        for _ in range(8): m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None)
        m.sync(); ms = (time.perf_counter() - t0) / 8 * 1e3
        m.set_overlap(0); m.profile(True)
        m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None); m.sync()
        rep = m.profile_report(); m.profile(False); m.set_overlap(-1 if False else 27)
        res.append((ms, rep.get("gru256_scan", (0, 0))))
    print(f"sr {sr} nb {nb} clips {B}: off {res[0][0]:.2f} ms (serial gru256_scan {res[0][1][0]:.2f} ms / {res[0][1][1]} launches), on {res[1][0]:.2f} ms ({res[1][1][0]:.2f} ms / {res[1][1][1]})")
    m.close()
