"""Interleaved A/B of dpdf_set_option("fcln_gi") on small offline batches (same box)."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
import torch
from dpdfnet_amd import backend as be
from dpdfnet_amd.weights import synth_blob
for sr, nb, B in ((16000, 4, 1), (16000, 4, 4), (16000, 4, 8), (16000, 4, 16), (16000, 4, 32)):
    m = be.HipModel(sr, nb, synth_blob(be.manifest(sr, nb), 20260417), 0)
    N = int(10.0 * sr)
    wav = torch.from_numpy((0.05 * np.random.default_rng(1).standard_normal((B, N))).astype(np.float32)).cuda()
    out = torch.empty_like(wav)
    res = {0: [], 1: []}
    for rep in range(3):
        for on in (0, 1):
            m.set_option("fcln_gi", on)
            for _ in range(2): m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None)
            m.sync(); t0 = time.perf_counter()
            for _ in range(6): m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None)
            m.sync(); res[on].append((time.perf_counter() - t0) / 6 * 1e3)
    print(f"clips {B}: off {'/'.join(f'{x:.2f}' for x in res[0])}  on {'/'.join(f'{x:.2f}' for x in res[1])}")
    m.close()
