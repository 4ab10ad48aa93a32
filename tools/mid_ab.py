"""Mid-size offline batches: ms per call for option sets (interleaved).  python tools/mid_ab.py opt=val,... opt=val,..."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
import torch
from dpdfnet_amd import backend as be
from dpdfnet_amd.weights import synth_blob
sets = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in a.split(",") if "=" in kv) for a in sys.argv[1:]] or [{}]
for sr, nb, B in ((16000, 4, 8), (16000, 4, 16), (16000, 4, 32), (16000, 4, 64), (16000, 4, 128)):
    N = 10 * sr
    wav = torch.from_numpy((0.05 * np.random.default_rng(1).standard_normal((B, N))).astype(np.float32)).cuda()
    out = torch.empty_like(wav)
    ms = []
    for opts in sets:
        m = be.HipModel(sr, nb, synth_blob(be.manifest(sr, nb), 20260417), 0)
        for k, v in opts.items(): m.set_option(k, v)
        for _ in range(2): m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None)
        m.sync()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(3): m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None)
            m.sync(); best = min(best, (time.perf_counter() - t0) / 3 * 1e3)
        ms.append(best); m.close()
    print(f"clips {B}: " + "   ".join(f"{sys.argv[1 + i] if len(sys.argv) > 1 else '-'}: {t:.2f} ms" for i, t in enumerate(ms)))
