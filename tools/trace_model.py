#!/usr/bin/env python3
"""One model / batch through the offline engine for rocprofv3 --kernel-trace (tools/trace_model.sh): argv = nb B [sr]."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
import torch
from dpdfnet_amd import backend as be
from dpdfnet_amd.weights import synth_blob
nb, B = int(sys.argv[1]), int(sys.argv[2]); sr = int(sys.argv[3]) if len(sys.argv) > 3 else 16000
m = be.HipModel(sr, nb, synth_blob(be.manifest(sr, nb), 20260417), 0)
N = 10 * sr
wav = torch.from_numpy((0.05 * np.random.default_rng(1).standard_normal((B, N))).astype(np.float32)).cuda()
out = torch.empty_like(wav)
for _ in range(3):
    m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None)
m.sync()
