import sys, time
from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from dpdfnet_amd import backend as be
from dpdfnet_amd.weights import synth_blob
m = be.HipModel(16000, 4, synth_blob(be.manifest(16000, 4), 20260417), 0)
wav = (0.05 * np.random.default_rng(1).standard_normal((256, 160000))).astype(np.float32)
m.enhance_batch(wav)
t0 = time.perf_counter()
for _ in range(3): out = m.enhance_batch(wav)
dt = (time.perf_counter() - t0) / 3
print("host-pointer call (pageable numpy in/out): %.1f ms per step, %.0f frames/s" % (dt * 1e3, 256 * 1003 / dt))
