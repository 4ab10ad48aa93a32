"""Pipelined multi-chunk run, limbs on vs off: which stage tensor of the LAST chunk differs first?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dpdfnet_amd import backend as be
from dpdfnet_amd.weights import synth_blob
sr, nb = 16000, 4
m = be.HipModel(sr, nb, synth_blob(be.manifest(sr, nb), 20260417), 0)
rng = np.random.default_rng(3)
B, n = 256, 160 * 64 * 6
wav = (0.05 * rng.standard_normal((B, n))).astype(np.float32)
m.set_chunk_frames(64)
names = ["feat_erb", "feat_spec", "e0", "e3", "c0", "c1", "e3_dprnn", "c1_dprnn", "emb", "m", "coefs"]
def run(limbs):
    m.set_option("gru64_limbs", limbs)
    y = m.enhance_batch(wav, None)
    return y, {k: m.debug_fetch(k).copy() for k in names}
y0, t0 = run(0)
for rep in range(4):
    y1, t1 = run(1)
    d = np.sqrt(np.mean((y1 - y0) ** 2, axis=1))
    print(f"rep {rep}: bad clips {int((d > 1e-6).sum())}; " + "  ".join(f"{k} {np.abs(t1[k] - t0[k]).max():.1e}" for k in names), flush=True)
y0b, t0b = run(0)
print("fp32 run-to-run:", np.abs(y0b - y0).max(), "  ".join(f"{k} {np.abs(t0b[k] - t0[k]).max():.1e}" for k in names))
