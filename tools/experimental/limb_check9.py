"""debug library build_ab/lib_S.so (stamps): did df_apply read taps BEFORE df_out wrote their row (ordering), or a stale copy (coherence)?"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dpdfnet_amd import backend as be
from dpdfnet_amd.weights import synth_blob
m = be.HipModel(16000, 4, synth_blob(be.manifest(16000, 4), 20260417), 0)
rng = np.random.default_rng(3)
B, n = 256, 160 * 64 * 8
wav = (0.05 * rng.standard_normal((B, n))).astype(np.float32)
m.set_chunk_frames(64)
L = m._L
out = (ctypes.c_uint * (4 + 960))()
L.dpdf_debug_stamp.argtypes = [ctypes.POINTER(ctypes.c_uint)]
m.set_option("gru64_limbs", 0)
y0 = m.enhance_batch(wav, None)
L.dpdf_debug_stamp(out); prev = list(out)
for limbs in (0, 3, 3, 3, 3, 0):
    m.set_option("gru64_limbs", limbs)
    y = m.enhance_batch(wav, None)
    L.dpdf_debug_stamp(out); cur = list(out)
    d = np.sqrt(np.mean((y - y0) ** 2, axis=1))
    cols = [i for i in range(960) if cur[4 + i] != prev[4 + i]]
    print(f"limbs {limbs}: bad clips {int((d > 1e-6).sum())}; taps compared {cur[1] - prev[1]}, plain != agent-scope {cur[2] - prev[2]}; columns (of 960) with mismatches: {cols[:40]} ({len(cols)} columns)", flush=True)
    prev = cur
