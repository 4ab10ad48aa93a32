"""DPDF_COEFS_CHECK build: inside df_apply, do plain loads of the deep-filter taps differ from agent-scope loads of the same address?"""
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
out = (ctypes.c_uint * 2)()
L.dpdf_debug_coefs_mismatch.argtypes = [ctypes.POINTER(ctypes.c_uint)]
prev = [0, 0]
for limbs in (0, 3, 3, 3, 0, 3):
    m.set_option("gru64_limbs", limbs)
    y = m.enhance_batch(wav, None)
    L.dpdf_debug_coefs_mismatch(out)
    print(f"limbs {limbs}: taps where the plain (volatile) load != the agent-scope load: {out[0] - prev[0]}, non-temporal load != agent-scope: {out[1] - prev[1]}", flush=True)
    prev = [out[0], out[1]]
