"""debug library build_ab/lib_S3.so: inside df_apply, compare the values delivered by the compiler-merged (under-aligned dwordx4) loads of the
taps with agent-scope dword loads of the same addresses issued right after."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from dpdfnet_amd import backend as be
from dpdfnet_amd.weights import synth_blob
m = be.HipModel(16000, 4, synth_blob(be.manifest(16000, 4), 20260417), 0)
rng = np.random.default_rng(3)
B, n = 256, 160 * 64 * 8
wav = (0.05 * rng.standard_normal((B, n))).astype(np.float32)
m.set_chunk_frames(64)
L = m._L
out = (ctypes.c_uint * (8 + 960))()
L.dpdf_debug_stamp.argtypes = [ctypes.POINTER(ctypes.c_uint)]
m.set_option("gru64_limbs", 0)
y0 = m.enhance_batch(wav, None)
L.dpdf_debug_stamp(out); prev = list(out)
for limbs in (0, 3, 3, 3, 3, 0):
    m.set_option("gru64_limbs", limbs)
    y = m.enhance_batch(wav, None)
    L.dpdf_debug_stamp(out); cur = list(out)
    d = np.sqrt(np.mean((y - y0) ** 2, axis=1))
    cols = [i for i in range(960) if cur[8 + i] != prev[8 + i]]
    print(f"limbs {limbs}: bad clips {int((d > 1e-6).sum())}; taps compared {cur[1] - prev[1]}, merged-load value != agent-scope value {cur[2] - prev[2]} "
          f"(equal to the previous ring row's value {cur[3] - prev[3]}, zero {cur[4] - prev[4]}, at a 16-byte aligned address {cur[5] - prev[5]}); columns mod 10: "
          f"{sorted(set(c % 10 for c in cols))}, {len(cols)} columns, first {cols[:12]}", flush=True)
    prev = cur
