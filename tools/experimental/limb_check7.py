"""DPDF_ROWSUMS build: which stage-2 tensor rows differ between limb and fp32 kernels in the pipelined run?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dpdfnet_amd import backend as be
from dpdfnet_amd.weights import synth_blob
sr, nb = 16000, 4
m = be.HipModel(sr, nb, synth_blob(be.manifest(sr, nb), 20260417), 0)
rng = np.random.default_rng(3)
B, Tc, NC = 256, 64, 8
n = 160 * Tc * NC
wav = (0.05 * rng.standard_normal((B, n))).astype(np.float32)
m.set_chunk_frames(Tc)
names = ["g256a", "g256b", "g256c(erb gru out)", "g256d(df_ga)", "g256e", "g256f(gc)", "emb", "pconv", "m", "c1d", "enh spec low bins", "coefs ring"]
def run(limbs):
    m.set_option("gru64_limbs", limbs)
    m.debug_fetch("rowsums")                         # resets the chunk counter
    y = m.enhance_batch(wav, None)
    r = m.debug_fetch("rowsums").reshape(16, 12, 256 * 64)[:NC + 1, :, : B * Tc].reshape(NC + 1, 12, B, Tc).copy()
    fr = m.debug_fetch("frames").reshape(B, -1, 320)
    return y, r, np.abs(fr).sum(axis=2)
y0, r0, f0 = run(0)
y0b, r0b, f0b = run(0)
print("fp32 run-to-run rowsum max diff", float(np.abs(r0 - r0b).max()))
for rep in range(3):
    y1, r1, f1 = run(3)
    e = np.abs(y1 - y0).reshape(B, -1, 160).max(axis=2)
    bad = np.argwhere(e > 1e-5)
    print(f"rep {rep}: {len(set(bad[:, 0].tolist()))} bad clips")
    rel = np.abs(r1 - r0) / (np.abs(r0) + 1e-3)
    for k, nm in enumerate(names):
        idx = np.argwhere(rel[:, k] > 1e-3)
        print(f"   {nm:22s} rows off by > 1e-3: {len(idx)}", ("first (chunk, clip, frame): " + str(idx[:4].tolist())) if len(idx) else "")
    fb = np.argwhere(np.abs(f1 - f0) > 1e-3 * (np.abs(f0) + 1e-3))
    print("   synthesis frames (iSTFT input side) off:", len(fb), fb[:6].tolist())
    print("   bad output (clip, frame~) first:", [(int(b), int(f)) for b, f in bad[:6]])
