import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dpdfnet_amd import backend as be
from dpdfnet_amd.weights import synth_blob
sr, nb = 16000, 4
m = be.HipModel(sr, nb, synth_blob(be.manifest(sr, nb), 20260417), 0)
rng = np.random.default_rng(3)
B, n = 256, 160 * 64 * 8
wav = (0.05 * rng.standard_normal((B, n))).astype(np.float32)
m.set_chunk_frames(64)
def run(limbs, **opts):
    m.set_option("gru64_limbs", limbs)
    return m.enhance_batch(wav, None)
y0 = run(0)
for tag, opt in [("limbs vs fp32", {"L": 3})] * 10:
    y0 = run(0)
    y1 = run(opt["L"])
    e = np.abs(y1 - y0).reshape(B, -1, 160).max(axis=2)        # [B][frame]
    bad = np.nonzero(e.max(axis=1) > 1e-5)[0]
    desc = []
    for b in bad[:6]:
        ch = sorted(set((np.nonzero(e[b] > 1e-5)[0] // 64).tolist()))
        desc.append(f"{b}:{ch}")
    print(f"{tag}: {len(bad)} bad clips; chunks with errors per clip: {desc}", flush=True)
