"""run-to-run determinism of the pipelined multi-chunk run with ablated limb kernels (gru64_dbg): which ingredient disturbs stage 2?"""
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
for tag, limbs, dbg in [("fp32 kernels", 0, 0), ("limbs", 3, 0), ("limbs, no MFMAs", 3, 1), ("limbs, no gate math", 3, 2), ("limbs, neither", 3, 3)]:
    m.set_option("gru64_limbs", limbs); m.set_option("gru64_dbg", dbg)
    ys = [m.enhance_batch(wav, None) for _ in range(4)]
    nb_ = [int((np.abs(y - ys[0]).reshape(B, -1).max(axis=1) > 0).sum()) for y in ys[1:]]
    print(f"{tag}: clips that differ from the first run, in three more runs: {nb_}; finite {bool(np.isfinite(ys[0]).all())}", flush=True)
