"""gru64_limbs on vs off through the engine: per-clip RMS difference for several batch shapes (fuse_dprnn forced on)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dpdfnet_amd import backend as be
from dpdfnet_amd.weights import synth_blob
sr, nb = 16000, int(sys.argv[1]) if len(sys.argv) > 1 else 4
m = be.HipModel(sr, nb, synth_blob(be.manifest(sr, nb), 20260417), 0)
rng = np.random.default_rng(3)
for B, n, fuse in [(1, 16000, 2), (16, 16000, 2), (19, 16000, 2), (19, 24000, 2), (64, 32000, 2), (256, 160000, 1), (256, 160000, 1)]:
    wav = (0.05 * rng.standard_normal((B, n))).astype(np.float32)
    m.set_fuse_dprnn({2: "always", 1: "auto"}[fuse])
    outs = {}
    for limbs in (0, 1, 1):
        m.set_option("gru64_limbs", limbs)
        outs.setdefault(limbs, []).append(m.enhance_batch(wav, None))
    d = np.sqrt(np.mean((outs[1][0] - outs[0][0]) ** 2, axis=1))
    rep = np.abs(outs[1][0] - outs[1][1]).max()
    worst = np.argsort(d)[-4:][::-1]
    print(f"B {B:4d} n {n:7d} fuse {fuse}: limbs vs fp32 per-clip RMS median {np.median(d):.2e} max {d.max():.2e} at clips {worst.tolist()} ({d[worst].round(7).tolist()}); limbs run-to-run max diff {rep:.1e}", flush=True)
