"""Hop time against the number of DPRNN blocks: slope = dependent chain per block, intercept = everything else."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
from dpdfnet_amd import backend as be
from dpdfnet_amd.weights import synth_blob
for sr, S in ((48000, 64), (16000, 1), (16000, 64)):
    res = []
    for nb in (0, 1, 2, 4, 8):
        m = be.HipModel(sr, nb, synth_blob(be.manifest(sr, nb), 20260417), 0)
        for kv in sys.argv[1:]:
            k, v = kv.split("="); m.set_option(k, int(v))
        st = be.HipStreams(m, S)
        rng = np.random.default_rng(0)
        st.prime((0.05 * rng.standard_normal((S, m.hop))).astype(np.float32))
        pcm = (0.05 * rng.standard_normal((S, m.hop))).astype(np.float32)
        for _ in range(30): st.process(pcm)
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            for _ in range(100): st.process(pcm)
            best = min(best, 1e6 * (time.perf_counter() - t0) / 100)
        res.append((nb, best))
        st.close(); m.close()
    print(f"sr {sr} streams {S}: " + "  ".join(f"nb{nb}: {t:.0f} us" for nb, t in res) + f"   per block (8 vs 1): {(res[-1][1] - res[1][1]) / 7:.1f} us")
