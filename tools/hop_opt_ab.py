"""Config 5 (64 x dpdfnet8_48khz_hr, one hop per call) with single options flipped, interleaved in one process.
usage: python tools/hop_opt_ab.py name=value [name=value ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dpdfnet_amd import backend as be
from dpdfnet_amd.weights import synth_blob
sr, nb, S = 48000, 8, int(os.environ.get("S", "64"))
m = be.HipModel(sr, nb, synth_blob(be.manifest(sr, nb), 20260417), 0)
st = be.HipStreams(m, S)
rng = np.random.default_rng(0)
st.prime((0.05 * rng.standard_normal((S, m.hop))).astype(np.float32))
pcm = (0.05 * rng.standard_normal((S, m.hop))).astype(np.float32)
opts = [a.split("=") for a in sys.argv[1:]]
def timeit(n=300):
    for _ in range(30): st.process(pcm)
    t0 = time.perf_counter()
    for _ in range(n): y = st.process(pcm)
    return 1e6 * (time.perf_counter() - t0) / n, y
base_y = None
for rep in range(3):
    t0, y0 = timeit()
    line = [f"default {t0:.1f}"]
    for k, v in opts:
        default = {"df_ring": 2}.get(k, 1)
        m.set_option(k, int(v)); t1, y1 = timeit(); m.set_option(k, default)
        line.append(f"{k}={v} {t1:.1f} (max diff {float(np.abs(y1 - y0).max()):.1e})")
    print("  ".join(line), flush=True)
