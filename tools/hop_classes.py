"""Per-class GPU time of streaming hops from the engine's own HIP-event scopes (no profiler attached, so the host is not
slowed down): python tools/hop_classes.py <sr> <nb> <streams> [opt=val ...]"""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
from dpdfnet_amd import backend as be
from dpdfnet_amd.weights import synth_blob
sr, nb, S = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
m = be.HipModel(sr, nb, synth_blob(be.manifest(sr, nb), 20260417), 0)
for kv in sys.argv[4:]:
    k, v = kv.split("="); m.set_option(k, int(v))
st = be.HipStreams(m, S)
rng = np.random.default_rng(0)
st.prime((0.05 * rng.standard_normal((S, m.hop))).astype(np.float32))
pcm = (0.05 * rng.standard_normal((S, m.hop))).astype(np.float32)
for _ in range(30): st.process(pcm)
t0 = time.perf_counter()
for _ in range(100): st.process(pcm)
plain = 1e6 * (time.perf_counter() - t0) / 100
for mode, ov in (("pipelined", 27), ("serial", 0)):
    m.set_overlap(ov); m.profile(True)
    t0 = time.perf_counter()
    for _ in range(50): st.process(pcm)
    wall = 1e6 * (time.perf_counter() - t0) / 50
    rep = m.profile_report(); m.profile(False)
    tot = sum(v[0] for v in rep.values()) * 1e3 / 50
    print(f"{mode}: wall {wall:.0f} us/hop (no scopes: {plain:.0f}); sum of class times {tot:.0f} us, launches/hop {sum(v[1] for v in rep.values()) / 50:.0f}")
    for k, v in sorted(rep.items(), key=lambda kv: -kv[1][0]):
        print(f"   {k:42s} {1e3 * v[0] / 50:8.1f} us/hop  {v[1] / 50:5.1f} scopes  {1e3 * v[0] / max(1, v[1]):7.1f} us each")
m.set_overlap(27)
st.close(); m.close()
