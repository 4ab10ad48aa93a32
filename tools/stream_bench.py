#!/usr/bin/env python3
"""BASELINE.json configs[4]: 64 concurrent device-resident StreamEnhancer states, dpdfnet8_48khz_hr,
one 10 ms hop per call.  Prints us/hop-call, frames/s and the real-time factor."""
import sys, time, json
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
from dpdfnet_amd import backend as be
from dpdfnet_amd.weights import synth_blob

def run(sr, nb, S, hops_per_call, calls, warm=20, overlap=None, fuse=None):
    m = be.HipModel(sr, nb, synth_blob(be.manifest(sr, nb), 20260417), 0)
    if overlap is not None: m.set_overlap(overlap)
    if fuse is not None: m.set_fuse_dprnn(fuse)
    st = be.HipStreams(m, S)
    rng = np.random.default_rng(0)
    hop = m.hop
    st.prime((0.05 * rng.standard_normal((S, hop))).astype(np.float32))
    pcm = (0.05 * rng.standard_normal((S, hops_per_call * hop))).astype(np.float32)
    for _ in range(warm): st.process(pcm)
    t0 = time.perf_counter()
    for _ in range(calls): st.process(pcm)
    dt = time.perf_counter() - t0
    us = 1e6 * dt / calls
    fps = S * hops_per_call * calls / dt
    audio_per_call = hops_per_call * hop / sr
    print(json.dumps({"overlap": overlap, "fuse": fuse, "sr": sr, "nb": nb, "streams": S, "hops_per_call": hops_per_call, "us_per_call": round(us, 1),
                      "frames_per_s": round(fps), "rtf": round((dt / calls) / audio_per_call, 4)}))
    if "--breakdown" in sys.argv:
        m.set_overlap(0); m.profile(True)
        for _ in range(20): st.process(pcm)
        m.sync()
        rep = m.profile_report(); m.profile(False)
        print("  serial us/call:", {k: round(1e3 * v[0] / 20, 1) for k, v in sorted(rep.items(), key=lambda kv: -kv[1][0])})
        print("  launches/call:", sum(v[1] for v in rep.values()) / 20)
    st.close(); m.close()

if __name__ == "__main__":
    if "--scale" in sys.argv:
        for S in (64, 256, 1024, 4096):
            run(48000, 8, S, 1, 60, warm=10)
            run(16000, 4, S, 1, 60, warm=10)
        sys.exit(0)
    if "--one" in sys.argv:
        run(48000, 8, 64, 1, 200, warm=20); sys.exit(0)
    for hops in (1, 4, 16):
        run(48000, 8, 64, hops, 200 if hops == 1 else 50)
    run(16000, 2, 1, 1, 200)
    if "--ablate" in sys.argv:
        for ov in (0, 1, 3):
            for fu in ("always", "never"):
                run(48000, 8, 64, 1, 200, overlap=ov, fuse=fu)
