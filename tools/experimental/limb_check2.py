"""limbs on vs off, B = 256 x 10 s, under several execution shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dpdfnet_amd import backend as be
from dpdfnet_amd.weights import synth_blob
sr, nb = 16000, 4
m = be.HipModel(sr, nb, synth_blob(be.manifest(sr, nb), 20260417), 0)
rng = np.random.default_rng(3)
B, n = 256, 160000
wav = (0.05 * rng.standard_normal((B, n))).astype(np.float32)
def run(tag):
    outs = {}
    for limbs in (0, 1, 1):
        m.set_option("gru64_limbs", limbs)
        outs.setdefault(limbs, []).append(m.enhance_batch(wav, None))
    d = np.sqrt(np.mean((outs[1][0] - outs[0][0]) ** 2, axis=1))
    rep = np.abs(outs[1][0] - outs[1][1]).max()
    bad = np.nonzero(d > 1e-6)[0]
    # where in time does a bad clip first differ?
    first = []
    for b in bad[:4]:
        e = np.abs(outs[1][0][b] - outs[0][0][b]); first.append(int(np.argmax(e > 1e-5)) // 160)
    print(f"{tag}: median {np.median(d):.2e} max {d.max():.2e}; {len(bad)} bad clips {bad[:8].tolist()} first bad frame {first}; run-to-run {rep:.1e}", flush=True)
run("default")
m.set_overlap(0); run("overlap 0")
m.set_overlap(15); m.set_chunk_frames(1003); run("one chunk")
m.set_chunk_frames(64); run("chunks of 64")
