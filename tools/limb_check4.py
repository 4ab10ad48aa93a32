"""The limb kernels (gru64_limbs = 3, the default) against the fp32-MFMA kernels (gru64_limbs = 0), 256 clips through the multi-chunk pipeline (eight 64-frame
chunks: stage 2 of a chunk under stage 1 of the next), run after run: any clip further than 1e-5 (max abs) from the fp32 path is a
"bad clip" (single wrong low-band frames before the taps were read with agent-scope loads).
usage: python tools/limb_check4.py [runs=10] [also_automatic_schedule=0]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dpdfnet_amd import backend as be
from dpdfnet_amd.weights import synth_blob
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 10
sr, nb = 16000, 4
m = be.HipModel(sr, nb, synth_blob(be.manifest(sr, nb), 20260417), 0)
rng = np.random.default_rng(3)
B, n = 256, 160 * 64 * 8
wav = (0.05 * rng.standard_normal((B, n))).astype(np.float32)
m.set_chunk_frames(64)
def run(limbs):
    m.set_option("gru64_limbs", limbs)
    return m.enhance_batch(wav, None)
y0 = run(0)
bad_runs, bad_clips, nondet = 0, 0, 0
first = None
for r in range(runs):
    y1 = run(3)
    e = np.abs(y1 - y0).reshape(B, -1, 160).max(axis=2)        # [B][frame]
    bad = np.nonzero(e.max(axis=1) > 1e-5)[0]
    if len(bad):
        bad_runs += 1; bad_clips += len(bad)
        print(f"run {r}: {len(bad)} bad clips, e.g. " + ", ".join(f"{b}:{sorted(set((np.nonzero(e[b] > 1e-5)[0] // 64).tolist()))}" for b in bad[:6]), flush=True)
    if first is None: first = y1
    elif not np.array_equal(first, y1): nondet += 1
print(f"{runs} runs of 256 clips x 8 chunks: {bad_runs} runs with bad clips ({bad_clips} clips in all), {nondet} runs not bit-identical to the first", flush=True)
