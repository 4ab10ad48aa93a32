"""A/B of the 48 kHz decoder stage kernels inside the pipeline: dec_seg = 1 (dec_last.h: dec_seg_kernel) against 2 (dec_seg2.h, tile-
pipelined, three launches) and 3 (one launch), same process, interleaved; 256 x 10 s clips.  python tools/dec_seg_ab.py [nb]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from dpdfnet_amd import backend as be
from dpdfnet_amd.weights import synth_blob
sr, nb, B = 48000, int(sys.argv[1]) if len(sys.argv) > 1 else 2, 256
m = be.HipModel(sr, nb, synth_blob(be.manifest(sr, nb), 20260417), 0)
N = 10 * sr
wav = torch.from_numpy((0.05 * np.random.default_rng(1).standard_normal((B, N))).astype(np.float32)).cuda()
out = torch.empty_like(wav)
ref = None
settings = [(2, 256), (2, 224), (2, 192), (2, 320), (2, 384), (2, 256), (2, 224)] if len(sys.argv) > 2 else [(1, 256), (2, 256), (3, 256), (3, 512), (1, 256), (2, 256), (3, 256)]
for seg, grid in settings:
    m.set_option("dec_seg", seg); m.set_option("dec_seg_grid", grid)
    m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None); m.sync()
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(3): m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None)
        m.sync()
        ts.append(1e3 * (time.perf_counter() - t0) / 3)
    o = out.double()
    if ref is None: ref = o.clone()
    rel = float(((o - ref) ** 2).mean().sqrt() / (ref ** 2).mean().sqrt())
    print(f"nb {nb} dec_seg {seg} grid {grid}: ms/step {min(ts):.2f} (runs {[round(t, 2) for t in ts]})  rel. difference to first setting {rel:.3g}", flush=True)
