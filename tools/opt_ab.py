"""Same-process A/B of one engine option on the 256 x 10 s workload, interleaved: python tools/opt_ab.py SR NB OPTION V0 V1 [V2 ...]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from dpdfnet_amd import backend as be
from dpdfnet_amd.weights import synth_blob
sr, nb, opt = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
vals = [int(v) for v in sys.argv[4:]]
B, N = 256, 10 * sr
m = be.HipModel(sr, nb, synth_blob(be.manifest(sr, nb), 20260417), 0)
wav = torch.from_numpy((0.05 * np.random.default_rng(1).standard_normal((B, N))).astype(np.float32)).cuda()
out = torch.empty_like(wav)
ref = None
for v in vals + vals:
    m.set_option(opt, v)
    m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None); m.sync()
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(3): m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None)
        m.sync()
        ts.append(1e3 * (time.perf_counter() - t0) / 3)
    o = out.double()
    if ref is None: ref = o.clone()
    print(f"sr {sr} nb {nb} {opt} = {v}: ms/step {min(ts):.2f} (runs {[round(t, 2) for t in ts]})  max |difference| to the first setting {float((o - ref).abs().max()):.3g}", flush=True)
