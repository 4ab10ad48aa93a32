#!/usr/bin/env python3
"""GRU-256 input projection inside the cluster scan (gru_clusterx.h) on / off: offline 256 x 10 s, interleaved, same process;
waveform difference between the two forms and against each other's serial class times.  usage: python tools/fusedx_ab.py [sr nb]"""
import sys, time, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from dpdfnet_amd import backend as be
from dpdfnet_amd.weights import synth_blob
from bench import synth_clips
sr, nb = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (16000, 4)
import os
B, N = int(os.environ.get("CLIPS", "256")), 10 * sr
wav = torch.from_numpy(synth_clips(B, N, sr, 1)).cuda()
m = be.HipModel(sr, nb, synth_blob(be.manifest(sr, nb), 20260417), 0)
T = m.num_frames(N)
outs = {}
for rep in range(3):
    for fx in (1, 0):
        m.set_option("gru256_fused_x", fx)
        out = torch.empty_like(wav)
        m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None); m.sync()
        t0 = time.perf_counter()
        for _ in range(3): m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None)
        m.sync(); dt = (time.perf_counter() - t0) / 3
        outs[fx] = out
        print(json.dumps({"sr": sr, "nb": nb, "fused_x": fx, "ms": round(dt * 1e3, 2), "frames_per_s": round(B * T / dt)}), flush=True)
d = (outs[0] - outs[1]).double()
print("rms diff", float(d.pow(2).mean().sqrt()), "max", float(d.abs().max()), "signal rms", float(outs[0].double().pow(2).mean().sqrt()), "finite", bool(torch.isfinite(outs[1]).all()))
for fx in (1, 0):
    m.set_option("gru256_fused_x", fx); m.set_overlap(0); m.profile(True)
    m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None); m.sync()
    rep = m.profile_report(); m.profile(False); m.set_overlap(27)
    print("serial fused_x=%d:" % fx, {k: round(v[0], 2) for k, v in rep.items() if "gru256" in k}, "total", round(sum(v[0] for v in rep.values()), 1), flush=True)
m.close()
