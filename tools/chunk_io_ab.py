#!/usr/bin/env python3
"""Device-pointer batch call, 256 x 10 s: per-chunk STFT / iSTFT beside the frame function ("chunk_io") on / off, interleaved;
bit-identity of the two forms.  usage: python tools/chunk_io_ab.py [sr nb]"""
import sys, time, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from dpdfnet_amd import backend as be
from dpdfnet_amd.weights import synth_blob
from bench import synth_clips
sr, nb = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (16000, 4)
B, N = 256, 10 * sr
wav = torch.from_numpy(synth_clips(B, N, sr, 1)).cuda()
outs = {}
m = be.HipModel(sr, nb, synth_blob(be.manifest(sr, nb), 20260417), 0)
T = m.num_frames(N)
for rep in range(3):
    for cio in (1, 0):
        m.set_option("chunk_io", cio)
        out = torch.empty_like(wav)
        m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None); m.sync()
        t0 = time.perf_counter()
        for _ in range(3): m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None)
        m.sync(); dt = (time.perf_counter() - t0) / 3
        outs[cio] = out
        print(json.dumps({"sr": sr, "nb": nb, "chunk_io": cio, "ms": round(dt * 1e3, 2), "frames_per_s": round(B * T / dt)}), flush=True)
print("bit-identical:", bool(torch.equal(outs[0], outs[1])), "finite:", bool(torch.isfinite(outs[1]).all()))
m.close()
