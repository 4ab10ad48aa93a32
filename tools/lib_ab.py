#!/usr/bin/env python3
"""A/B of two builds of the library (DPDFNET_HIP_LIB) on one offline configuration, each build in processes of its own, interleaved.
usage: python tools/lib_ab.py <other.so> [sr nb]   (prints ms/step + the serial per-class times of both builds)"""
import json, os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
other = sys.argv[1]
sr, nb = (sys.argv[2], sys.argv[3]) if len(sys.argv) > 3 else ("48000", "2")
CHILD = r'''
import sys, time, json
sys.path.insert(0, %r)
import numpy as np, torch
from dpdfnet_amd import backend as be
from dpdfnet_amd.weights import synth_blob
sr, nb, B = int(sys.argv[1]), int(sys.argv[2]), 256
N = 10 * sr
wav = torch.from_numpy((0.05 * np.random.default_rng(1).standard_normal((B, N))).astype(np.float32)).cuda()
out = torch.empty_like(wav)
m = be.HipModel(sr, nb, synth_blob(be.manifest(sr, nb), 20260417), 0)
m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None); m.sync()
t0 = time.perf_counter()
for _ in range(3): m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None)
m.sync(); dt = (time.perf_counter() - t0) / 3
m.set_overlap(0); m.profile(True)
m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None); m.sync()
rep = m.profile_report()
print("RES " + json.dumps({"ms": round(dt * 1e3, 2), "serial": {k: round(v[0], 2) for k, v in rep.items() if v[0] > 1.0}, "sum": float(out.double().abs().sum())}))
''' % str(ROOT)
for rep in range(2):
    for name, lib in (("tree", ""), ("other", other)):
        env = dict(os.environ)
        if lib:
            env["DPDFNET_HIP_LIB"] = str(Path(lib).resolve())
        r = subprocess.run([sys.executable, "-c", CHILD, sr, nb], capture_output=True, text=True, env=env)
        line = next((l for l in r.stdout.splitlines() if l.startswith("RES ")), r.stderr[-300:])
        print(name, line, flush=True)
