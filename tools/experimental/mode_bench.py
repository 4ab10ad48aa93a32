"""ms/step of the headline workload under engine options: python tools/mode_bench.py "name=v,name=v" ... (one run per argument)"""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
import torch
from dpdfnet_amd import backend as be
from dpdfnet_amd.weights import synth_blob
sr, nb, B = 16000, int(__import__("os").environ.get("NB", "4")), 256
m = be.HipModel(sr, nb, synth_blob(be.manifest(sr, nb), 20260417), 0)
N = 10 * sr
wav = torch.from_numpy((0.05 * np.random.default_rng(1).standard_normal((8, N))).astype(np.float32)).cuda().repeat(B // 8, 1).contiguous()
out = torch.empty_like(wav)
for spec in sys.argv[1:] or ["-"]:
    for kv in spec.split(","):
        if "=" in kv:
            k, v = kv.split("="); m.set_option(k, int(v))
    for ov in (27, 0):
        m.set_overlap(ov)
        for _ in range(2): m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None)
        m.sync(); t0 = time.perf_counter()
        for _ in range(4): m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None)
        m.sync(); ms = (time.perf_counter() - t0) / 4 * 1e3
        print(f"{spec:32s} overlap={ov:2d}: {ms:7.2f} ms/step  {B * m.num_frames(N) / ms * 1e3 / 1e6:.3f} M frames/s")
    m.set_overlap(0); m.profile(True); m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None); m.sync()
    rep = m.profile_report(); m.profile(False); m.set_overlap(27)
    print("   serial per class:", {k: round(v[0], 2) for k, v in sorted(rep.items(), key=lambda kv: -kv[1][0])[:9]})
