#!/usr/bin/env python3
"""float64 analysis DFT (dft64.h) on / off, same process, interleaved: offline 256 x 10 s (48 kHz dpdfnet2, 16 kHz dpdfnet4; device
pointers) and single streaming hops (64 x 48 kHz dpdfnet8, one 16 kHz dpdfnet2 stream).
usage: python tools/dft64_ab.py [offline|hops|all]"""
import sys, time, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from dpdfnet_amd import backend as be
from dpdfnet_amd.weights import synth_blob
what = sys.argv[1] if len(sys.argv) > 1 else "all"
if what in ("offline", "all"):
    for sr, nb in ((48000, 2), (16000, 4)):
        B, N = 256, 10 * sr
        wav = torch.from_numpy((0.05 * np.random.default_rng(1).standard_normal((B, N))).astype(np.float32)).cuda()
        out = torch.empty_like(wav)
        m = be.HipModel(sr, nb, synth_blob(be.manifest(sr, nb), 20260417), 0)
        T = m.num_frames(N)
        res = {0: [], 2: []}
        for rep in range(3):
            for v in (2, 0):
                m.set_option("dft64", v)
                m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None); m.sync()
                t0 = time.perf_counter()
                for _ in range(3): m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None)
                m.sync(); res[v].append((time.perf_counter() - t0) / 3 * 1e3)
        print(json.dumps({"offline": f"{sr}/nb{nb}", "ms_dft64": [round(x, 2) for x in res[2]], "ms_fp32": [round(x, 2) for x in res[0]]}), flush=True)
        for v in (2, 0):
            m.set_option("dft64", v); m.set_overlap(0); m.profile(True)
            m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None); m.sync()
            rep = m.profile_report(); m.profile(False); m.set_overlap(27)
            print(f"  serial stft class dft64={v}:", {k: round(x[0], 2) for k, x in rep.items() if k in ("stft", "istft")}, flush=True)
        m.close(); del wav, out
if what in ("hops", "all"):
    for sr, nb, S in ((48000, 8, 64), (48000, 8, 1), (16000, 2, 1), (16000, 4, 8)):
        m = be.HipModel(sr, nb, synth_blob(be.manifest(sr, nb), 20260417), 0)
        st = be.HipStreams(m, S)
        rng = np.random.default_rng(0)
        st.prime((0.05 * rng.standard_normal((S, m.hop))).astype(np.float32))
        pcm = (0.05 * rng.standard_normal((S, m.hop))).astype(np.float32)
        res = {0: [], 2: []}
        for rep in range(3):
            for v in (2, 0):
                m.set_option("dft64", v)
                for _ in range(30): st.process(pcm)
                t0 = time.perf_counter()
                for _ in range(300): st.process(pcm)
                res[v].append(1e6 * (time.perf_counter() - t0) / 300)
        print(json.dumps({"hop": f"{sr}/nb{nb}/S{S}", "us_dft64": [round(x, 1) for x in res[2]], "us_fp32": [round(x, 1) for x in res[0]]}), flush=True)
        st.close(); m.close()
