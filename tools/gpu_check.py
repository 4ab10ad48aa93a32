#!/usr/bin/env python3
"""Ad-hoc GPU parity probe: HIP engine vs CPU oracle vs reference goldens, stage by stage."""
import json, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from oracle import oracle as orc
from dpdfnet_amd import backend as be
from dpdfnet_amd.weights import synth_blob

def rms(x): return float(np.sqrt(np.mean(np.square(x, dtype=np.float64))))

def check(tag, chunk=0):
    g = np.load(ROOT / f"tests/golden/model_{tag}.npz")
    meta = json.loads(bytes(g["meta_json"]).decode())
    sr, nb, seed = meta["sample_rate"], meta["nb"], meta["seed"]
    C = np.load(ROOT / "tests/golden/constants.npz")
    k = "16k" if sr == 16000 else "48k"
    ents = be.manifest(sr, nb)
    blob = synth_blob(ents, seed)
    o = orc.Oracle(sr, nb, blob, C[f"erb_norm_init_{k}"], C[f"spec_norm_init_{k}"])
    m = be.HipModel(sr, nb, blob, 0, C[f"erb_norm_init_{k}"], C[f"spec_norm_init_{k}"])
    m.set_chunk_frames(chunk)
    wav = g["wav"]
    spec = o.stft(wav)
    T = spec.shape[0]
    d = be.query_dims(sr, nb)
    # oracle frame loop with probes
    probe_at = sorted({0, 1, 5, T - 1})
    st = o.initial_state(); ref_out = np.zeros_like(spec); probes = {}
    names = ["feat_erb", "e0", "e1", "e2", "e3", "e3_dprnn", "c0", "c1", "c1_dprnn", "emb", "m", "coefs"]
    for t in range(T):
        ref_out[t], st = o.frame(spec[t], st)
        if t in probe_at:
            probes[t] = {n: o.probe(n) for n in names}
    out, st_gpu = m.run_frames(spec, m.initial_state())
    print(f"[{tag} chunk={chunk}] spec_e maxdiff {np.abs(out-ref_out).max():.3e} (scale {np.abs(ref_out).max():.2f})  rms {rms(out-ref_out):.3e}; state maxdiff {np.abs(st_gpu-st).max():.3e}")
    if chunk == 0:
        shapes = {"e0": (d.Ec, 64), "e1": (d.F1, 64), "e2": (d.F2, 64), "e3": (d.F3, 64), "e3_dprnn": (d.F3, 64),
                  "c1": (d.Fd, 64), "c1_dprnn": (d.Fd, 64)}
        for n in names:
            try: buf = m.debug_fetch(n)
            except ValueError: continue
            for t in probe_at:
                ref = probes[t][n]
                if n == "feat_erb": mine = buf.reshape(T + 2, d.E)[2 + t]
                elif n in shapes: mine = buf.reshape(T, *shapes[n])[t].T.reshape(-1)
                elif n == "c0": mine = buf.reshape(T + 4, d.D, 64)[4 + t].T.reshape(-1)
                elif n == "emb": mine = buf.reshape(T, 512)[t]
                elif n == "m": mine = buf.reshape(T, d.E)[t][:ref.size]
                elif n == "coefs": mine = buf.reshape(T + 2, d.D, 5, 2)[2 + t].transpose(1, 0, 2).reshape(-1)
                err = np.abs(mine - ref).max()
                flag = "" if err < 1e-4 * max(1.0, np.abs(ref).max()) else "   <<<<<<"
                print(f"    t={t:3d} {n:9s} maxdiff {err:.3e} scale {np.abs(ref).max():.3f}{flag}")
    enh = m.enhance_batch(np.stack([wav, wav[::-1].copy()]))
    ref = g["enhanced"]
    print(f"    enhance_batch vs golden: rms err {rms(enh[0]-ref):.3e} (signal rms {rms(ref):.4f}); vs oracle(reversed clip) {rms(enh[1]-o.enhance(wav[::-1].copy())):.3e}")
    for db in (0, 12):
        e2 = m.enhance_batch(wav[None], db)[0]
        print(f"    attn{db}: rms err {rms(e2-g[f'enhanced_attn{db}']):.3e}")
    m.close()

if __name__ == "__main__":
    tags = sys.argv[1:] or ["16k_nb0", "16k_nb1", "16k_nb2", "16k_nb4"]
    for tag in tags:
        check(tag, 0)
        check(tag, 16)
