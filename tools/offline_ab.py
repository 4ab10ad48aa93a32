"""Interleaved A/B of option sets on an offline batch (device-resident): python tools/offline_ab.py <sr> <nb> <clips> name=val,... name=val,...
('-' = defaults).  Prints ms per step (min over rounds) per set."""
import sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
from dpdfnet_amd import backend as be
from dpdfnet_amd.weights import synth_blob
sr, nb, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
sets = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in a.split(",") if "=" in kv) for a in sys.argv[4:]] or [{}]
keys = sorted({k for s_ in sets for k in s_})
base = {'gru256_c8_tiles': 4, 'gru256_c16_tiles': 2, 'inter_fuse_rows': 1024, 'scan4_max_wgs': 512, 'enc_seg_rows': 512, 'dec_pyr_rows': 512, 'fuse_enc': 1, 'fuse_dec': 1, 'dec_seg': 1}
m = be.HipModel(sr, nb, synth_blob(be.manifest(sr, nb), 20260417), 0)
N = 10 * sr
wav = torch.from_numpy((0.05 * np.random.default_rng(1).standard_normal((B, N))).astype(np.float32)).cuda()
outs = [torch.empty_like(wav) for _ in sets]
res = [[] for _ in sets]
for rep in range(3):
    for i, opts in enumerate(sets):
        for k in keys: m.set_option(k, opts[k] if k in opts else base[k])
        m.enhance_batch_device(wav.data_ptr(), B, N, outs[i].data_ptr(), None); m.sync()
        t0 = time.perf_counter()
        for _ in range(2): m.enhance_batch_device(wav.data_ptr(), B, N, outs[i].data_ptr(), None)
        m.sync()
        res[i].append(1e3 * (time.perf_counter() - t0) / 2)
print(f"sr {sr} nb {nb} clips {B} x 10 s: " + "   ".join(f"{sys.argv[4 + i] if len(sys.argv) > 4 else '-'}: {min(r):.1f} ms" for i, r in enumerate(res))
      + "   max |out_i - out_0|: " + " ".join(f"{(o - outs[0]).abs().max().item():.1e}" for o in outs))
