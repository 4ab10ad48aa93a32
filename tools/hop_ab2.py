"""Interleaved A/B of option sets on single hops, same process, alternating: python tools/hop_ab2.py name=val,... name=val,...
(each argument is one option set; '-' = defaults).  Prints min and median us/hop per set for three configurations."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
from dpdfnet_amd import backend as be
from dpdfnet_amd.weights import synth_blob
sets = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in a.split(",") if "=" in kv) for a in sys.argv[1:]] or [{}]
keys = sorted({k for s_ in sets for k in s_})
for sr, nb, S in ((48000, 8, 64), (16000, 2, 1), (16000, 4, 8)):
    m = be.HipModel(sr, nb, synth_blob(be.manifest(sr, nb), 20260417), 0)
    st = be.HipStreams(m, S)
    rng = np.random.default_rng(0)
    st.prime((0.05 * rng.standard_normal((S, m.hop))).astype(np.float32))
    pcm = (0.05 * rng.standard_normal((S, m.hop))).astype(np.float32)
    res = [[] for _ in sets]
    base = {'hop_feat': 1, 'fuse_small': 1, 'dec_seg': 1, 'df_ring': 2, 'scan4_max_wgs': 512, 'stft_ksplit': 7, 'hop_glue': 1, 'fcln_gi': 1,
            'gru256_step': 1, 'fuse_mask': 1, 'hoist_gi': 1, 'glue8': 1, 'fuse_gl': 1, 'interleave': 1, 'fuse_enc': 1, 'fuse_dec': 1, 'overlap': 27, 'single_chunk_inline': 1, 'snapshot': 1, 'late_export': 1, 'hop_prologue': 1, 'hop_dec_fork': 0, 'dual_step': 1, 'hop_pconv': 1, 'dfout_in_decin': 1, 'seg10': 1}
    for rep in range(6):
        for i, opts in enumerate(sets):
            for k in keys:
                v = opts[k] if k in opts else base[k]
                if k == 'overlap': m.set_overlap(v)
                else: m.set_option(k, v)
            for _ in range(15): st.process(pcm)
            t0 = time.perf_counter()
            for _ in range(100): st.process(pcm)
            res[i].append(1e6 * (time.perf_counter() - t0) / 100)
    print(f"sr {sr} nb {nb} streams {S}: " + "   ".join(f"{sys.argv[1 + i] if len(sys.argv) > 1 else '-'}: min {min(r):.0f} med {sorted(r)[len(r) // 2]:.0f}" for i, r in enumerate(res)))
    st.close(); m.close()
