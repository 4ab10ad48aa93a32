"""Interleaved A/B of the single-hop streaming switches on one box: us per hop for (sr, nb, streams) x option sets."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
from dpdfnet_amd import backend as be
from dpdfnet_amd.weights import synth_blob
SETS = {"base": {"hop_glue": 0, "stft_ksplit": 0, "fcln_gi": 0, "gru256_step": 0}, "glue+stft+step": {"hop_glue": 1, "stft_ksplit": 1, "fcln_gi": 1, "gru256_step": 1},
        "+istft": {"hop_glue": 1, "stft_ksplit": 3, "fcln_gi": 1, "gru256_step": 1}}
for sr, nb, S in ((48000, 8, 64), (16000, 2, 1), (16000, 4, 8)):
    m = be.HipModel(sr, nb, synth_blob(be.manifest(sr, nb), 20260417), 0)
    st = be.HipStreams(m, S)
    rng = np.random.default_rng(0)
    st.prime((0.05 * rng.standard_normal((S, m.hop))).astype(np.float32))
    pcm = (0.05 * rng.standard_normal((S, m.hop))).astype(np.float32)
    res = {k: [] for k in SETS}
    for rep in range(3):
        for name, opts in SETS.items():
            for k, v in opts.items(): m.set_option(k, v)
            for _ in range(20): st.process(pcm)
            t0 = time.perf_counter()
            for _ in range(150): st.process(pcm)
            res[name].append(1e6 * (time.perf_counter() - t0) / 150)
    print(f"sr {sr} nb {nb} streams {S}: " + "  ".join(f"{k}: {'/'.join(f'{x:.0f}' for x in v)}" for k, v in res.items()))
    st.close(); m.close()
