"""Same input through two settings of one engine option: python tools/opt_equiv.py <option> [v0 v1] -- prints the largest
difference of the stage tensors and of the waveforms (16 kHz and 48 kHz, a few short clips)."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
from dpdfnet_amd import backend as be
from dpdfnet_amd.weights import synth_blob
opt = sys.argv[1]; v0, v1 = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (0, 1)
KEYS = ("e0", "e1", "e2", "e3", "e3_dprnn", "c0", "c1", "c1_dprnn", "emb", "m", "coefs")
import os
SHAPES = [tuple(float(v) for v in sh.split("x")) for sh in os.environ.get("SHAPES", "3x0.12").split(",")]      # clips x seconds
for sr, nb, (nclip, secs) in [(sr, nb, sh) for sr, nb in ((16000, 2), (48000, 1), (48000, 8)) for sh in SHAPES]:
    m = be.HipModel(sr, nb, synth_blob(be.manifest(sr, nb), 4242), 0)
    rng = np.random.default_rng(1)
    wav = (0.1 * rng.standard_normal((int(nclip), int(secs * sr) + 5))).astype(np.float32)
    res = {}
    for v in (v0, v1):
        m.close(); m = be.HipModel(sr, nb, synth_blob(be.manifest(sr, nb), 4242), 0)      # fresh buffers: a row one form forgets to write must not be inherited
        m.set_option(opt, v)
        out = m.enhance_batch(wav, 100.0)
        res[v] = (out, {k: m.debug_fetch(k) for k in KEYS})
    line = [f"wave {np.abs(res[v0][0] - res[v1][0]).max():.2e}"]
    for k in KEYS:
        a, b = res[v0][1][k], res[v1][1][k]
        line.append(f"{k} {np.abs(a - b).max():.1e}/{np.abs(a).max():.1e}")
    print(f"sr {sr} nb {nb} clips {int(nclip)} x {secs} s {opt} {v0} vs {v1}: " + "  ".join(line))
    m.close()
