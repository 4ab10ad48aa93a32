import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dpdfnet_amd import backend as be
from dpdfnet_amd.weights import synth_blob
sr, nb = 16000, 4
m = be.HipModel(sr, nb, synth_blob(be.manifest(sr, nb), 20260417), 0)
rng = np.random.default_rng(3)
B, n = 256, 160 * 64 * 8
wav = (0.05 * rng.standard_normal((B, n))).astype(np.float32)
m.set_chunk_frames(64)
def run(limbs):
    m.set_option("gru64_limbs", limbs)
    return m.enhance_batch(wav, None)
y0 = run(0)
for rep in range(3):
    y1 = run(1)
    e = np.abs(y1 - y0).reshape(B, -1, 160).max(axis=2)
    bad = np.nonzero(e.max(axis=1) > 1e-5)[0]
    print("rep", rep, "bad clips", bad.tolist())
    for b in bad[:5]:
        fr = np.nonzero(e[b] > 1e-6)[0]
        print(f"  clip {b}: frames {fr.min()}..{fr.max()} ({len(fr)}), profile", " ".join(f"{v:.0e}" for v in e[b][fr.min(): fr.min() + 24]))
# spectrum of the difference in the bad frames of the last rep
for b in bad[:6]:
    fr = np.nonzero(e[b] > 1e-6)[0]
    seg = (y1[b] - y0[b])[fr.min() * 160: (fr.max() + 1) * 160]
    sp = np.abs(np.fft.rfft(seg * np.hanning(len(seg)), 640))
    band = sp.reshape(-1)[:320]
    lo, hi = band[: 96 * 2].max(), band[96 * 2:].max()
    print(f"  clip {b}: difference spectrum max below 4.8 kHz {lo:.2e}, above {hi:.2e}; peak bin (of 320 = 8 kHz) {int(band.argmax())}")
