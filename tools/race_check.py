import sys, json, numpy as np
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
from tests.util import load_golden, golden_blob, make_oracle, norm_inits
from dpdfnet_amd import backend as be
tag = sys.argv[1] if len(sys.argv) > 1 else "48k_nb1"
g, meta = load_golden(tag); blob = golden_blob(meta); e, s = norm_inits(meta["sample_rate"])
o = make_oracle(meta, blob)
m = be.HipModel(meta["sample_rate"], meta["nb"], blob, 0, e, s)
d = be.query_dims(meta["sample_rate"], meta["nb"])
segs = [("erb_norm", d.E), ("spec_norm", d.D), ("erb_conv0_buf", 3*d.E), ("dprnn_erb", d.nb*d.F3*64), ("df_conv0_buf", 6*d.D),
        ("dprnn_df", d.nb*d.Fd*64), ("emb_gru", 256), ("erb_dec", 512), ("df_dec_gru", 512), ("df_convp", 5*64*d.D),
        ("mask_buf", 6*d.F), ("coefs_buf", 30*d.D), ("spec_buf", 10*d.F)]
NT = int(sys.argv[2]) if len(sys.argv) > 2 else 6
spec = o.stft(g["wav"])[:NT]
m.set_chunk_frames(0); ref, st_ref = m.run_frames(spec, m.initial_state())
m.set_overlap(27); m.set_chunk_frames(1)
bad = 0
for rep in range(300):
    out, st = m.run_frames(spec, m.initial_state())
    dd = np.abs(out - ref)
    if dd.max() > 1e-4 * np.abs(ref).max():
        bad += 1
        if bad <= 6:
            fr = sorted(set(np.argwhere(dd > 1e-4 * np.abs(ref).max())[:, 0].tolist()))
            off = 0; rep_s = []
            for name, n in segs:
                dm = np.abs(st[off:off+n] - st_ref[off:off+n]).max(); off += n
                if dm > 1e-4: rep_s.append(f"{name}:{dm:.2e}")
            print(f"rep {rep}: bad frames {fr}; state segs differing: {rep_s}")
print(f"{bad}/300 bad with T={NT}")
