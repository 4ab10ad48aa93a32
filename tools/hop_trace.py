"""Single-hop streaming call under rocprofv3 --kernel-trace: python tools/hop_trace.py run <sr> <nb> <streams>  (the traced program)
                                                               python tools/hop_trace.py read <kernel_trace.csv>  (per-hop span vs busy union)"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
if sys.argv[1] == "run":
    import numpy as np
    from dpdfnet_amd import backend as be
    from dpdfnet_amd.weights import synth_blob
    sr, nb, S = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    m = be.HipModel(sr, nb, synth_blob(be.manifest(sr, nb), 20260417), 0)
    for kv in sys.argv[5:]:                       # engine A/B switches: name=int ("overlap=25": the ERB branch on the main stream)
        k, v = kv.split("=")
        if k == "overlap": m.set_overlap(int(v))
        else: m.set_option(k, int(v))
    st = be.HipStreams(m, S)
    rng = np.random.default_rng(0)
    st.prime((0.05 * rng.standard_normal((S, m.hop))).astype(np.float32))
    pcm = (0.05 * rng.standard_normal((S, m.hop))).astype(np.float32)
    for _ in range(30): st.process(pcm)
    t0 = time.perf_counter()
    for _ in range(200): st.process(pcm)
    print("wall us/hop", 1e6 * (time.perf_counter() - t0) / 200)
else:
    import csv
    rows = list(csv.DictReader(open(sys.argv[2])))
    ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0]) for r in rows)
    # hops delimited by stream_ola_kernel (last kernel of a hop)
    hops, cur = [], []
    for e in ev:
        cur.append(e)
        if e[2].startswith('stream_ola_kernel'): hops.append(cur); cur = []
    hops = hops[-100:]
    span = busy = n = 0
    for h in hops:
        s0 = h[0][0]; e1 = max(e[1] for e in h); span += e1 - s0; n += len(h)
        end = s0
        for s, e, _ in sorted(h):
            if e > end: busy += e - max(s, end); end = e
    print("per hop: %.1f launches, span %.1f us, some kernel running %.1f us (%.0f %%)" % (n / len(hops), span / len(hops) / 1e3, busy / len(hops) / 1e3, 100.0 * busy / span))
