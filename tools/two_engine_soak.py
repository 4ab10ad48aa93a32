#!/usr/bin/env python3
"""Forward progress of the spinning streaming kernels under co-tenancy: N engine handles (one host thread each, different models and
stream counts) hop at the same time for `seconds`; every handle's outputs must equal the same sequence run alone (bit for bit) and
dpdf_recovery_count must stay 0 (no glue tile or GRU-256 step ever ran into its time-out).  argv: seconds [handles]."""
import json, sys, threading, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
from dpdfnet_amd import backend as be
from dpdfnet_amd.weights import synth_blob

CFGS = [(48000, 8, 64), (16000, 4, 40), (48000, 2, 17), (16000, 2, 1), (16000, 8, 130), (48000, 1, 5)]


def session(cfg, hops, out, barrier=None):
    sr, nb, S = cfg
    m = be.HipModel(sr, nb, synth_blob(be.manifest(sr, nb), 20260417), 0)
    st = be.HipStreams(m, S)
    r = np.random.default_rng(sr + nb + S)
    st.prime((0.05 * r.standard_normal((S, m.hop))).astype(np.float32))
    if barrier is not None:
        barrier.wait()
    acc = np.zeros(hops, np.float64)
    for i in range(hops):
        y = st.process((0.05 * r.standard_normal((S, m.hop))).astype(np.float32))
        acc[i] = float(np.abs(y).astype(np.float64).sum()) + float(y[S // 2, 7])      # a fingerprint of the hop's output
    out.append((acc, int(m.recovery_count)))
    st.close(); m.close()


def run(seconds=60.0, handles=3):
    cfgs = CFGS[:handles]
    t_end = time.time() + seconds
    rounds = hops_total = 0
    while time.time() < t_end:
        hops = [120 + 40 * ((rounds + i) % 3) for i in range(len(cfgs))]
        alone = []
        for c, h in zip(cfgs, hops):
            o = []; session(c, h, o); alone.append(o[0])
        bar = threading.Barrier(len(cfgs))
        outs = [[] for _ in cfgs]
        ths = [threading.Thread(target=session, args=(c, h, outs[i], bar)) for i, (c, h) in enumerate(zip(cfgs, hops))]
        for th in ths: th.start()
        for th in ths: th.join()
        for i, c in enumerate(cfgs):
            acc, rec = outs[i][0]
            if rec or alone[i][1] or not np.array_equal(acc, alone[i][0]):
                return {"FAIL": True, "cfg": c, "recoveries": rec, "first_diff": int(np.argmax(acc != alone[i][0]))}
        rounds += 1; hops_total += sum(hops)
        cfgs = cfgs[1:] + cfgs[:1] if handles < len(CFGS) else cfgs
        if handles < len(CFGS):
            cfgs = (CFGS * 2)[rounds % len(CFGS):][:handles]
    return {"rounds": rounds, "handles": handles, "hops_side_by_side": hops_total, "recoveries": 0, "seconds": seconds}


if __name__ == "__main__":
    rec = run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 3)
    print(json.dumps(rec))
    sys.exit(1 if rec.get("FAIL") else 0)
