"""Randomised soak of the streaming entry points: two engines on the same weights -- one with every hop-only form on (defaults), one
with all of them off (the plain chain) -- are driven with the same random sequence of calls (single hops, multi-hop calls, masked
calls, resets, state save / restore through the host) and must agree on every output and on the final states.
argv: seconds [seed].  Prints one JSON line; exit code 1 on a mismatch."""
import json, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
from dpdfnet_amd import backend as be
from dpdfnet_amd.weights import synth_blob
PLAIN = {"single_chunk_inline": 0, "hop_prologue": 0, "late_export": 0, "dual_step": 0, "hop_pconv": 0,
         "fuse_enc": 0, "fuse_dec": 0, "fuse_small": 0}

def rms(x): return float(np.sqrt(np.mean(np.square(x.astype(np.float64)))))

def run(seconds: float, seed: int, min_cases: int = 0) -> dict:
    rng = np.random.default_rng(seed)
    t_end = time.time() + seconds
    cases = calls = 0; worst = 0.0
    while time.time() < t_end or cases < min_cases:   # (min_cases: a fixed floor of cases however slow the box)
        sr, nb = [(16000, 2), (16000, 4), (48000, 1), (48000, 2)][int(rng.integers(4))]
        S = int(rng.choice([1, 2, 3, 4, 5, 8, 17, 64, 70]))
        blob = synth_blob(be.manifest(sr, nb), int(rng.integers(1 << 30)))
        ms, sts = [], []
        for opts in ({}, PLAIN):
            m = be.HipModel(sr, nb, blob, 0)
            for k, v in opts.items(): m.set_option(k, v)
            ms.append(m); sts.append(be.HipStreams(m, S))
        hop = ms[0].hop
        first = (0.05 * rng.standard_normal((S, hop))).astype(np.float32)
        for st in sts: st.prime(first)
        for _ in range(int(rng.integers(10, 40))):
            op = rng.random()
            if op < 0.6:
                T = 1
            elif op < 0.8:
                T = int(rng.integers(2, 6))
            else:
                T = 0
            if T:
                pcm = (0.05 * rng.standard_normal((S, T * hop))).astype(np.float32)
                if rng.random() < 0.25 and S > 1:
                    act = rng.random(S) < 0.6
                    if not act.any(): act[0] = True
                    outs = [st.process_masked(pcm, act) for st in sts]
                    e = rms(outs[0][act] - outs[1][act])
                else:
                    outs = [st.process(pcm) for st in sts]
                    e = rms(outs[0] - outs[1])
                calls += 1
            elif rng.random() < 0.5:
                i = int(rng.integers(S))
                saved = [(st.get_state(i), st.get_tails(i)) for st in sts]
                e = float(np.abs(saved[0][0] - saved[1][0]).max()) * 1e-2
                for st, (s0, (ti, to)) in zip(sts, saved):
                    st.reset(i); st.set_state(i, s0, ti, to)
            else:
                i = int(rng.integers(S))
                for st in sts:
                    st.reset(i); st.prime_one(i, first[i])
                e = 0.0
            worst = max(worst, e)
            if not (e < 2e-6):
                return {"FAIL": True, "sr": sr, "nb": nb, "S": S, "err": e, "calls": calls}
        states = [np.stack([st.get_state(i) for i in range(S)]) for st in sts]
        e = float(np.abs(states[0] - states[1]).max())
        if not (e < 1e-4):
            return {"FAIL": True, "what": "final state", "sr": sr, "nb": nb, "S": S, "err": e}
        for st, m in zip(sts, ms): st.close(); m.close()
        cases += 1
    return {"cases": cases, "calls": calls, "worst_rms": worst, "seconds": seconds}

if __name__ == "__main__":
    rec = run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print(json.dumps(rec))
    sys.exit(1 if rec.get("FAIL") else 0)
