#!/usr/bin/env python3
"""Forward progress of a streaming hop's cross-stream counter join (hop_spin_join: the first kernel of stage 2 spins on a counter that
the ERB stack's last block, a kernel of ANOTHER stream of the same handle, bumps) while the SAME PROCESS runs other GPU work on its
own streams: torch GEMMs and elementwise kernels on two torch streams, issued by a second host thread, for the whole soak.  With
HIP's default of four hardware queues (GPU_MAX_HW_QUEUES=4) the engine's four streams and torch's share queues.  Every hop must
equal the same input sequence run WITHOUT the co-tenant (bit for bit: the engine is deterministic) and dpdf_recovery_count must
stay 0 -- no wait ran into its time-out.  argv: hops [sr nb streams] [queues]"""
import json, os, sys, threading, time
from pathlib import Path
hops = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
sr, nb, S = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (16000, 2, 1)
os.environ["GPU_MAX_HW_QUEUES"] = sys.argv[5] if len(sys.argv) > 5 else "4"      # before HIP comes up; the package's default of 8 is NOT applied
import numpy as np
import torch
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
from dpdfnet_amd import backend as be
from dpdfnet_amd.weights import synth_blob


def session(n_hops, fingerprints):
    m = be.HipModel(sr, nb, synth_blob(be.manifest(sr, nb), 20260417), 0)
    st = be.HipStreams(m, S)
    r = np.random.default_rng(7)
    pcm = (0.05 * r.standard_normal((64, S, m.hop))).astype(np.float32)
    st.prime(pcm[0])
    t0 = time.perf_counter()
    for i in range(n_hops):
        y = st.process(pcm[i & 63])
        fingerprints[i] = float(y[S // 2, 7]) + float(np.abs(y[0]).sum())
    dt = time.perf_counter() - t0
    rec = int(m.recovery_count)
    st.close(); m.close()
    return rec, dt


def run():
    torch.cuda.set_device(0)
    alone = np.zeros(hops); rec0, dt0 = session(hops, alone)
    stop = threading.Event()
    launched = [0]

    def tenant():
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        a = torch.randn(2048, 2048, device="cuda"); b = torch.randn(2048, 2048, device="cuda")
        x = torch.randn(1 << 22, device="cuda")
        while not stop.is_set():
            with torch.cuda.stream(s1):
                c = a @ b
            with torch.cuda.stream(s2):
                y = torch.tanh(x) * 0.5 + x
            launched[0] += 2
            if launched[0] % 64 == 0:
                s1.synchronize(); s2.synchronize()            # (keeps the queues full without unbounded backlog)
        torch.cuda.synchronize()

    th = threading.Thread(target=tenant); th.start()
    time.sleep(0.2)
    shared = np.zeros(hops); rec1, dt1 = session(hops, shared)
    stop.set(); th.join()
    diff = int(np.count_nonzero(shared != alone))
    return {"FAIL": bool(rec0 or rec1 or diff), "hops": hops, "model": f"{sr}/nb{nb}/S{S}", "GPU_MAX_HW_QUEUES": os.environ["GPU_MAX_HW_QUEUES"],
            "recoveries_alone": rec0, "recoveries_beside_torch": rec1, "hops_differing_from_the_run_alone": diff,
            "us_per_hop_alone": round(1e6 * dt0 / hops, 1), "us_per_hop_beside_torch": round(1e6 * dt1 / hops, 1), "torch_kernels_launched": launched[0]}


if __name__ == "__main__":
    print(json.dumps(run()))
