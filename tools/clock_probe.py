"""Does a latency-bound call slow down right after a long throughput run (clock / power management)?"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from dpdfnet_amd import backend as be
from dpdfnet_amd.weights import synth_blob
sr, nb = 16000, 4
blob = synth_blob(be.manifest(sr, nb), 20260417)
big = be.HipModel(sr, nb, blob, 0)
N = 10 * sr
wb = torch.from_numpy((0.05 * np.random.default_rng(1).standard_normal((256, N))).astype(np.float32)).cuda(); ob = torch.empty_like(wb)
def small(tag):
    m = be.HipModel(sr, nb, blob, 0)
    w1 = wb[:1].contiguous(); o1 = torch.empty_like(w1)
    m.enhance_batch_device(w1.data_ptr(), 1, N, o1.data_ptr(), None); m.sync()
    ts = []
    for _ in range(6):
        t0 = time.perf_counter(); m.enhance_batch_device(w1.data_ptr(), 1, N, o1.data_ptr(), None); m.sync(); ts.append((time.perf_counter() - t0) * 1e3)
    print(tag, " ".join(f"{t:.2f}" for t in ts)); m.close()
small("cold, big handle idle:      ")
for _ in range(6): big.enhance_batch_device(wb.data_ptr(), 256, N, ob.data_ptr(), None)
big.sync()
small("right after 0.65 s of load: ")
time.sleep(2.0)
small("after 2 s idle:             ")
big.close()
small("big handle closed:          ")
