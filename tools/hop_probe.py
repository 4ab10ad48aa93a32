"""Where does a streaming hop's wall time go?  Host-pointer call (what StreamEnhancer does) vs device-pointer call split into
host enqueue time and time until the GPU is done."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
import torch
from dpdfnet_amd import backend as be
from dpdfnet_amd.weights import synth_blob
for sr, nb, S in ((48000, 8, 64), (16000, 2, 1), (16000, 4, 8)):
    m = be.HipModel(sr, nb, synth_blob(be.manifest(sr, nb), 20260417), 0)
    for kv in sys.argv[1:]:
        k, v = kv.split("="); m.set_option(k, int(v))
    st = be.HipStreams(m, S)
    rng = np.random.default_rng(0)
    st.prime((0.05 * rng.standard_normal((S, m.hop))).astype(np.float32))
    pcm = (0.05 * rng.standard_normal((S, m.hop))).astype(np.float32)
    for _ in range(30): st.process(pcm)
    t0 = time.perf_counter()
    for _ in range(200): st.process(pcm)
    host_us = 1e6 * (time.perf_counter() - t0) / 200
    d_in = torch.from_numpy(pcm).cuda(); d_out = torch.empty_like(d_in)
    L = m._L
    enq, tot = [], []
    for _ in range(100):
        t0 = time.perf_counter()
        rc = L.dpdf_streams_process(st._h, d_in.data_ptr(), 1, d_out.data_ptr(), be.DPDF_DEVICE_PTRS)
        t1 = time.perf_counter(); m.sync(); t2 = time.perf_counter()
        assert rc == 0
        enq.append(1e6 * (t1 - t0)); tot.append(1e6 * (t2 - t0))
    print(f"sr {sr} nb {nb} streams {S}: host-pointer call {host_us:.0f} us/hop; device-pointer call: host enqueue {np.median(enq):.0f} us, until done {np.median(tot):.0f} us")
    st.close(); m.close()
