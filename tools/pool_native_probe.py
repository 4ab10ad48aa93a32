"""The library's stream pool fed by NATIVE threads (tools/pool_native_feeders.cpp): rounds against the number of feeder threads and the
polling time.  usage: python tools/pool_native_probe.py [S=64]"""
import os, sys, ctypes, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from dpdfnet_amd import StreamEnhancer
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sr, hop = 48000, 480
kw = dict(model="dpdfnet8_48khz_hr", onnx_path=f"synthetic:{bench.WEIGHT_SEED}")
pcm = (0.05 * np.random.default_rng(1).standard_normal((S, hop))).astype(np.float32)
helper = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libpool_native_feeders.so"))
helper.pool_native_feeders.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]
pool = StreamEnhancer.pool(S, **kw)
members = [pool.enhancer() for _ in range(S)]
pool.process_many([(m_, np.concatenate([pcm[i], pcm[i]])) for i, m_ in enumerate(members)])
L = pool._streams.model._L
fn = ctypes.cast(L.dpdf_streams_submit_block, ctypes.c_void_p)
us = ctypes.c_double(0.0)
g_in = pcm.copy(); 
for spin in (1e-3, 0.0):
    pool._streams.pool_tune(2e-4, 2e-3, spin)
    for nt in (1, 2, 4, 8, 16, 64):
        if nt > S: continue
        res = []
        for rep in range(3):
            dc = pool.device_calls; t0 = pool._streams.pool_timing()
            helper.pool_native_feeders(pool._streams._h, fn, S, nt, 300, hop, pcm.ctypes.data, None, ctypes.byref(us))
            t1 = pool._streams.pool_timing()
            res.append((round(us.value, 1), round((pool.device_calls - dc) / 300, 2), "wait/call/gap us", [round(1e6 * (b - a) / 300, 1) for a, b in zip(t0, t1)]))
        print(f"spin {spin:g} threads {nt:3d}: us/round, calls/round {res}", flush=True)
# the plain lock-step call from this thread, same process
st = pool._streams
t0 = time.perf_counter()
for _ in range(300): st.process(pcm)
print("plain process() from python, us", round(1e6 * (time.perf_counter() - t0) / 300, 1))
