"""Timeline INSIDE dprnn_hop_stack_kernel (DF stack, stream group 0) from s_memrealtime stamps (100 MHz): needs a -DDPDF_PHASE_TRACE build
(DPDFNET_HIP_LIB=build_ab/lib_trace.so).  usage: python tools/stack_trace.py [sr nb S]"""
import sys, os, ctypes, numpy as np
sys.path.insert(0, ".")
from dpdfnet_amd import backend as be
from dpdfnet_amd.weights import synth_blob
sr, nb, S = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (48000, 8, 64)
m = be.HipModel(sr, nb, synth_blob(be.manifest(sr, nb), 1), 0)
st = be.HipStreams(m, S)
rng = np.random.default_rng(0)
st.prime((0.05 * rng.standard_normal((S, m.hop))).astype(np.float32))
pcm = (0.05 * rng.standard_normal((S, m.hop))).astype(np.float32)
m._L.dpdf_debug_stack_trace.argtypes = [ctypes.c_void_p]
for i in range(25):
    st.process(pcm)
buf = (ctypes.c_ulonglong * 512)()
m._L.dpdf_debug_stack_trace(buf)
t = np.array(buf[:], dtype=np.int64)
t0 = t[1]                                   # forward scan, block 0, first step
us = lambda v: (v - t0) / 100.0
print(f"{sr} Hz nb {nb}, {S} streams: one hop, microseconds from the first step of block 0's forward scan")
for n in range(nb):
    f = [us(t[4 * n + e]) for e in range(3)]; b = [us(t[64 + 4 * n + e]) for e in range(3)]
    line = f"block {n}: fwd scan flags {f[0]:7.1f} first {f[1]:7.1f} last {f[2]:7.1f} | bwd {b[0]:7.1f} {b[1]:7.1f} {b[2]:7.1f} | glue s0:"
    for k in range(3):
        g = [us(t[128 + 16 * n + 4 * k + e]) for e in range(4)]
        line += f"  tile{k} in {g[0]:6.1f} valid {g[1]:6.1f} done {g[2]:6.1f} flag {g[3]:6.1f}"
    print(line)
st.close(); m.close()
