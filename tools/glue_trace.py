"""Phase timing INSIDE dprnn_hop_glue8_kernel<true> (s_memtime stamps of workgroup 0 at entry, behind every barrier and at exit).
Needs the library built with -DDPDF_PHASE_TRACE (not the shipped build):
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DDPDF_PHASE_TRACE -o dpdfnet_amd/libdpdfnet_hip.so dpdfnet_amd/csrc/dpdf_model.hip -Iinclude
then `python tools/glue_trace.py` on the GPU, then rebuild normally (python -c "import __graft_entry__ as g; g.build()").
With two branches running side by side (>1 launch in flight) stamps of different launches mix; the one-stream line is the clean one."""
import sys, ctypes, numpy as np
sys.path.insert(0, ".")
from dpdfnet_amd import backend as be
from dpdfnet_amd.weights import synth_blob
for sr, nb, S in ((16000, 2, 1), (48000, 8, 64)):
    m = be.HipModel(sr, nb, synth_blob(be.manifest(sr, nb), 1), 0)
    st = be.HipStreams(m, S)
    rng = np.random.default_rng(0)
    st.prime((0.05 * rng.standard_normal((S, m.hop))).astype(np.float32))
    pcm = (0.05 * rng.standard_normal((S, m.hop))).astype(np.float32)
    acc = None
    for i in range(30):
        st.process(pcm)
        buf = (ctypes.c_ulonglong * 32)()
        m._L.dpdf_debug_trace.argtypes = [ctypes.c_void_p]
        m._L.dpdf_debug_trace(buf)
        t = np.array(buf[:10], dtype=np.int64)
        t10 = int(buf[10]) - int(buf[8])
        ts = np.array(buf[16:20], dtype=np.int64)      # scan role of the one-launch form (dprnn_hop_block.h): entry, first step, behind the last step, flag out
        d = np.concatenate([np.diff(t), np.diff(ts), [t10]])
        if i >= 10: acc = d if acc is None else acc + d
    a = (acc / 20).astype(int).tolist()
    print(sr, nb, S, "s_memtime ticks (~shader clock; counters of different XCDs are not comparable) between stamps, avg of 20 hops, last <true> launch of the DF stack:")
    print("   glue tile 0: entry -> operands in + rows in -> fc_intra -> LN -> ... -> exit:", a[:9])
    print("   scan (0, fwd): entry -> first step -> behind the last step -> flag out:", a[9:12])
    print("   glue: stamp 8 -> matrix part of the next projection done:", a[12])
    st.close(); m.close()
