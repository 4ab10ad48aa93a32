"""Cycle accounting of gru256_clusterN_kernel (build with -DG256_VARIANT=256): step / sweep-finish / barrier cycles of WG 0."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from dpdfnet_amd import backend
from dpdfnet_amd.weights import synth_blob
m = backend.HipModel(16000, 4, synth_blob(backend.manifest(16000, 4), 1), 0)
m.set_overlap(0)
m.set_option("gru256_pair", int(sys.argv[1]) if len(sys.argv) > 1 else 4)
wav = (0.05 * np.random.default_rng(0).standard_normal((256, 32000))).astype(np.float32)
m.enhance_batch(wav)
L = backend.load_library()
buf = (ctypes.c_ulonglong * 8)()
L.dpdf_debug_g256(buf)
b0 = list(buf)
m.enhance_batch(wav)
L.dpdf_debug_g256(buf)
d = [x - y for x, y in zip(buf, b0)]
nb = max(1, d[4])
print("blocks", d[4], "cycles/block: step %.0f  finish+issue %.0f  barrier %.0f   re-sweeps/block %.3f" % (d[0] / nb, d[1] / nb, d[2] / nb, d[3] / nb))
