"""One configuration's 256 x 10 s step with the stages back to back on one stream (overlap 0), for a kernel trace of what every kernel costs
alone: rocprofv3 --kernel-trace --stats -d DIR -o t -- python tools/serial_step.py SR NB [steps]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from dpdfnet_amd import backend as be
from dpdfnet_amd.weights import synth_blob
sr, nb, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 2
B, N = 256, 10 * sr
m = be.HipModel(sr, nb, synth_blob(be.manifest(sr, nb), 20260417), 0)
m.set_overlap(0)
wav = torch.from_numpy((0.05 * np.random.default_rng(1).standard_normal((B, N))).astype(np.float32)).cuda()
out = torch.empty_like(wav)
m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None); m.sync()
t0 = time.perf_counter()
for _ in range(steps): m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None)
m.sync()
print("serial ms/step", 1e3 * (time.perf_counter() - t0) / steps, "calls traced", steps + 1)
