"""Where a pipelined host-pointer batch call spends its time (csrc/dpdf_model.hip enhance_impl): C-ABI call on a numpy block,
the row-pointer form on a list of arrays, and the public API, 256 clips x 10 s dpdfnet4, with the library's host-thread trace
(DPDF_HOST_PIPE_TRACE) and a few copy-thread counts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ["DPDF_HOST_PIPE_TRACE"] = "1"
import numpy as np
import torch  # noqa: F401  (device buffers for the HBM-resident reference point)
from bench import synth_clips, WEIGHT_SEED, SR, NB, MODEL
from dpdfnet_amd import backend
from dpdfnet_amd.weights import synth_blob

B, N = int(os.environ.get("CLIPS", "256")), 160000
wav = synth_clips(B, N, SR, WEIGHT_SEED)
m = backend.HipModel(SR, NB, synth_blob(backend.manifest(SR, NB), WEIGHT_SEED), device=0)
d_in = torch.from_numpy(wav).cuda(); d_out = torch.empty_like(d_in)


def t(fn, reps=3):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    return 1e3 * (time.perf_counter() - t0) / reps, r


ms, _ = t(lambda: (m.enhance_batch_device(d_in.data_ptr(), B, N, d_out.data_ptr(), None), m.sync()))
print(f"HBM-resident            {ms:8.2f} ms", flush=True)
out_fixed = np.empty_like(wav)
import ctypes
def call_fixed():
    backend._check(m._L.dpdf_enhance_batch(m._h, wav.ctypes.data, B, N, float("nan"), out_fixed.ctypes.data, 0))
for thr in (4, 1, 2, 8):
    m.set_option("host_copy_threads", thr)
    ms, _ = t(call_fixed)
    print(f"host block, reused out buffer, {thr} copy threads {ms:8.2f} ms", flush=True)
m.set_option("host_copy_threads", 4)
def call_fresh():
    o = np.empty_like(wav)
    backend._check(m._L.dpdf_enhance_batch(m._h, wav.ctypes.data, B, N, float("nan"), o.ctypes.data, 0))
    return o
for pf in (1, 0):
    ms, _ = t(call_fresh)
    print(f"host block, fresh np.empty out, prefault {pf} {ms:8.2f} ms", flush=True)
ms, _ = t(lambda: m.enhance_batch(wav))
print(f"host block, HipModel.enhance_batch (leased out block) {ms:8.2f} ms", flush=True)
clips = [c.copy() for c in wav]
ms, _ = t(lambda: m.enhance_batch_ragged(clips))
print(f"row pointers, list in / list out {ms:8.2f} ms", flush=True)
m.set_option("host_pipe", 0)
ms, _ = t(call_fixed)
print(f"host block, pipeline OFF        {ms:8.2f} ms", flush=True)
m.close()
import dpdfnet_amd
kw = dict(model=MODEL, onnx_path=f"synthetic:{WEIGHT_SEED}")
ms, _ = t(lambda: dpdfnet_amd.enhance_batch(clips, SR, **kw))
print(f"public enhance_batch            {ms:8.2f} ms", flush=True)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); dpdfnet_amd.enhance_batch(clips, SR, **kw); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
