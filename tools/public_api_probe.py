"""The public API alone in a fresh process: dpdfnet_amd.enhance_batch(list of 256 numpy clips x 10 s, 16000, model="dpdfnet4").
usage: python tools/public_api_probe.py [hw_queues]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    os.environ["GPU_MAX_HW_QUEUES"] = sys.argv[1]
os.environ["DPDF_HOST_PIPE_TRACE"] = "1"
import numpy as np
import dpdfnet_amd
from bench import synth_clips, WEIGHT_SEED, SR, MODEL
B, N = 256, 160000
clips = [c.copy() for c in synth_clips(B, N, SR, WEIGHT_SEED)]
kw = dict(model=MODEL, onnx_path=f"synthetic:{WEIGHT_SEED}")
dpdfnet_amd.enhance_batch(clips, SR, **kw)
for rep in range(2):
    t0 = time.perf_counter()
    for _ in range(3):
        outs = dpdfnet_amd.enhance_batch(clips, SR, **kw)
    print(f"public enhance_batch (fresh process, GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')}) {1e3 * (time.perf_counter() - t0) / 3:8.2f} ms", flush=True)
