#!/bin/bash
# What does a kernel class COST in the pipeline?  Offline 256 x 10 s step with a debug build that can leave out launches
# (getenv switches patched into a copy of dpdf_model.hip; results are garbage; timing only).
# usage: tools/skip_probe.sh <debug.so> sr nb
O=$(realpath $1); SR=$2; NB=$3
cat > /tmp/skip_one.py <<'PY'
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from dpdfnet_amd import backend as be
from dpdfnet_amd.weights import synth_blob
sr, nb, B = int(sys.argv[1]), int(sys.argv[2]), 256
m = be.HipModel(sr, nb, synth_blob(be.manifest(sr, nb), 20260417), 0)
N = 10 * sr
wav = torch.from_numpy((0.05 * np.random.default_rng(1).standard_normal((B, N))).astype(np.float32)).cuda()
out = torch.empty_like(wav)
m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None); m.sync()
t0 = time.perf_counter()
for _ in range(3): m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None)
m.sync()
print("ms/step %.2f" % (1e3 * (time.perf_counter() - t0) / 3))
PY
for rep in 1 2; do
  echo "full            $(DPDFNET_HIP_LIB=$O python /tmp/skip_one.py $SR $NB 2>/dev/null | tail -1)"
  for v in SKIP_CONV0 SKIP_DEC SKIP_PROJ SKIP_SCAN256 SKIP_DPRNN_ERB SKIP_DPRNN_DF; do
    echo "no $v   $(env DPDFNET_HIP_LIB=$O DPDF_DEBUG_$v=1 python /tmp/skip_one.py $SR $NB 2>/dev/null | tail -1)"
  done
done
