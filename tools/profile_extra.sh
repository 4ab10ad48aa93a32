#!/bin/bash
# rocprofv3 evidence for the configurations beside the headline (round-3 review item 7):
#   * single streaming hops (config 5: 64 x dpdfnet8_48khz_hr; one dpdfnet2 16 kHz stream): kernel stats + a per-hop timeline
#   * offline 256 x 10 s steps of dpdfnet2 / dpdfnet8 (16 kHz) and dpdfnet2_48khz_hr / dpdfnet8_48khz_hr: kernel stats
# usage (on the GPU box): tools/profile_extra.sh <tag>      -> gpurun_out/prof_<tag>_extra/
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-rX}
OUT=$R/gpurun_out/prof_${TAG}_extra
rm -rf $OUT; mkdir -p $OUT
cd $R
for cfg in "48000 8 64 hop_48k_nb8_64streams" "16000 2 1 hop_16k_nb2_1stream"; do
    set -- $cfg
    D=$OUT/$4
    rocprofv3 --kernel-trace --stats --output-format csv -d $D -o t -- python tools/hop_trace.py run $1 $2 $3 > $D.log 2>&1
    grep "wall us/hop" $D.log
    f=$(find $D -name '*kernel_trace.csv' | head -1)
    python tools/hop_timeline.py $f > $OUT/$4_timeline.txt
    cp $(find $D -name '*kernel_stats.csv' | head -1) $OUT/$4_kernel_stats.csv
    rm -rf $D
done
cat > /tmp/offline_one.py <<'PY'
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from dpdfnet_amd import backend as be
from dpdfnet_amd.weights import synth_blob
sr, nb, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
m = be.HipModel(sr, nb, synth_blob(be.manifest(sr, nb), 20260417), 0)
N = 10 * sr
wav = torch.from_numpy((0.05 * np.random.default_rng(1).standard_normal((B, N))).astype(np.float32)).cuda()
out = torch.empty_like(wav)
m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None); m.sync()
t0 = time.perf_counter()
for _ in range(2): m.enhance_batch_device(wav.data_ptr(), B, N, out.data_ptr(), None)
m.sync()
print("ms/step", 1e3 * (time.perf_counter() - t0) / 2, "frames/s", B * m.num_frames(N) * 2 / (time.perf_counter() - t0))
PY
for cfg in "16000 2 offline_16k_nb2_256x10s" "16000 8 offline_16k_nb8_256x10s" "48000 2 offline_48k_nb2_256x10s" "48000 8 offline_48k_nb8_256x10s"; do
    set -- $cfg
    D=$OUT/$3
    rocprofv3 --kernel-trace --stats --output-format csv -d $D -o t -- python /tmp/offline_one.py $1 $2 256 > $D.log 2>&1
    grep "ms/step" $D.log
    cp $(find $D -name '*kernel_stats.csv' | head -1) $OUT/$3_kernel_stats.csv
    rm -rf $D
done
ls -la $OUT
