#!/bin/bash
# Timing ablations of gru256_clusterN_kernel (G256_VARIANT bits, gru_scan.h): builds one library per variant on the GPU
# box and reads the serial gru256_scan class time of the headline workload.  Results of variants != 0 are wrong by design.
cd ${GRAFT_REPO_ROOT:-.}
for v in "$@"; do
  lib=/tmp/libdpdf_v$v.so
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DG256_VARIANT=$v $XDEF -o $lib dpdfnet_amd/csrc/dpdf_model.hip 2>/dev/null || { echo "build $v failed"; continue; }
  DPDFNET_HIP_LIB=$lib python bench.py --no-cpu-baseline --no-other-configs --no-pcie --no-isolated --overlap 0 --steps 2 --opt gru256_pair=${PAIR:-4} 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][0])
print('variant $v pair ${PAIR:-4}: gru256_scan %.2f ms/step (serial total %.1f)' % (d['roofline']['per_class_ms_per_step']['gru256_scan'], d['ms_per_step']))
"
done
