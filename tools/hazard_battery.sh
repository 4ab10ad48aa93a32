#!/bin/bash
# the probe battery (tools/hazard_probe.py) on the probe build.  Output: gpurun_out/hazard/*.txt
set -u
export DPDFNET_HIP_LIB=$PWD/build_ab/lib_probe.so
O=gpurun_out/hazard; mkdir -p $O
R=${RUNS:-20}
run() { name=$1; shift; echo "== $name: $*"; ( time timeout 600 "$@" ) > $O/$name.txt 2>&1; tail -4 $O/$name.txt; }
run 01_repro_nodump            python tools/hazard_probe.py $R 1 0 3 0
run 02_dump_modes              python tools/hazard_probe.py $R 1,4,5,2,3,0 0 3 1
run 03_uncached                python tools/hazard_probe.py $R 1,4 1 3 1
AMD_OPT_FLUSH=0 run 04_optflush0 python tools/hazard_probe.py $R 1 0 3 1
GPU_MAX_HW_QUEUES=8 run 05_hwq8 python tools/hazard_probe.py $R 1 0 3 1
run 06_fp32_merged_soak        python tools/hazard_probe.py ${SOAK:-150} 1 0 0 0
