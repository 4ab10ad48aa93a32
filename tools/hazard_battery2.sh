#!/bin/bash
set -u
export DPDFNET_HIP_LIB=$PWD/build_ab/lib_probe.so
O=gpurun_out/hazard; mkdir -p $O
R=${RUNS:-6}
run() { name=$1; shift; echo "== $name: $*"; ( time timeout 900 "$@" ) > $O/$name.txt 2>&1; tail -12 $O/$name.txt; }
run 11_dump_xm_modes   python tools/hazard_probe.py $R 2,4,3,1 0 3 1
PROBE_REF_TAPS=6 run 12_no_tap_loads python tools/hazard_probe.py $R 6 0 3 1
run 13_fp32_kernels_mode2 python tools/hazard_probe.py $R 2 0 0 1
