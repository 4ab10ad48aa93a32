#!/bin/bash
set -u
export DPDFNET_HIP_LIB=$PWD/build_ab/lib_probe.so
O=gpurun_out/hazard; mkdir -p $O
R=${RUNS:-6}
run() { name=$1; shift; echo "== $name: $*"; ( time timeout 900 "$@" ) > $O/$name.txt 2>&1; grep -v "^  taps" $O/$name.txt | head -60; }
run 31_snapshot   python tools/hazard_probe.py $R 2,32,34,102,104,12,14,22,4 0 3 1
