#!/bin/bash
set -u
export DPDFNET_HIP_LIB=$PWD/build_ab/lib_probe.so
O=gpurun_out/hazard; mkdir -p $O
R=${RUNS:-8}
run() { name=$1; shift; echo "== $name: $*"; ( time timeout 900 "$@" ) > $O/$name.txt 2>&1; grep -v "^  taps" $O/$name.txt | cut -c1-1200 | grep -v "frame pos\|chunk index\|bins f\|history" | head -80; }
run 61_scalar_twins   python tools/hazard_probe.py $R 62,a2,b2,92 0 3 1
