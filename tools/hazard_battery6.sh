#!/bin/bash
set -u
export DPDFNET_HIP_LIB=$PWD/build_ab/lib_probe.so
O=gpurun_out/hazard; mkdir -p $O
R=${RUNS:-8}
run() { name=$1; shift; echo "== $name: $*"; ( time timeout 900 "$@" ) > $O/$name.txt 2>&1; grep -v "^  taps" $O/$name.txt | cut -c1-900 | grep -v "frame pos\|chunk index\|bins f\|history" | head -60; }
run 51_exact_stream   python tools/hazard_probe.py $R 62,72,82,92,2 0 3 1
