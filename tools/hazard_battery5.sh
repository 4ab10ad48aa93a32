#!/bin/bash
set -u
export DPDFNET_HIP_LIB=$PWD/build_ab/lib_probe.so
O=gpurun_out/hazard; mkdir -p $O
R=${RUNS:-8}
run() { name=$1; shift; echo "== $name: $*"; ( time timeout 900 "$@" ) > $O/$name.txt 2>&1; grep -v "^  taps" $O/$name.txt | cut -c1-600 | head -60; }
run 41_overlap_shape   python tools/hazard_probe.py $R 42,52,44,54,43,32,2 0 3 1
rocminfo | grep -i -E "xnack|Name:.*gfx" | head -5 >> $O/41_overlap_shape.txt; tail -3 $O/41_overlap_shape.txt
