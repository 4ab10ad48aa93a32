#!/bin/bash
set -u
export DPDFNET_HIP_LIB=$PWD/build_ab/lib_probe.so
O=gpurun_out/hazard; mkdir -p $O
run() { name=$1; shift; echo "== $name: $*"; ( time timeout 900 "$@" ) > $O/$name.txt 2>&1; grep -v "^  taps" $O/$name.txt | cut -c1-400 | grep -v "frame pos\|chunk index\|bins f\|history\|consumer\|     e.g\|^        " | head -60; }
run 71_coissue_standalone  tools/pk_fma_coissue_probe 4
PROBE_EXPLAIN=0 run 72_twins_fp32_stage1   python tools/hazard_probe.py 30 b2 0 0 1
