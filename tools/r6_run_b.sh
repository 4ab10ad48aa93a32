#!/bin/bash
DPDFNET_HIP_LIB=$PWD/build_ab/lib_trace.so timeout 300 python tools/stack_trace.py 2>&1 | grep -v amdgpu.ids | cut -c1-330
( timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "hop_forms" 2>&1 | tail -5 )
for cfg in streams48 streams16; do timeout 300 python bench.py --side-config $cfg 2>&1 | grep "^SIDE" | cut -c1-200; done
