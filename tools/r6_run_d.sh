#!/bin/bash
mkdir -p gpurun_out/r6d
( time timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 ) > gpurun_out/r6d/gpu_suite_default.txt 2>&1; tail -6 gpurun_out/r6d/gpu_suite_default.txt
( DPDF_GRU64_LIMBS=3 timeout 900 python -m pytest tests/test_gpu_bench_flow.py -q -m gpu -k contract 2>&1 | tail -5 ) > gpurun_out/r6d/bench_flow_limbs_env.txt 2>&1; tail -3 gpurun_out/r6d/bench_flow_limbs_env.txt
