#!/bin/bash
# round 6: (1) the whole GPU suite, default engine; (2) the whole suite with every handle in the bf16-limb mode; (3) bench line
mkdir -p gpurun_out/r6c
( time timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 ) > gpurun_out/r6c/gpu_suite_default.txt 2>&1; tail -6 gpurun_out/r6c/gpu_suite_default.txt
( time DPDF_GRU64_LIMBS=3 timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -25 ) > gpurun_out/r6c/gpu_suite_limbs.txt 2>&1; tail -12 gpurun_out/r6c/gpu_suite_limbs.txt
timeout 900 python bench.py > gpurun_out/r6c/bench_line.json 2> gpurun_out/r6c/bench_err.txt; python tools/show_line.py gpurun_out/r6c/bench_line.json 2>/dev/null | head -40 || head -c 1500 gpurun_out/r6c/bench_line.json
tools/gru64_limb_bench | tail -12 > gpurun_out/r6c/gru64_limb_bench.txt; cat gpurun_out/r6c/gru64_limb_bench.txt
