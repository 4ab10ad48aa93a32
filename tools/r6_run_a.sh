#!/bin/bash
# round 6, first look at the persistent DPRNN stack: the hop-forms test, the non-finite test, the two streaming side configurations with and without it
mkdir -p gpurun_out/r6a
( timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "hop_forms or nonfinite" 2>&1 | tail -25 ) > gpurun_out/r6a/hop_forms.txt; cat gpurun_out/r6a/hop_forms.txt
for cfg in streams48 streams16; do
  timeout 300 python bench.py --side-config $cfg 2>&1 | grep "^SIDE" | cut -c1-400
done
