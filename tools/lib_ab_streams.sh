#!/bin/bash
# A/B of two library builds on the streaming side configurations, each measurement in its own process, interleaved.
# usage: tools/lib_ab_streams.sh <other.so>
O=$(realpath $1)
for rep in 1 2 3; do
  for cfg in streams48 streams16; do
    a=$(python bench.py --side-config $cfg 2>/dev/null | grep '^SIDE' | python -c "import sys,json; print(json.loads(sys.stdin.read()[5:])['us_per_call'])")
    b=$(DPDFNET_HIP_LIB=$O python bench.py --side-config $cfg 2>/dev/null | grep '^SIDE' | python -c "import sys,json; print(json.loads(sys.stdin.read()[5:])['us_per_call'])")
    echo "$cfg tree $a other $b"
  done
done
