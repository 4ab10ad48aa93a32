#!/bin/bash
# A/B of several library builds on the latency-regime side configurations, each measurement in its own process, interleaved.
# usage: tools/libs_ab.sh "<cfg> <cfg> ..." a.so b.so ...     (cfg: streams48 streams16 one_clip ...)
CFGS=$1; shift
for rep in 1 2 3; do
  for cfg in $CFGS; do
    line="$cfg"
    for so in "$@"; do
      v=$(DPDFNET_HIP_LIB=$(realpath $so) python bench.py --side-config $cfg 2>/dev/null | grep '^SIDE' | python -c "import sys,json; d=json.loads(sys.stdin.read()[5:]); print(d.get('us_per_call', d.get('ms_per_step', d.get('ms_per_call'))))")
      line="$line  $(basename $so) $v"
    done
    echo "$line"
  done
done
