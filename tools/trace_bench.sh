#!/bin/bash
# kernel trace (begin/end timestamps) of a short bench run, for timeline analysis with tools/timeline.py
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_bench; rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o tb -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-isolated "$@" > $OUT/log.txt 2>&1
grep '^{"metric"' $OUT/log.txt | cut -c1-300
ls -la $OUT
