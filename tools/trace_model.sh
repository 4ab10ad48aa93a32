#!/bin/bash
# usage: tools/trace_model.sh <nb> <B> [sr]  -> gpurun_out/trace_model/tm_kernel_trace.csv (analyse with tools/timeline.py)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_model; rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o tm -- python tools/trace_model.py "$@" > $OUT/log.txt 2>&1
python tools/timeline.py $OUT/tm_kernel_trace.csv
