#!/bin/bash
# kernel trace of config 5's hop + timeline of one hop (profiles/r6_stream_hop_timeline.txt)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/stream_prof; rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o sp -- python tools/stream_bench.py --one ${STREAM_ARGS:-} > $OUT/log.txt 2>&1
tail -3 $OUT/log.txt
f=$(find $OUT -name '*kernel_trace.csv' | head -1)
python tools/hop_timeline.py $f > $OUT/timeline.txt 2>&1; head -70 $OUT/timeline.txt
