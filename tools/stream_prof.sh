#!/bin/bash
# kernel trace of the single-hop streaming path (config 5): launches per hop and busy time vs wall time
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/stream_prof; rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o sp -- python tools/stream_bench.py --one > $OUT/log.txt 2>&1
tail -3 $OUT/log.txt
f=$(find $OUT -name '*kernel_stats.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows); calls = sum(int(r["Calls"]) for r in rows)
print("kernels", calls, "busy_ms", tot / 1e6)
for r in rows[:25]:
    print(r["Name"][:70], r["Calls"], r["TotalDurationNs"], r["AverageNs"])
PY
