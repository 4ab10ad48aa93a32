#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per kernel for the default bench (quick check of HBM traffic per launch)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/pmc_fetch_only; rm -rf $OUT; mkdir -p $OUT; cd $R
CMD="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-isolated"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/f -o pmc -- $CMD > $OUT/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/w -o pmc -- $CMD > $OUT/w.log 2>&1
python - <<PY
import csv, collections
def agg(path):
    a = collections.defaultdict(float); n = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0][:64]; a[k] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    return a, n
f, nf = agg("$OUT/f/pmc_counter_collection.csv"); w, nw = agg("$OUT/w/pmc_counter_collection.csv")
for k in sorted(f, key=lambda k: -f[k])[:14]:
    d = len(nf[k]); print("%-66s n=%4d  read(x2) %8.1f MB  write %8.1f MB per launch" % (k, d, 2 * f[k] / d / 1024, w.get(k, 0) / max(1, len(nw.get(k, [1]))) / 1024))
PY
