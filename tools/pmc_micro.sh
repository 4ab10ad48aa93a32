#!/bin/bash
# SQ counters per kernel: tools/pmc_micro.sh "<command relative to the repo root>" [kernel-name substring ...]
cd /tmp && export TMPDIR=/tmp
CMD="$1"; shift; PATS="${*:-gru64 mfma}"; cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_micro; rm -rf $OUT; mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA" \
           "SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC" \
           "SQ_IFETCH SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -o p -- $CMD > $OUT/g$i.log 2>&1 || tail -3 $OUT/g$i.log
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob("$OUT/g*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        pat = next((p for p in "$PATS".split() if p in r["Kernel_Name"]), None)
        if pat is None: continue
        agg[(pat, r["Counter_Name"])] += float(r["Counter_Value"]); n[(pat, r["Counter_Name"])] += 1
for k in sorted(agg): print("%-14s %-28s %16.0f  per dispatch %14.0f" % (k[0], k[1], agg[k], agg[k] / max(1, n[k])))
PY
