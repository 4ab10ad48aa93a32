#!/bin/bash
# Collect the rocprofv3 evidence for profiles/: kernel-trace stats + separate PMC passes.
# usage (on the GPU box): tools/profile_round.sh <tag>
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-rX}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
python $R/tools/build_manifest.py > $OUT/build_manifest.json        # what exactly was measured (checked by tests/test_abi_and_layout.py)
# Every launch of a traced run has ONE execution shape (no appended serial step, no side configs), so that the CSV's
# AverageNs of a kernel IS the bench line's roofline.avg_launch_ms of the same run:
#   trace_pipelined: the default 4-stream pipeline  -> roofline.frac
#   trace_serial:    --overlap 0, kernels back to back -> roofline.frac_isolated of the default run
CMD="timeout 600 python bench.py --steps 3 --warmup 1 --no-isolated --no-other-configs --no-cpu-baseline --no-pcie --no-dist-selftest"
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_pipelined -o bench -- $CMD > $OUT/bench_trace_pipelined.log 2>&1
grep '^{"metric"' $OUT/bench_trace_pipelined.log > $OUT/bench_line_under_trace_pipelined.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_serial -o bench -- $CMD --overlap 0 > $OUT/bench_trace_serial.log 2>&1
grep '^{"metric"' $OUT/bench_trace_serial.log > $OUT/bench_line_under_trace_serial.json
# the fp32-MFMA GRU-64 kernels (bench.py --fp32-mfma: the default of rounds 1-5, an A/B mode since), the same two execution shapes
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_fp32mfma_pipelined -o bench -- $CMD --fp32-mfma > $OUT/bench_trace_fp32mfma_pipelined.log 2>&1
grep '^{"metric"' $OUT/bench_trace_fp32mfma_pipelined.log > $OUT/bench_line_under_trace_fp32mfma_pipelined.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_fp32mfma_serial -o bench -- $CMD --fp32-mfma --overlap 0 > $OUT/bench_trace_fp32mfma_serial.log 2>&1
grep '^{"metric"' $OUT/bench_trace_fp32mfma_serial.log > $OUT/bench_line_under_trace_fp32mfma_serial.json
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o pmc -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o pmc -- $CMD > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace --output-format csv -d $OUT/pmc_mfma -o pmc -- $CMD > $OUT/pmc_mfma.log 2>&1
python - <<PY
import csv, collections, json, sys
out = "$OUT"
def agg(path, counters):
    rows = list(csv.DictReader(open(path)))
    a = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for r in rows:
        k = r["Kernel_Name"].split("(")[0][:70]
        a[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    return a, {k: len(v) for k, v in n.items()}
res = {}
f, nf = agg(out + "/pmc_fetch/pmc_counter_collection.csv", ["FETCH_SIZE"])
w, nw = agg(out + "/pmc_write/pmc_counter_collection.csv", ["WRITE_SIZE"])
mm, nm = agg(out + "/pmc_mfma/pmc_counter_collection.csv", [])
for k in sorted(set(f) | set(w) | set(mm)):
    e = {"dispatches": nf.get(k, nw.get(k, nm.get(k, 0)))}
    if k in f: e["FETCH_SIZE_KB_sum"] = f[k]["FETCH_SIZE"]
    if k in w: e["WRITE_SIZE_KB_sum"] = w[k]["WRITE_SIZE"]
    if k in mm:
        gui = mm[k].get("GRBM_GUI_ACTIVE", 0.0)
        e["mfma_busy_cycles"] = mm[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        e["mfma_flop"] = mm[k].get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) * 512
        e["mfma_flop_bf16"] = mm[k].get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0) * 512      # (the limb kernels' MFMAs: gru_limb.h)
        e["gui_active_sum_over_xcd"] = gui
        e["mfma_util_pct"] = 100.0 * e["mfma_busy_cycles"] / (gui / 8 * 1024) if gui else None   # 1024 SIMDs; GUI_ACTIVE is summed over 8 XCDs
    # gfx950: FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM section)
    if "FETCH_SIZE_KB_sum" in e and "WRITE_SIZE_KB_sum" in e and e["dispatches"]:
        e["hbm_bytes_per_dispatch_corrected"] = (2 * e["FETCH_SIZE_KB_sum"] + e["WRITE_SIZE_KB_sum"]) * 1024 / e["dispatches"]
    res[k] = e
json.dump(res, open(out + "/pmc_summary.json", "w"), indent=1)
for k, e in sorted(res.items(), key=lambda kv: -kv[1].get("mfma_flop", 0))[:8]:
    print(k, {a: (round(b, 1) if isinstance(b, float) else b) for a, b in e.items()})
PY
ls $OUT $OUT/trace_pipelined $OUT/trace_serial
