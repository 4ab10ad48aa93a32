# Which kernels of the library does the GPU suite launch?  rocprofv3 --kernel-trace --stats over `pytest -m gpu` (the pytest process itself; the
# bench.py child processes of tests/test_gpu_bench_flow.py are not in this file), reduced to name + calls.  On the GPU box:
#   bash tools/suite_kernel_census.sh   ->  gpurun_out/suite_kernel_census.csv     then: python tools/suite_kernel_census.py
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp -o suite -- python -m pytest tests -q -m gpu 2>&1 | grep -a "passed\|failed" | tail -2
python - <<'PY'
import csv, glob
rows = {}
for f in glob.glob("/tmp/sp/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows[r["Name"]] = rows.get(r["Name"], 0) + int(r["Calls"])
with open("gpurun_out/suite_kernel_census.csv", "w") as o:
    w = csv.writer(o); w.writerow(["Name", "Calls"])
    for k in sorted(rows): w.writerow([k, rows[k]])
print(len(rows), "kernel names")
PY
