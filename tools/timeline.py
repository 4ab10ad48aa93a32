#!/usr/bin/env python3
"""Timeline analysis of a rocprofv3 kernel trace of bench.py (tools/trace_bench.sh): per HIP queue busy time,
concurrency histogram, and the kernels of the last traced step in start order."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"]); r["q"] = r["Queue_Id"]
    n = r["Kernel_Name"]; n = n.replace("void ", "").split("(")[0]
    r["n"] = n[:60]
# the last step = after the last STFT launch
stft = [i for i, r in enumerate(rows) if "StftA" in r["n"]]
lo = stft[-1]
step = rows[lo:]
t0 = step[0]["s"]; t1 = max(r["e"] for r in step)
print("step wall ms", (t1 - t0) / 1e6, "kernels", len(step))
byq = collections.defaultdict(list)
for r in step: byq[r["q"]].append(r)
for q, rs in byq.items():
    busy = sum(r["e"] - r["s"] for r in rs)
    print("queue", q, "kernels", len(rs), "busy ms", round(busy / 1e6, 2), "span", round((rs[0]["s"] - t0) / 1e6, 2), "->", round((max(r["e"] for r in rs) - t0) / 1e6, 2))
    cls = collections.defaultdict(float)
    for r in rs: cls[r["n"]] += (r["e"] - r["s"]) / 1e6
    for k, v in sorted(cls.items(), key=lambda kv: -kv[1])[:8]: print("    %7.2f  %s" % (v, k))
# concurrency histogram
ev = []
for r in step: ev.append((r["s"], 1)); ev.append((r["e"], -1))
ev.sort()
hist = collections.defaultdict(int); cur = 0; last = t0
for t, d in ev:
    hist[cur] += t - last; last = t; cur += d
print("concurrency (kernels in flight -> ms):", {k: round(v / 1e6, 2) for k, v in sorted(hist.items())})
if len(sys.argv) > 2:
    qn = sys.argv[2]
    prev = None
    for r in byq[qn][: int(sys.argv[3]) if len(sys.argv) > 3 else 80]:
        gap = (r["s"] - prev) / 1e3 if prev else 0
        print("%9.3f +%8.1f us gap %7.1f  %s grid %s" % ((r["s"] - t0) / 1e6, (r["e"] - r["s"]) / 1e3, gap, r["n"], r["Grid_Size_X"]))
        prev = r["e"]
