"""Which kernels run concurrently, by share of time: python tools/concurrency.py <kernel_trace.csv> [lo_ms hi_ms]
(rocprofv3 --kernel-trace --output-format csv).  Classes: GRU-64 kernels by name:workgroups, g256 = GRU-256 cluster scans, o = the rest."""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', ''),
             int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) // int(r['Workgroup_Size_X'])) for r in rows)
t0 = ev[0][0]
olas = [(e - t0) / 1e6 for s, e, n, w in ev if n.startswith('ola_kernel')]
print('ola ends (ms):', [round(x, 1) for x in olas])
lo = t0 + int(float(sys.argv[2]) * 1e6) if len(sys.argv) > 2 else t0 + int((olas[0] + 0.5) * 1e6)
hi = t0 + int(float(sys.argv[3]) * 1e6) if len(sys.argv) > 3 else t0 + int((olas[min(3, len(olas) - 1)] - 0.4) * 1e6)
def short(i):
    s, e, n, w = ev[i]
    if 'gru64' in n: return n.replace('gru64_', '')[:14] + ':%d' % w
    if 'gru256_cluster' in n: return 'g256'
    return 'o'
pts = []
for i, (s, e, n, w) in enumerate(ev):
    if e < lo or s > hi: continue
    pts.append((max(s, lo), 1, i)); pts.append((min(e, hi), -1, i))
pts.sort()
run = set(); dur = collections.Counter(); last = lo
for t, d, i in pts:
    dur[tuple(sorted(short(j) for j in run))] += t - last; last = t
    run.add(i) if d > 0 else run.discard(i)
tot = sum(dur.values())
print('window %.1f ms' % (tot / 1e6))
no64 = sum(v for k, v in dur.items() if not any('kernel' in x for x in k))
print('no GRU-64 kernel resident: %.1f %%' % (100 * no64 / tot))
for k, v in sorted(dur.items(), key=lambda kv: -kv[1])[:25]:
    print('%5.1f %%  %s' % (100 * v / tot, k))
