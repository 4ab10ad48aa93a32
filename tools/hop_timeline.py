"""Timeline of ONE streaming hop out of a rocprofv3 --kernel-trace CSV: python tools/hop_timeline.py <kernel_trace.csv> [hop index from the end]
Prints every launch of that hop (start offset, duration, gap to the previous end on the critical order) and per-kernel averages over the last 100 hops."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', ''), r.get('Stream_Id', r.get('Queue_Id', '?'))) for r in rows)
hops, cur = [], []
for e in ev:
    cur.append(e)
    if e[2].startswith('stream_ola'): hops.append(cur); cur = []
hops = hops[-100:]
agg = collections.defaultdict(lambda: [0, 0.0])
for h in hops:
    for s, e, n, q in h: agg[n][0] += 1; agg[n][1] += (e - s) / 1e3
print("per hop averages over %d hops: span %.1f us" % (len(hops), sum(max(e[1] for e in h) - h[0][0] for h in hops) / len(hops) / 1e3))
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("  %-70s %5.1f launches %8.1f us total %6.1f us each" % (n[:70], c / len(hops), t / len(hops), t / c))
h = hops[-int(sys.argv[3]) if len(sys.argv) > 3 else -2]
t0 = h[0][0]
print("one hop:")
for s, e, n, q in h:
    print("  +%7.1f us  %6.1f us  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, n[:90]))
