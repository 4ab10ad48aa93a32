"""Per-stream timeline of the LAST engine call in a rocprofv3 kernel trace (tools/trace_model.sh): for every HIP stream
(Queue_Id), the runs of back-to-back kernels with their first/last kernel and busy share.  python tools/call_timeline.py <csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '')[:34], r['Queue_Id']) for r in rows)
olas = [i for i, e in enumerate(ev) if e[2].startswith('ola_kernel')]
lo = ev[olas[-2]][1] if len(olas) > 1 else ev[0][0]
call = [e for e in ev if e[0] >= lo]
t0 = call[0][0]
print('call span %.2f ms, %d launches' % ((call[-1][1] - t0) / 1e6, len(call)))
byq = collections.defaultdict(list)
for e in call: byq[e[3]].append(e)
for qid, es in sorted(byq.items()):
    busy = sum(e[1] - e[0] for e in es)
    print('queue %s: %d launches, busy %.2f ms, first %.2f last %.2f ms' % (qid, len(es), busy / 1e6, (es[0][0] - t0) / 1e6, (es[-1][1] - t0) / 1e6))
    # gaps > 100 us
    prev = es[0]
    for e in es[1:]:
        if e[0] - prev[1] > 100000:
            print('    idle %.2f -> %.2f ms (%.2f) before %s' % ((prev[1] - t0) / 1e6, (e[0] - t0) / 1e6, (e[0] - prev[1]) / 1e6, e[2]))
        prev = e
g = [e for e in call if 'gru256' in e[2]]
print('gru256 scans: %d, total %.2f ms; first starts %.2f, last ends %.2f' % (len(g), sum(e[1] - e[0] for e in g) / 1e6, (g[0][0] - t0) / 1e6, (g[-1][1] - t0) / 1e6))
for e in g: print('   %.2f-%.2f %s q%s' % ((e[0] - t0) / 1e6, (e[1] - t0) / 1e6, e[2], e[3]))
