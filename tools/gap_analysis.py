"""GPU idle time inside the optimisation steps: parse a rocprofv3 --kernel-trace CSV of `bench.py` and report, for the
steady-state steps, kernel-busy time vs wall span and the largest gaps.  usage: python tools/gap_analysis.py <kernel_trace.csv>"""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0]))
rows.sort()
# steps are delimited by adam_kernel launches (two per step: coarse+fine share one flat buffer -> one launch per step)
adam = [i for i, r in enumerate(rows) if r[2].startswith('adam_kernel')]
print('kernels', len(rows), 'adam launches', len(adam))
spans = []
for a, b in zip(adam[3:-1], adam[4:]):   # skip warm-up
    seg = rows[a + 1:b + 1]
    t0, t1 = seg[0][0], seg[-1][1]
    busy = 0; cur_s, cur_e = seg[0][0], seg[0][1]
    gaps = []
    for s, e, n in seg[1:]:
        if s > cur_e:
            gaps.append((s - cur_e, n)); busy += cur_e - cur_s; cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    spans.append((t1 - t0, busy, len(seg), gaps))
for span, busy, n, gaps in spans[:6]:
    gaps.sort(reverse=True)
    print('step: %d kernels, span %.3f ms, busy %.3f ms (%.1f %%), idle %.3f ms; largest gaps (us): %s' % (
        n, span / 1e6, busy / 1e6, 100 * busy / span, (span - busy) / 1e6, ', '.join('%.1f before %s' % (g / 1e3, k[:28]) for g, k in gaps[:5])))
