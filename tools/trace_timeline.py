"""Timeline of one steady-state fit from a rocprofv3 kernel trace:  python tools/trace_timeline.py <trace dir> [fit index]
(the trace of `python tools/fit_only.py 4096 8`): every kernel with start / end relative to the fit's first kernel, its
queue, and what ran beside it."""
import csv, glob, re, sys
d = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -2
rows = []
for f in glob.glob(f'{d}/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r'(k_[a-z_0-9]+)', r['Kernel_Name'])
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), m.group(1) if m else r['Kernel_Name'][:30], r.get('Queue_Id', '?')))
rows.sort()
# fits are delimited by k_layout_targets on the first queue
starts = [i for i, r in enumerate(rows) if r[2] == 'k_layout_targets']
qs = sorted({rows[i][3] for i in starts})
first = [i for i in starts if rows[i][3] == qs[0]] if len(qs) > 1 else starts
nper = len(starts) // max(1, len(first))
a = first[which]
b = first[which + 1] if which + 1 < len(first) and which != -1 else len(rows)
seg = rows[a:b]
t0 = seg[0][0]
print(f'{len(first)} fits in the trace, {nper} chunk(s); fit {which}: {len(seg)} kernels, span {(max(r[1] for r in seg) - t0) / 1e3:.1f} us')
for s, e, n, q in seg:
    beside = [m for (s2, e2, m, q2) in seg if q2 != q and s2 < e and e2 > s]
    print(f'{(s - t0) / 1e3:8.1f} {(e - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f}  q{q}  {n:28s} | {" ".join(sorted(set(beside)))}')
