"""Summarise a rocprofv3 --kernel-trace CSV (…_kernel_trace.csv): per kernel the launch count, the average
/ min / max duration (us), the share of the summed kernel time, and how much of the wall-clock span of
the trace had >= 1 / >= 2 kernels in flight (overlap of the chunked fit's streams).

    python tools/trace_summary.py gpurun_out/<dir> [skip_first_n_fits]
"""
import csv, glob, re, sys, collections

d = sys.argv[1]
files = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)
assert files, 'no *kernel_trace.csv under ' + d
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
def short(n):
    m = re.search(r'(k_[a-z_0-9]+)(<[^>]*>)?', n)
    return (m.group(1) + (m.group(2) or '')) if m else n[:50]
# steady state: drop the first third of the trace (model set-up, target generation, warm-up fits)
t_lo = rows[0][0] + (rows[-1][1] - rows[0][0]) // 3
rows = [r for r in rows if r[0] >= t_lo]
per = collections.defaultdict(list)
for s, e, n in rows:
    per[short(n)].append((e - s) / 1e3)
tot = sum(sum(v) for v in per.values())
print(f'{"kernel":58s} {"n":>6s} {"avg us":>9s} {"min":>8s} {"max":>8s} {"share":>6s}')
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    print(f'{k:58s} {len(v):6d} {sum(v)/len(v):9.1f} {min(v):8.1f} {max(v):8.1f} {100*sum(v)/tot:5.1f}%')
ev = sorted([(s, 1) for s, e, n in rows] + [(e, -1) for s, e, n in rows])
depth, last, busy1, busy2 = 0, ev[0][0], 0, 0
for tme, dlt in ev:
    if depth >= 1: busy1 += tme - last
    if depth >= 2: busy2 += tme - last
    depth += dlt; last = tme
span = ev[-1][0] - ev[0][0]
print(f'span {span/1e6:.2f} ms, kernel time summed {tot/1e3:.2f} ms, >=1 kernel in flight {100*busy1/span:.1f}% of span, >=2 {100*busy2/span:.1f}%')
