"""Turn the raw rocprofv3 output of tools/profile_round.sh into the small files committed under profiles/:
   <tag>_kernel_stats_chunks1.csv / <tag>_kernel_stats.csv  per-kernel launch statistics (from the kernel traces)
   <tag>_trace_summary.txt                                  overlap of the chunk streams
   pmc_traffic.json                                         HBM bytes per launch per kernel + per fit, with the build id
   <tag>_pmc_mfma.json                                      matrix-pipe / VALU counters per kernel
Usage: python tools/profile_collect.py gpurun_out/prof_<tag> <tag>"""
import collections, csv, glob, json, os, re, sys

WINDOW = sys.argv[1] == '--window'  # python tools/profile_collect.py --window <trace dir> <out.csv> "<header>" [chunks]
if not WINDOW:
    src, tag = sys.argv[1], sys.argv[2]
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    DST = os.path.join(src, 'out')  # gpurun merges gpurun_out/ back; copy DST/* into profiles/ afterwards
    os.makedirs(DST, exist_ok=True)


def short(n):
    m = re.search(r'(k_[a-z_0-9]+)', n)
    return m.group(1) if m else n.split('(')[0][:48]


REFINE = ('k_refine_epilogue', 'k_refine_bm')  # the last kernel of a fit (round 6: the lane = instance form for models of <= 32 joints)


def trace_rows(d):
    rows = []
    for f in glob.glob(f'{d}/**/*kernel_trace.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
    rows.sort()
    return rows


STEPS = 10  # timed fits of the traced bench command (tools/profile_round.sh: --steps 10 --warmup 3)


def stats_csv(rows, path, head, chunks, warm=3):
    # steady state = the timed fits of the bench command: every fit ends with one k_refine_epilogue per chunk; the
    # window opens when the last warm-up fit has ended and closes with the last timed fit (bench.py's roofline leg —
    # repeated single-kernel launches — comes after it and is left out)
    ends = sorted(e for s, e, n in rows if short(n) in REFINE)
    # (bench.py's roofline leg launches k_refine_epilogue too — timing hook 10, round 5: only the refinements of the
    # warm-up and timed fits count)
    ends = ends[:(warm + STEPS) * chunks]
    t_lo, t_hi = ends[warm * chunks - 1], ends[-1]
    nfit = (len(ends) - warm * chunks) // chunks
    head += f'; statistics over the {nfit} timed fits (window between the end of the {warm}rd and of the last refinement kernel)'
    rows = [r for r in rows if r[0] >= t_lo and r[1] <= t_hi]
    per = collections.defaultdict(list)
    for s, e, n in rows:
        if s >= t_lo:
            per[short(n)].append((e - s) / 1e3)
    tot = sum(sum(v) for v in per.values())
    with open(path, 'w') as fh:
        fh.write(f'# {head}\n')
        fh.write('kernel,calls,total_us,avg_us,min_us,max_us,percent\n')
        for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            fh.write(f'{k},{len(v)},{sum(v):.1f},{sum(v)/len(v):.2f},{min(v):.2f},{max(v):.2f},{100*sum(v)/tot:.2f}\n')
    ev = sorted([(s, 1) for s, e, n in rows if s >= t_lo] + [(e, -1) for s, e, n in rows if s >= t_lo])
    depth, last, b1, b2 = 0, ev[0][0], 0, 0
    for tme, dl in ev:
        if depth >= 1: b1 += tme - last
        if depth >= 2: b2 += tme - last
        depth += dl; last = tme
    span = ev[-1][0] - ev[0][0]
    return (f'{os.path.basename(path)}: {nfit} fits, span {span/1e6:.2f} ms ({span/1e3/nfit:.1f} us per fit), kernel time summed '
            f'{tot/1e3:.2f} ms ({tot/nfit:.1f} us per fit), >= 1 kernel in flight {100*b1/span:.1f} %, >= 2 kernels {100*b2/span:.1f} %')


if WINDOW:
    # the same window for any traced `bench.py --steps 10 --warmup 3` command (round 6: the SMPL-X configuration —
    # round 5's c3 file came straight from rocprofv3 --stats and averaged warm-up and target synthesis in)
    print(stats_csv(trace_rows(sys.argv[2]), sys.argv[3], sys.argv[4], int(sys.argv[5]) if len(sys.argv) > 5 else 1))
    sys.exit(0)

bench = json.load(open(f'{src}/bench_chunks1.json'))  # build id / configuration (the default line is measured after this script)
build = bench['build']
head = f'rocprofv3 --kernel-trace of `python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs`, build "{build}", git HEAD see profiles/README.md'
lines = []
for sub, name, extra, nch in (('trace1', f'{tag}_kernel_stats_chunks1.csv', 'SMPLFIT_CHUNKS=1 (4096-instance launches)', 1),
                              ('trace2', f'{tag}_kernel_stats_chunks2.csv', 'SMPLFIT_CHUNKS=2 (two 2048-instance chunks on two streams)', 2)):
    rows = trace_rows(f'{src}/{sub}')
    if rows:
        lines.append(stats_csv(rows, f'{DST}/{name}', head + '; ' + extra, nch))
open(f'{DST}/{tag}_trace_summary.txt', 'w').write('\n'.join(lines) + '\n')

# ---- PMC: mean counter value per dispatch per kernel (steady-state dispatches of the fit only)
def pmc(d):
    res = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f'{d}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            res[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
    return res

fetch, write = pmc(f'{src}/pmc_FETCH_SIZE'), pmc(f'{src}/pmc_WRITE_SIZE')
B = bench['config']['batch_per_gpu']
kern = {}
for k in sorted(set(fetch) | set(write)):
    if not k.startswith('k_'):
        continue
    f_kib = fetch[k].get('FETCH_SIZE', [0]); w_kib = write[k].get('WRITE_SIZE', [0])
    # MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the
    # bytes of wide coalesced streaming reads -> x2 (WRITE_SIZE uncalibrated, taken as is)
    kern[k] = dict(bytes_per_launch=int((2 * sum(f_kib) / len(f_kib) + sum(w_kib) / len(w_kib)) * 1024),
                   launches_in_pass=len(f_kib))
# launches per fit (num_iter = 3) from the one-chunk trace
rows = trace_rows(f'{src}/trace1')
# only the fits themselves: bench.py's roofline leg (smplfit_time_kernel_f32: repeated single-kernel launches) and
# its round-trip forward come after the last fit's epilogue
fit_ends = sorted(e for s, e, n in rows if short(n) in REFINE)[:3 + STEPS]  # (not the timing hook's launches)
last_fit_end = fit_ends[-1] if fit_ends else None
if last_fit_end is not None:
    rows = [r for r in rows if r[0] <= last_fit_end]
cnt = collections.Counter(short(n) for s, e, n in rows)
fits = max(1, sum(cnt.get(k, 0) for k in REFINE) or 1)
per_fit = 0
for k, v in kern.items():
    v['launches_per_fit'] = round(cnt.get(k, 0) / fits, 2)
    per_fit += v['bytes_per_launch'] * v['launches_per_fit']
json.dump(dict(build=build, batch=B, config=bench['config']['name'], chunks=1,
               note='HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB, rocprofv3 --pmc, separate passes, SMPLFIT_CHUNKS=1',
               kernels={k: v['bytes_per_launch'] for k, v in kern.items()}, detail=kern,
               bytes_per_fit=int(per_fit / B)), open(f'{DST}/pmc_traffic.json', 'w'), indent=1)
m = pmc(f'{src}/pmc_MFMA')
json.dump({k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in m.items() if k.startswith('k_')},
          open(f'{DST}/{tag}_pmc_mfma.json', 'w'), indent=1)
for n in ('bench_default.json', 'bench_chunks1.json'):
    if os.path.exists(f'{src}/{n}'):
        open(f'{DST}/{tag}_{n}', 'w').write(open(f'{src}/{n}').read())
print('\n'.join(lines)); print('bytes per fit (PMC):', int(per_fit / B))
