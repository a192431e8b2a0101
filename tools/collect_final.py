"""Copy what tools/gpu_final.sh wrote under gpurun_out/ into profiles/ (run here, after the gpurun call):
    python tools/collect_final.py [tag]
"""
import csv, json, os, re, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else 'r06'
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O, F, P = f'{R}/gpurun_out/prof_{tag}/out', f'{R}/gpurun_out/final_{tag}', f'{R}/profiles'
shutil.copy(f'{O}/pmc_traffic.json', f'{P}/pmc_traffic.json')
for n in ('bench_chunks1.json', 'bench_default.json', 'kernel_stats_chunks2.csv', 'kernel_stats_chunks1.csv', 'pmc_mfma.json',
          'trace_summary.txt'):
    shutil.copy(f'{O}/{tag}_{n}', f'{P}/{tag}_{n}')
for c in ('c3', 'c4', 'c5'):
    shutil.copy(f'{F}/bench_{c}.json', f'{P}/{tag}_bench_{c}.json')
shutil.copy(f'{F}/bench_callers.txt', f'{P}/{tag}_bench_callers.txt')
shutil.copy(f'{F}/latency.json', f'{P}/{tag}_latency.json')
for n in ('bench_skin.json', 'bench_general.json', 'kstats_general.txt', 'pmc_general.json', 'wave_stamps_4096.txt'):
    if os.path.exists(f'{F}/{n}'):
        shutil.copy(f'{F}/{n}', f'{P}/{tag}_{n}')
build = json.load(open(f'{P}/{tag}_bench_c3.json'))['build']
shutil.copy(f'{F}/kernel_stats_c3.csv', f'{P}/{tag}_kernel_stats_c3.csv')  # (windowed on the box: tools/profile_collect.py --window)
for n in ('solve_stamps.txt', 'solve_ab_time.txt', 'overlap_probe.txt', 'general_path_parity.txt', 'ubench_lds_neighbour.txt'):
    if os.path.exists(f'{F}/{n}'):
        shutil.copy(f'{F}/{n}', f'{P}/{tag}_{n}')
if os.path.exists(f'{F}/bench_collective.json'):
    lines = [ln for ln in open(f'{F}/bench_collective.json') if ln.startswith('{')]
    open(f'{P}/{tag}_bench_collective.json', 'w').write(lines[0])
# the windowed c3 statistics against the bench line's HIP-event time of the GEMM (review item: within 10 %)
c3 = json.load(open(f'{P}/{tag}_bench_c3.json'))
for r in csv.DictReader(ln for ln in open(f'{P}/{tag}_kernel_stats_c3.csv') if not ln.startswith('#')):
    if r['kernel'].startswith('k_posedirs_gemm'):
        ev = c3['roofline'].get('kernel_ms', {}).get('posedirs_gemm')
        print('c3 GEMM: rocprofv3 avg', r['avg_us'], 'us, HIP events', None if ev is None else round(ev * 1e3, 1), 'us')
d = json.load(open(f'{R}/gpurun_out/pmc_sq_{tag}_smplx.json'))
g = d['k_posedirs_gemm_bf16x3_tiled']
cyc = g['GRBM_GUI_ACTIVE'] / 8
out = {'note': 'rocprofv3 --kernel-trace --pmc (4 separate passes) of tools/ab_fit.py smplx 4096, per-launch averages; '
               'tools/pmc_sq.sh <tag> - smplx', 'build': build,
       'k_posedirs_gemm_bf16x3_tiled': {
           'counters': g, 'launches': d['_launches']['k_posedirs_gemm_bf16x3_tiled'],
           'derived': {'kernel_cycles': cyc, 'mfma_busy_frac_of_simd_cycles': g['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024),
                       'wait_any_frac_of_wave_cycles': g['SQ_WAIT_ANY'] / g['SQ_WAVE_CYCLES'],
                       'lds_idx_active_frac_of_cu_cycles': g['SQ_LDS_IDX_ACTIVE'] / (cyc * 256)}},
       'other_kernels': {k: v for k, v in d.items() if k not in ('k_posedirs_gemm_bf16x3_tiled', '_launches')}}
json.dump(out, open(f'{P}/{tag}_pmc_sq_gemm_smplx.json', 'w'), indent=1)
print(build, out['k_posedirs_gemm_bf16x3_tiled']['derived'])
if os.path.exists(f'{R}/gpurun_out/pmc_sq_{tag}_smpl.json'):
    ds = json.load(open(f'{R}/gpurun_out/pmc_sq_{tag}_smpl.json'))
    json.dump({'note': 'rocprofv3 --kernel-trace --pmc (4 separate passes) of tools/ab_fit.py smpl 4096 (SMPLFIT_CHUNKS=1), per-launch averages; tools/pmc_sq.sh <tag> - smpl', 'build': build, **ds},
              open(f'{P}/{tag}_pmc_sq_vertex_passes.json', 'w'), indent=1)
for c in ('default', 'c3', 'c4', 'c5'):
    b = json.load(open(f'{P}/{tag}_bench_{c}.json'))
    print(c, b['value'], b['ms_per_step'], b['roofline']['kernel'], b['roofline']['frac'], b.get('cpu_baseline', {}).get('value'))
print('traffic build', json.load(open(f'{P}/pmc_traffic.json'))['build'])
