"""Copy what tools/gpu_final.sh wrote under gpurun_out/ into profiles/ (run here, after the gpurun call):
    python tools/collect_final.py [tag]
"""
import csv, json, os, re, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else 'r05'
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
rows = list(csv.DictReader(open(f'{F}/kernel_stats_c3.csv')))
with open(f'{P}/{tag}_kernel_stats_c3.csv', 'w') as f:
    f.write('# rocprofv3 --kernel-trace --stats of `python bench.py --config c3 --steps 10 --warmup 3 --no-cpu-baseline` '
            f'(SMPL-X-shaped model, 4096 instances, SMPLFIT_CHUNKS=1: 4096-instance launches), build "{build}"\n')
    f.write('kernel,calls,total_us,avg_us,min_us,max_us,percent\n')
    for r in rows:
        m = re.search(r'(k_[a-z_0-9]+)', r['Name'])
        k = m.group(1) if m else r['Name'][:60].replace(',', ';')
        f.write(f"{k},{r['Calls']},{float(r['TotalDurationNs'])/1e3:.1f},{float(r['AverageNs'])/1e3:.2f},"
                f"{float(r['MinNs'])/1e3:.2f},{float(r['MaxNs'])/1e3:.2f},{float(r['Percentage']):.2f}\n")
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
