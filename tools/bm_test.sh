cd /root/repo
SMPLFIT_BM=1 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "fit_goldens or fit_vs_oracle or full_size" 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp
SMPLFIT_BM=1 SMPLFIT_CHUNKS=1 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_bm -- python /root/repo/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /root/repo/gpurun_out/bench_bm.json 2>/dev/null
cat /root/repo/gpurun_out/bench_bm.json | cut -c1-300
python - <<'PY'
import csv,glob
f=glob.glob('/root/repo/gpurun_out/prof_bm/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(r['Name'][:60], r['Calls'], r['AverageNs'])
PY
