cd /tmp && export TMPDIR=/tmp
for e in 0 3 4 5; do
SMPLFIT_EXP=$e SMPLFIT_BM=1 SMPLFIT_CHUNKS=1 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_exp$e -- python /root/repo/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob('/root/repo/gpurun_out/prof_exp$e/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'partsum_bm' in r['Name'] or 'transpose' in r['Name']: print($e, r['Name'][:50], r['Calls'], r['AverageNs'])
PY
done
