cd /tmp; export TMPDIR=/tmp
for cap in 384 256 192; do
  SMPLFIT_GROUP_CAP=$cap SMPLFIT_CHUNKS=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cap$cap -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > /tmp/cap$cap.json 2>/dev/null
  echo "== cap $cap: $(python -c "import json;d=json.load(open('/tmp/cap$cap.json'));print(d['value'], d['ms_per_step'])")"
  python - /tmp/cap$cap <<'PY'
import csv,glob,sys,re
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    m=re.search(r'(k_[a-z_0-9]+)',r['Name']); 
    if m: print('   %-28s calls %4s avg %8.1f us  total %8.1f us' % (m.group(1), r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e3))
PY
done
