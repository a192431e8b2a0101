#!/bin/bash
# kernel stats (1 chunk) for two vertex-group caps
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for c in 100000 256; do
SMPLFIT_GROUP_CAP=$c SMPLFIT_CHUNKS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cap$c -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_cap$c.json 2>/dev/null
echo cap $c; cut -c1-110 $R/gpurun_out/prof_cap$c.json
f=$(find $R/gpurun_out/prof_cap$c -name "*kernel_stats.csv" | head -1); python - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    n = re.search(r'(k_[a-z_0-9]+)', r['Name']); n = n.group(1) if n else r['Name'][:30]
    print(f"  {n:32s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:8.1f} pct {r['Percentage']}")
PY
done
