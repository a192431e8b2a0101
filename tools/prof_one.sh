#!/bin/bash
# one-chunk kernel averages of the current build (top 14)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
SMPLFIT_CHUNKS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_one -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_one.json 2>/dev/null
f=$(find $R/gpurun_out/prof_one -name "*kernel_stats.csv" | xargs ls -t | head -1); python - "$f" <<'PY'
import csv, sys, re
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    n = re.search(r'(k_[a-z_0-9]+)', r['Name']); n = n.group(1) if n else r['Name'][:30]
    print(f"  {n:32s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:8.1f} pct {r['Percentage']}")
PY
