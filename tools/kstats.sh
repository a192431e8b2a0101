#!/bin/bash
# Per-kernel averages of one bench configuration under rocprofv3 (one chunk: launches of the whole batch):
#   tools/kstats.sh [bench args...]        e.g.  tools/kstats.sh --config c3
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kstats
SMPLFIT_CHUNKS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kstats -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > /tmp/kstats.json 2>/dev/null < /dev/null
f=$(find /tmp/kstats -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] || { echo "no kernel_stats.csv"; exit 1; }
python - "$f" <<'PY'
import csv, re, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    m = re.search(r'(k_[a-z_0-9]+)', r['Name'])
    print(f"{(m.group(1) if m else r['Name'][:40]):32s} calls {r['Calls']:>4s}  avg {float(r['AverageNs'])/1e3:8.1f} us  {float(r['Percentage']):5.1f} %")
PY
python -c "
import json; d=json.load(open('/tmp/kstats.json')); print('fits/s', d['value'], 'ms', d['ms_per_step'])"
