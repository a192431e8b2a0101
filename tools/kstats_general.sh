#!/bin/bash
# Per-kernel averages of the general path under rocprofv3:   tools/kstats_general.sh smpl_b300 [batch]
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kstats_g
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kstats_g -o t -- python $GRAFT_REPO_ROOT/tools/bench_general.py "$@" > /tmp/kstats_g.json 2>/dev/null < /dev/null
f=$(find /tmp/kstats_g -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] || { echo "no kernel_stats.csv"; exit 1; }
python - "$f" <<'PY'
import csv, re, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    m = re.search(r'(k_[a-z_0-9]+)', r['Name'])
    print(f"{(m.group(1) if m else r['Name'][:40]):32s} calls {r['Calls']:>4s}  avg {float(r['AverageNs'])/1e3:9.1f} us  {float(r['Percentage']):5.1f} %")
PY
tail -1 /tmp/kstats_g.json
