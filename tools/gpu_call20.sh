#!/bin/bash
export SMPLFIT_BM_KID=1
python tools/dbg_kid.py 2>/dev/null | tail -3
cd /tmp; export TMPDIR=/tmp; SMPLFIT_CHUNKS=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o t -- python $GRAFT_REPO_ROOT/tools/dbg_kid_trace.py > /dev/null 2>&1; python - <<PY
import csv, glob, collections
per = collections.defaultdict(list)
for f in glob.glob("/tmp/kt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        per[r["Kernel_Name"][:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:8]:
    v2 = sorted(v); print(f"{k:70s} n={len(v):4d} med={v2[len(v2)//2]:8.1f}")
PY
