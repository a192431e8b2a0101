#!/bin/bash
bash tools/gpu_ab2.sh c13 build_ab/libprev.so - build_ab/libprev.so -
SMPLFIT_CHUNKS=1 bash tools/gpu_ab2.sh c13b build_ab/libprev.so -
cd /tmp; export TMPDIR=/tmp
SMPLFIT_CHUNKS=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o t -- python $GRAFT_REPO_ROOT/tools/ab_fit.py smpl 4096 > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
per = collections.defaultdict(list)
for f in glob.glob('/tmp/tr/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        per[r['Kernel_Name'][:60]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v); print(f'{k:60s} n={len(v):5d} med={v2[len(v2)//2]:8.1f} us')
PY
