#!/bin/bash
cd /tmp; export TMPDIR=/tmp
for B in 32 1024; do
rm -rf /tmp/st; rocprofv3 --kernel-trace --output-format csv -d /tmp/st -o t -- python $GRAFT_REPO_ROOT/tools/small_trace.py $B > /dev/null 2>&1
python - $B <<'PY'
import csv, glob, sys, re
rows = []
for f in glob.glob('/tmp/st/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
# last fit call: find the last k_layout_targets
idx = [i for i, r in enumerate(rows) if 'k_layout_targets' in r[2]]
i0 = idx[-2]; i1 = idx[-1]
seg = rows[i0:i1]
t0 = seg[0][0]
print(f'B={sys.argv[1]}: one call = {(rows[i1][0]-t0)/1e3:.1f} us, {len(seg)} kernels, kernel time {sum(e-s for s,e,_ in seg)/1e3:.1f} us')
for s, e, n in seg:
    m = re.search(r'(k_[a-z_0-9]+)', n); nm = m.group(1) if m else n[:40]
    print(f'  +{(s-t0)/1e3:7.1f} {(e-s)/1e3:7.1f} us  {nm}')
PY
done
