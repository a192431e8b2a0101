#!/bin/bash
# HBM traffic per fit (PMC, as profiles/pmc_traffic.json: (2 x FETCH_SIZE + WRITE_SIZE) KiB) against the batch of a call:
# what a depth-first sub-batch schedule would see.   bash tools/pmc_batch.sh "<B>:<chunks> ..."   (on the GPU box)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for spec in ${1:-1024:1 2048:2 4096:1 4096:2}; do
  B=${spec%%:*}; CH=${spec##*:}
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pb_$c
    SMPLFIT_CHUNKS=$CH timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pb_$c -o p -- python $R/tools/fit_only.py $B 4 > /dev/null 2>&1 < /dev/null
  done
  python - $B $CH <<'PY'
import csv, glob, re, sys
B, ch = int(sys.argv[1]), int(sys.argv[2])
tot = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    rows = []
    for f in glob.glob(f'/tmp/pb_{c}/**/*counter_collection.csv', recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    first = next(i for i, r in enumerate(rows) if 'k_layout_targets' in r['Kernel_Name'])   # the fits start here
    rows = [r for r in rows[first:] if re.search(r'\bk_[a-z_0-9]+', r['Kernel_Name'])]
    fits = sum(('k_refine_epilogue' in r['Kernel_Name'] or 'k_refine_bm' in r['Kernel_Name']) for r in rows) / ch
    tot[c] = sum(float(r['Counter_Value']) for r in rows) * 1024 / fits / B
print(f'B {B} chunks {ch}: HBM bytes per fit {2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]:,.0f} '
      f'(reads {2 * tot["FETCH_SIZE"]:,.0f}, writes {tot["WRITE_SIZE"]:,.0f})')
PY
done
