#!/bin/bash
# A/B of library variants: tools/gpu_ab2.sh <out-tag> <lib-or-"-"> ...   (env applies to all)
tag=$1; shift
out=gpurun_out/ab_$tag.jsonl; : > $out
for l in "$@"; do
  if [ "$l" = "-" ]; then timeout 120 python tools/ab_fit.py smpl 4096 >> $out 2>>gpurun_out/ab_$tag.err
  else SMPLFIT_LIB=$l timeout 120 python tools/ab_fit.py smpl 4096 >> $out 2>>gpurun_out/ab_$tag.err; fi
done
python - $out <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); print(d['lib'].split('/')[-1], d['env'], d['kernel_us'], d['fits_per_s'], d['checksum'])
PY
