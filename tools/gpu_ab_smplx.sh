#!/bin/bash
# SMPL-X (c3) A/B of library builds: tools/gpu_ab_smplx.sh <tag> <lib-or-"-"> ...
tag=$1; shift
out=gpurun_out/abx_$tag.jsonl; : > $out
for l in "$@"; do
  if [ "$l" = "-" ]; then timeout 200 python tools/ab_fit.py smplx 4096 >> $out 2>>gpurun_out/abx_$tag.err
  else SMPLFIT_LIB=$l timeout 200 python tools/ab_fit.py smplx 4096 >> $out 2>>gpurun_out/abx_$tag.err; fi
done
python - $out <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); print(d['lib'].split('/')[-1], d['env'], d['kernel_us'], d['fits_per_s'], d['checksum'])
PY
