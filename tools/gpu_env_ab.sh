#!/bin/bash
# A/B of environment settings on one box: tools/gpu_env_ab.sh <tag> "VAR=val VAR2=val" "..." ...   ("-" = no variables)
tag=$1; shift
out=gpurun_out/env_$tag.jsonl; : > $out
for e in "$@"; do
  if [ "$e" = "-" ]; then timeout 120 python tools/ab_fit.py smpl 4096 >> $out 2>>gpurun_out/env_$tag.err
  else env $e timeout 120 python tools/ab_fit.py smpl 4096 >> $out 2>>gpurun_out/env_$tag.err; fi
done
python - $out <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); print(d['env'], d['kernel_us'], d['fits_per_s'], d['checksum'])
PY
