#!/bin/bash
out=gpurun_out/lpt.jsonl; : > $out
for ch in 1 2; do for lpt in 0 1; do for cap in 384 256 192 128; do
  SMPLFIT_CHUNKS=$ch SMPLFIT_LPT=$lpt SMPLFIT_GROUP_CAP=$cap timeout 120 python tools/ab_fit.py smpl 4096 >> $out 2>>gpurun_out/lpt.err
done; done; done
python - <<'PY'
import json
for l in open('gpurun_out/lpt.jsonl'):
    d = json.loads(l); print(d['env'], d['kernel_us'].get('accum'), d['kernel_us'].get('lbs'), d['fits_per_s'], d['checksum'])
PY
