#!/bin/bash
# per-kernel times of the ablation builds (tools/build_variant.sh ablN -DSMPLFIT_ABL=N)
out=gpurun_out/abl.jsonl; : > $out
python tools/ab_fit.py smpl 4096 >> $out 2>gpurun_out/abl.err
for l in build_ab/libabl*.so; do SMPLFIT_LIB=$l timeout 120 python tools/ab_fit.py smpl 4096 >> $out 2>>gpurun_out/abl.err; done
python - <<'PY'
import json
for l in open('gpurun_out/abl.jsonl'):
    d = json.loads(l); print(d['lib'].split('/')[-1], d['kernel_us'], d['fits_per_s'][-1])
PY
