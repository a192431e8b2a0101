#!/bin/bash
for l in 1 2 1 2; do SMPLFIT_CHUNKS=1 SMPLFIT_LPT=$l python tools/ab_fit.py smpl 4096 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['env'], d['kernel_us'], d['fits_per_s'], d['checksum'])"; done
