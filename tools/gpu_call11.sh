#!/bin/bash
python -m pytest tests -m gpu -x -q > gpurun_out/c11_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/c11_tests.log
for cap in 384 256 192; do SMPLFIT_GROUP_CAP=$cap python tools/ab_fit.py smpl 4096 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['env'], d['kernel_us'], d['fits_per_s'])"; done
python tools/ab_fit.py smplx 4096 2>/dev/null | tail -1
python tools/ab_fit.py smpl 32768 2>/dev/null | tail -1
