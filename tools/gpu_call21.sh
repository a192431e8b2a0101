#!/bin/bash
python -m pytest tests -m gpu -x -q -k "smplx or gemm" > gpurun_out/c21_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/c21_tests.log
for g in f32 bf16; do SMPLFIT_GEMM=$g python tools/ab_fit.py smplx 4096 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['env'], d['kernel_us'], d['fits_per_s'], d['checksum'])"; done
