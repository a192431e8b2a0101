#!/bin/bash
python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/c18_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/c18_tests.log
bash tools/gpu_ab2.sh c18 build_ab/libjw1.so - build_ab/libjw4.so build_ab/libjw1.so - build_ab/libjw4.so
SMPLFIT_CHUNKS=1 bash tools/gpu_ab2.sh c18b build_ab/libjw1.so - build_ab/libjw4.so
for l in build_ab/libjw1.so ""; do SMPLFIT_LIB=$l python tools/ab_fit.py smplx 4096 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['lib'], d['fits_per_s'], d['checksum'])"; done
