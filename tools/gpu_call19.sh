#!/bin/bash
python -m pytest tests -m gpu -x -q > gpurun_out/c19_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/c19_tests.log
bash tools/gpu_ab2.sh c19 - build_ab/libpg1.so - build_ab/libpg1.so
python tools/latency.py 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items():
    if k.startswith('cfg_'): print(k, v)"
