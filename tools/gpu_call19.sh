#!/bin/bash
python -m pytest tests -m gpu -x -q > gpurun_out/c19_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/c19_tests.log
python tools/latency.py 2>/dev/null > gpurun_out/latency.json; python -c "
import json
d=json.load(open('gpurun_out/latency.json'))
for k,v in d.items(): print(k, v)"
python tools/bench_callers.py 2>/dev/null | tee gpurun_out/bench_callers.txt
