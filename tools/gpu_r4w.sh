#!/bin/bash
out=gpurun_out/r4w; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "share or known_pose or weighted_batch_major or stage_half" > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log; tail -6 $out/pytest.log
timeout 200 python tools/latency.py > $out/latency.json 2>/dev/null; python - <<'PY'
import json
d = json.load(open('gpurun_out/r4w/latency.json'))
for k, v in d.items():
    if k.startswith('cfg'): print(k, v)
PY
