#!/bin/bash
out=gpurun_out/r4v; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log; tail -6 $out/pytest.log
timeout 200 python tools/latency.py > $out/latency.json 2>/dev/null; python - <<'PY'
import json
d = json.load(open('gpurun_out/r4v/latency.json'))
for k, v in d.items(): print(k, v)
PY
