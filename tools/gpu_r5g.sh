#!/bin/bash
out=gpurun_out/r5g; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log; tail -4 $out/pytest.log
bash tools/gpu_final.sh r04 2>&1 | grep -v "^ *entry\|first step\|last step\|exit:\|loop time\|prologue\|xcc" | cut -c1-400 | tail -30
