#!/bin/bash
out=gpurun_out/r5a; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log; tail -6 $out/pytest.log
