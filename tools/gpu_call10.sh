#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/c10_tests.log 2>&1; echo "tests rc=$?"
tail -4 gpurun_out/c10_tests.log
python __graft_entry__.py smoke 2>&1 | tail -2
python bench.py --steps 20 --warmup 5 > gpurun_out/c10_bench.json 2> gpurun_out/c10_bench.err; cut -c1-400 gpurun_out/c10_bench.json
