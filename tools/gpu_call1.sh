#!/bin/bash
# GPU call 1 of round 2: full gpu test suite, A/B of the GEMM rewrite against the round-1 library, bench lines
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/c1_tests.log 2>&1; echo "tests rc=$?" | tee gpurun_out/c1_rc.txt
tail -5 gpurun_out/c1_tests.log
for rep in 1 2; do
SMPLFIT_LIB=$PWD/build_ab/libsmplfit_r1.so python tools/ab_fit.py smpl 4096 2>/dev/null | tail -1 | tee -a gpurun_out/c1_ab.jsonl
python tools/ab_fit.py smpl 4096 2>/dev/null | tail -1 | tee -a gpurun_out/c1_ab.jsonl
done
SMPLFIT_CHUNKS=1 SMPLFIT_LIB=$PWD/build_ab/libsmplfit_r1.so python tools/ab_fit.py smpl 4096 2>/dev/null | tail -1 | tee -a gpurun_out/c1_ab.jsonl
SMPLFIT_CHUNKS=1 python tools/ab_fit.py smpl 4096 2>/dev/null | tail -1 | tee -a gpurun_out/c1_ab.jsonl
SMPLFIT_LIB=$PWD/build_ab/libsmplfit_r1.so python tools/ab_fit.py smplx 4096 2>/dev/null | tail -1 | tee -a gpurun_out/c1_ab.jsonl
python tools/ab_fit.py smplx 4096 2>/dev/null | tail -1 | tee -a gpurun_out/c1_ab.jsonl
python bench.py --steps 20 --warmup 5 > gpurun_out/c1_bench.json 2> gpurun_out/c1_bench.err; tail -c 1500 gpurun_out/c1_bench.json
