#!/bin/bash
python -m pytest tests -m gpu -x -q -k "gemm or full_size or fit_goldens or determin or batch_sizes or ragged" > gpurun_out/c16_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/c16_tests.log
bash tools/gpu_ab2.sh c16 build_ab/libtp1.so - build_ab/libgabl1.so build_ab/libtp1.so -
timeout 300 python tools/dbg_pg.py 2>&1 | tail -3
