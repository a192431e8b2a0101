#!/bin/bash
python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/c12_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/c12_tests.log
bash tools/gpu_ab2.sh c12 build_ab/libprev.so - build_ab/libslab64.so build_ab/libslab16.so build_ab/libprev.so -
SMPLFIT_CHUNKS=1 bash tools/gpu_ab2.sh c12b build_ab/libprev.so -
