#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fit_goldens or full_size or hipgraph" 2>&1 | tail -2
for rep in 1 2; do
SMPLFIT_LIB=$PWD/build_ab/libsmplfit_prev.so python tools/ab_fit.py smpl 4096 2>/dev/null | tail -1
python tools/ab_fit.py smpl 4096 2>/dev/null | tail -1
done
SMPLFIT_CHUNKS=1 SMPLFIT_LIB=$PWD/build_ab/libsmplfit_prev.so python tools/ab_fit.py smpl 4096 2>/dev/null | tail -1
SMPLFIT_CHUNKS=1 python tools/ab_fit.py smpl 4096 2>/dev/null | tail -1
