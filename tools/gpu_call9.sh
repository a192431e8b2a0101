#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/dbg_det.py 4096 2>&1 | grep -v amdgpu.ids | cut -c1-150
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "hipgraph or concurrent or full_size or fit_goldens" 2>&1 | tail -3
for ch in 1 2; do for ax in 0 1; do SMPLFIT_AUX=$ax SMPLFIT_CHUNKS=$ch python tools/ab_fit.py smpl 4096 2>/dev/null | tail -1; done; done
