#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/dbg_pg.py 2>&1 | grep -v amdgpu.ids | cut -c1-200
python tools/dbg_det.py 4096 2>&1 | grep -v amdgpu.ids | cut -c1-150
python tools/dbg_gemm.py 2048 2>&1 | grep -v amdgpu.ids | tail -2
for ch in 1 2 3; do SMPLFIT_CHUNKS=$ch python tools/ab_fit.py smpl 4096 2>/dev/null | tail -1; done
