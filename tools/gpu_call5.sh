#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== default"; python tools/dbg_det.py 4096 2>&1 | grep -v amdgpu.ids
echo "== chunks1"; SMPLFIT_CHUNKS=1 python tools/dbg_det.py 4096 2>&1 | grep -v amdgpu.ids
echo "== f32"; SMPLFIT_GEMM=f32 python tools/dbg_det.py 4096 2>&1 | grep -v amdgpu.ids
echo "== B=2048"; python tools/dbg_det.py 2048 2>&1 | grep -v amdgpu.ids
