#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/dbg_gemm.py 2048 2>&1 | grep -v amdgpu.ids
python tools/dbg_gemm.py 4096 2>&1 | grep -v amdgpu.ids
