#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== bf16 NI=3 FA=1"; NI=3 FA=1 python tools/dbg_ws.py 2>&1 | grep -v amdgpu.ids | grep -A80 "poison test"
echo "== f32 NI=3 FA=1"; SMPLFIT_GEMM=f32 NI=3 FA=1 python tools/dbg_ws.py 2>&1 | grep -v amdgpu.ids | grep -A80 "poison test"
