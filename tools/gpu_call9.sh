#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/dbg_pg.py 2>&1 | grep -v amdgpu.ids | cut -c1-200
python tools/dbg_det.py 4096 2>&1 | grep -v amdgpu.ids | cut -c1-120 | head -4
for ch in 1 2; do SMPLFIT_CHUNKS=$ch python tools/ab_fit.py smpl 4096 2>/dev/null | tail -1; done
python tools/ab_fit.py smpl 32768 2>/dev/null | tail -1
python tools/ab_fit.py smpl 1024 2>/dev/null | tail -1
