#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in WAIT0 SYNC WAIT0SYNC; do echo "== $v"; SMPLFIT_LIB=$PWD/build_ab/libsmplfit_$v.so python tools/dbg_det.py 4096 2>&1 | grep -v amdgpu.ids | cut -c1-100; done
