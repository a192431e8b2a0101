#!/bin/bash
# Cycle stamps at the sync points of k_joint_stage (one instance, lane 0; s_memtime ticks), printed by a debug build.
#   here (no GPU):   [STAMP_B=3] bash tools/stage_stamps.sh build   -> build_ab/lib_stamp.so (-DSMPLFIT_STAGE_STAMPS; instance 1000 or STAMP_B)
#   on the box:      bash tools/stage_stamps.sh run [smpl|smplx] [batch]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
if [ "$1" = "build" ]; then
  mkdir -p $R/build_ab
  cd $R/smplfitter_amd/csrc
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -DSMPLFIT_BUILD_ID='"stamps"' \
    -DSMPLFIT_STAGE_STAMPS ${STAMP_B:+-DSMPLFIT_STAMP_B=$STAMP_B} smplfit_hip.hip sf_tables.cpp -o $R/build_ab/lib_stamp.so
else
  cd $R
  SMPLFIT_LIB=build_ab/lib_stamp.so SMPLFIT_CHUNKS=1 timeout 200 python tools/ab_fit.py ${2:-smpl} ${3:-4096} < /dev/null 2>&1 | grep stamps | tail -${STAMP_LINES:-12}
fi
