#!/bin/bash
# Device-side AddressSanitizer run (ROCm ASAN, xnack+): the library built with -fsanitize=address for gfx950:xnack+
# (tools/asan_device.sh build, in the build container) runs a set of small calls over every kernel family on the GPU
# box (tools/asan_device.sh run) — out-of-bounds global / LDS accesses of the kernels are reported by the runtime.
set -e
cd "$(dirname "$0")/.."
if [ "$1" = "build" ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950:xnack+ -O1 -g -std=c++17 -fPIC -shared -Wno-unused-value -fsanitize=address \
    -shared-libsan -DSMPLFIT_BUILD_ID='"asan"' smplfitter_amd/csrc/smplfit_hip.hip smplfitter_amd/csrc/sf_tables.cpp \
    -o build_ab/libasan.so
  echo build_ab/libasan.so; exit 0
fi
export HSA_XNACK=1
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
export LD_PRELOAD=$RT
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=0:protect_shadow_gap=0
export SMPLFIT_LIB=$PWD/build_ab/libasan.so
timeout ${ASAN_TIMEOUT:-400} python tools/asan_device_calls.py
