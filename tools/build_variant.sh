#!/bin/bash
# build_ab/lib<name>.so = the library with extra compiler flags:  tools/build_variant.sh <name> [-D...]
set -e
cd "$(dirname "$0")/.."
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value \
  -DSMPLFIT_BUILD_ID="\"variant-$name\"" "$@" smplfitter_amd/csrc/smplfit_hip.hip smplfitter_amd/csrc/sf_tables.cpp \
  -o build_ab/lib$name.so
echo build_ab/lib$name.so
