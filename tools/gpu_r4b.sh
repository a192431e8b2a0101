#!/bin/bash
# round 4: parity tests of the cell-table kernels, then A/B (prefetch variants, waves per workgroup, cell sizes, slots)
out=gpurun_out/r4b; mkdir -p $out
timeout 1200 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
tail -8 $out/pytest.log
ab() { # name lib kind B env...
  local name=$1 lib=$2 kind=$3 B=$4; shift 4
  ( [ "$lib" != "-" ] && export SMPLFIT_LIB=$lib; for e in "$@"; do export "$e"; done; timeout 200 python tools/ab_fit.py $kind $B ) >> $out/ab.jsonl 2>> $out/ab.err
}
: > $out/ab.jsonl
ab r3 build_ab/libr3.so smpl 4096
ab new - smpl 4096
ab newc1 - smpl 4096 SMPLFIT_CHUNKS=1
ab pf0c1 build_ab/libpf0.so smpl 4096 SMPLFIT_CHUNKS=1
ab pf1c1 build_ab/libpf1.so smpl 4096 SMPLFIT_CHUNKS=1
ab pf2c1 build_ab/libpf2.so smpl 4096 SMPLFIT_CHUNKS=1
ab w1c1 build_ab/libw1.so smpl 4096 SMPLFIT_CHUNKS=1
ab c128c1 build_ab/libc128.so smpl 4096 SMPLFIT_CHUNKS=1
ab c32c1 build_ab/libc32.so smpl 4096 SMPLFIT_CHUNKS=1
ab news2k - smpl 4096 SMPLFIT_BM_SLOTS=2048
ab news2kc1 - smpl 4096 SMPLFIT_BM_SLOTS=2048 SMPLFIT_CHUNKS=1
ab r3x build_ab/libr3.so smplx 4096
ab newx - smplx 4096
ab newxc1 - smplx 4096 SMPLFIT_CHUNKS=1
ab r3_32k build_ab/libr3.so smpl 32768
ab new_32k - smpl 32768
python - $out/ab.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); print(d['lib'].split('/')[-1], d['kind'], d['B'], d['env'], d['kernel_us'], d['fits_per_s'], d['checksum'])
PY
tail -3 $out/ab.err
