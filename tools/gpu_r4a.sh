#!/bin/bash
# round 4, first GPU call: parity tests of the share-table kernels, then A/B against the round-3 library
out=gpurun_out/r4a; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
tail -5 $out/pytest.log
ab() { # name lib env...
  local name=$1 lib=$2; shift 2
  ( [ "$lib" != "-" ] && export SMPLFIT_LIB=$lib; for e in "$@"; do export "$e"; done; timeout 200 python tools/ab_fit.py smpl 4096 ) >> $out/ab.jsonl 2>> $out/ab.err
}
: > $out/ab.jsonl
ab r3 build_ab/libr3.so
ab r3c1 build_ab/libr3.so SMPLFIT_CHUNKS=1
ab new -
ab newc1 - SMPLFIT_CHUNKS=1
ab newc1all - SMPLFIT_CHUNKS=1 SMPLFIT_LBS_LAST=all
ab news2k - SMPLFIT_BM_SLOTS=2048
ab news8k - SMPLFIT_BM_SLOTS=8192
ab w1c1 build_ab/libw1.so SMPLFIT_CHUNKS=1
ab w2c1 build_ab/libw2.so SMPLFIT_CHUNKS=1
ab pc0c1 build_ab/libpc0.so SMPLFIT_CHUNKS=1
ab pc6c1 build_ab/libpc6.so SMPLFIT_CHUNKS=1
python - $out/ab.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); print(d['lib'].split('/')[-1], d['env'], d['kernel_us'], d['fits_per_s'], d['checksum'])
PY
tail -3 $out/ab.err
