#!/bin/bash
out=gpurun_out/r4d; mkdir -p $out
ab() { local name=$1 lib=$2 kind=$3 B=$4; shift 4
  ( [ "$lib" != "-" ] && export SMPLFIT_LIB=$lib; for e in "$@"; do export "$e"; done; timeout 200 python tools/ab_fit.py $kind $B ) >> $out/ab.jsonl 2>> $out/ab.err; }
: > $out/ab.jsonl
ab new - smpl 4096 SMPLFIT_CHUNKS=1
ab nopre - smpl 4096 SMPLFIT_CHUNKS=1 SMPLFIT_TIME_NOPRE=1
ab nt1 build_ab/libnt1.so smpl 4096 SMPLFIT_CHUNKS=1
ab nt2 build_ab/libnt2.so smpl 4096 SMPLFIT_CHUNKS=1
ab nt3 build_ab/libnt3.so smpl 4096 SMPLFIT_CHUNKS=1
ab nt3c2 build_ab/libnt3.so smpl 4096
ab new2 - smpl 2048 SMPLFIT_CHUNKS=1
ab new1 - smpl 1024 SMPLFIT_CHUNKS=1
python - $out/ab.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); print(d['lib'].split('/')[-1], d['kind'], d['B'], d['env'], d['kernel_us'], d['fits_per_s'], d['checksum'])
PY
