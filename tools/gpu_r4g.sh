#!/bin/bash
out=gpurun_out/r4g; mkdir -p $out
ab() { local name=$1 lib=$2 kind=$3 B=$4; shift 4
  ( [ "$lib" != "-" ] && export SMPLFIT_LIB=$lib; for e in "$@"; do export "$e"; done; timeout 200 python tools/ab_fit.py $kind $B ) >> $out/ab.jsonl 2>> $out/ab.err; }
: > $out/ab.jsonl
ab c1 - smpl 4096 SMPLFIT_CHUNKS=1
for s in 0 1 2 3 4 5; do ab c2s$s - smpl 4096 SMPLFIT_CHUNKS=2 SMPLFIT_STAGGER=$s; done
for s in 0 2 3; do ab c3s$s - smpl 4096 SMPLFIT_CHUNKS=3 SMPLFIT_STAGGER=$s; done
for s in 0 2; do ab c4s$s - smpl 4096 SMPLFIT_CHUNKS=4 SMPLFIT_STAGGER=$s; done
ab x_c1 - smplx 4096 SMPLFIT_CHUNKS=1
for s in 0 2 3; do ab x_c2s$s - smplx 4096 SMPLFIT_CHUNKS=2 SMPLFIT_STAGGER=$s; done
python - $out/ab.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); print(d['kind'], d['B'], d['env'], d['fits_per_s'], d['checksum'])
PY
