#!/bin/bash
out=gpurun_out/r4e; mkdir -p $out
ab() { local name=$1 lib=$2 kind=$3 B=$4; shift 4
  ( [ "$lib" != "-" ] && export SMPLFIT_LIB=$lib; for e in "$@"; do export "$e"; done; timeout 200 python tools/ab_fit.py $kind $B ) >> $out/ab.jsonl 2>> $out/ab.err; }
: > $out/ab.jsonl
ab nt15 - smpl 4096 SMPLFIT_CHUNKS=1
ab nt15c2 - smpl 4096
ab nt0 build_ab/libnt0.so smpl 4096 SMPLFIT_CHUNKS=1
ab nt3 build_ab/libnt3.so smpl 4096 SMPLFIT_CHUNKS=1
ab nt7 build_ab/libnt7.so smpl 4096 SMPLFIT_CHUNKS=1
ab nt11 build_ab/libnt11.so smpl 4096 SMPLFIT_CHUNKS=1
ab nt15pf0 build_ab/libnt15pf0.so smpl 4096 SMPLFIT_CHUNKS=1
ab nt15s2k - smpl 4096 SMPLFIT_BM_SLOTS=2048
ab nt15c3 - smpl 4096 SMPLFIT_CHUNKS=3
ab r3x build_ab/libr3.so smplx 4096
ab newx - smplx 4096
ab newxc1 - smplx 4096 SMPLFIT_CHUNKS=1
ab new_32k - smpl 32768
python - $out/ab.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); print(d['lib'].split('/')[-1], d['kind'], d['B'], d['env'], d['kernel_us'], d['fits_per_s'], d['checksum'])
PY
for c in c4 c3; do timeout 300 python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$c new', d['value'], d['ms_per_step'])"; SMPLFIT_LIB=build_ab/libr3.so timeout 300 python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$c r3', d['value'], d['ms_per_step'])"; done
