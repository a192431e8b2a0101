#!/bin/bash
out=gpurun_out/r4j; mkdir -p $out
ab() { local name=$1 lib=$2 kind=$3 B=$4; shift 4
  ( [ "$lib" != "-" ] && export SMPLFIT_LIB=$lib; for e in "$@"; do export "$e"; done; timeout 200 python tools/ab_fit.py $kind $B ) >> $out/ab.jsonl 2>> $out/ab.err; }
: > $out/ab.jsonl
for l in - build_ab/libpgu3.so build_ab/libpgu9.so build_ab/libpgp2.so build_ab/libpgp8.so; do ab x $l smpl 4096 SMPLFIT_CHUNKS=1; ab x $l smplx 4096; done
python - $out/ab.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); print(d['lib'].split('/')[-1], d['kind'], d['B'], d['kernel_us']['pair_gram'], d['fits_per_s'], d['checksum'])
PY
timeout 1500 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log; tail -3 $out/pytest.log
for c in c2 c3 c4 c5; do timeout 300 python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$c', d['value'], d['ms_per_step'])"; done
