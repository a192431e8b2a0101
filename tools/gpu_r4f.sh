#!/bin/bash
out=gpurun_out/r4f; mkdir -p $out
timeout 1200 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
tail -3 $out/pytest.log
ab() { local name=$1 lib=$2 kind=$3 B=$4; shift 4
  ( [ "$lib" != "-" ] && export SMPLFIT_LIB=$lib; for e in "$@"; do export "$e"; done; timeout 200 python tools/ab_fit.py $kind $B ) >> $out/ab.jsonl 2>> $out/ab.err; }
: > $out/ab.jsonl
ab nt27c1 - smpl 4096 SMPLFIT_CHUNKS=1
ab nt27c2 - smpl 4096
ab nt11c1 build_ab/libnt11.so smpl 4096 SMPLFIT_CHUNKS=1
python - $out/ab.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); print(d['lib'].split('/')[-1], d['kind'], d['B'], d['env'], d['kernel_us'], d['fits_per_s'], d['checksum'])
PY
timeout 300 python tools/latency.py > $out/latency_nt27.json 2>/dev/null; SMPLFIT_LIB=build_ab/libnt11.so timeout 300 python tools/latency.py > $out/latency_nt11.json 2>/dev/null
python - <<'PY'
import json
for n in ('nt27','nt11'):
    try:
        d=json.load(open(f'gpurun_out/r4f/latency_{n}.json')); print(n, {k:(v.get('fits_per_s') if isinstance(v,dict) else v) for k,v in d.items()})
    except Exception as e: print(n, 'failed', e)
PY
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench.json; python -c "import json; d=json.load(open('$out/bench.json')); print(d['value'], d['ms_per_step'])"
