#!/bin/bash
bash tools/profile_round.sh r03 > gpurun_out/prof_r03.log 2>&1; tail -3 gpurun_out/prof_r03.log
for c in c3 c4 c5; do python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_$c.json 2>gpurun_out/bench_$c.err; python -c "
import json
d=json.load(open('gpurun_out/bench_$c.json')); print('$c', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'])"; done
python -c "
import json
d=json.load(open('gpurun_out/prof_r03/out/r03_bench_default.json')); print('c2', d['value'], d['ms_per_step'], json.dumps(d['roofline'])[:1500])"
