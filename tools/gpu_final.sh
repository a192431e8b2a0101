#!/bin/bash
# Round-end evidence on the GPU box (through gpurun): everything profiles/ holds for the round, from the tree as it is.
#   bash tools/gpu_final.sh [tag]      -> gpurun_out/prof_<tag>/out/* and gpurun_out/final_<tag>/*
TAG=${1:-r05}
R=$GRAFT_REPO_ROOT; F=$R/gpurun_out/final_$TAG; mkdir -p $F
bash tools/profile_round.sh $TAG > gpurun_out/prof_$TAG.log 2>&1 < /dev/null; tail -3 gpurun_out/prof_$TAG.log
for c in c3 c4 c5; do python bench.py --config $c --steps 10 --warmup 3 > $F/bench_$c.json 2>$F/bench_$c.err < /dev/null; python -c "
import json
d=json.load(open('$F/bench_$c.json')); print('$c', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'])"; done
python -c "
import json
d=json.load(open('gpurun_out/prof_$TAG/out/${TAG}_bench_default.json')); print('c2', d['value'], d['ms_per_step'], json.dumps(d['roofline'])[:1500])"
# per-kernel statistics of the SMPL-X configuration, SQ counters of its GEMM, callers, small-batch latency
(cd /tmp && export TMPDIR=/tmp && SMPLFIT_CHUNKS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c3trace -o t -- python $R/bench.py --config c3 --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1 < /dev/null)
f=$(find /tmp/c3trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $F/kernel_stats_c3.csv
timeout 300 bash tools/pmc_sq.sh ${TAG}_smplx - smplx > $F/pmc_sq_smplx.log 2>&1 < /dev/null
timeout 300 bash tools/pmc_sq.sh ${TAG}_smpl - smpl > $F/pmc_sq_smpl.log 2>&1 < /dev/null
timeout 300 python tools/bench_callers.py > $F/bench_callers.txt 2>$F/bench_callers.err < /dev/null; tail -12 $F/bench_callers.txt
timeout 200 python tools/latency.py > $F/latency.json 2>$F/latency.err < /dev/null; tail -2 $F/latency.json | cut -c1-300
timeout 200 python tools/bench_skin.py > $F/bench_skin.json 2>/dev/null < /dev/null; cat $F/bench_skin.json
timeout 300 python tools/bench_general.py > $F/bench_general.json 2>/dev/null < /dev/null; cat $F/bench_general.json
SMPLFIT_LIB=build_ab/libwstamp.so timeout 200 python tools/wave_stamps.py 4096 > $F/wave_stamps_4096.txt 2>&1 < /dev/null; head -12 $F/wave_stamps_4096.txt
