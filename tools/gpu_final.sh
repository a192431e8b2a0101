#!/bin/bash
# Round-end evidence on the GPU box (through gpurun): everything profiles/ holds for the round, from the tree as it is.
#   bash tools/gpu_final.sh [tag]      -> gpurun_out/prof_<tag>/out/* and gpurun_out/final_<tag>/*
TAG=${1:-r06}
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
# (windowed like the c2 file: the timed fits only, between the last warm-up refinement and the last timed one)
BUILD=$(python -c "import json; print(json.load(open('$F/bench_c3.json'))['build'])")
python tools/profile_collect.py --window /tmp/c3trace $F/kernel_stats_c3.csv "rocprofv3 --kernel-trace of \`python bench.py --config c3 --steps 10 --warmup 3 --no-cpu-baseline\` (SMPL-X-shaped model, 4096 instances, SMPLFIT_CHUNKS=1: 4096-instance launches), build \"$BUILD\"" 1 > $F/kernel_stats_c3.log 2>&1; cat $F/kernel_stats_c3.log
timeout 300 bash tools/pmc_sq.sh ${TAG}_smplx - smplx > $F/pmc_sq_smplx.log 2>&1 < /dev/null
timeout 300 bash tools/pmc_sq.sh ${TAG}_smpl - smpl > $F/pmc_sq_smpl.log 2>&1 < /dev/null
timeout 300 python tools/bench_callers.py > $F/bench_callers.txt 2>$F/bench_callers.err < /dev/null; tail -12 $F/bench_callers.txt
timeout 200 python tools/latency.py > $F/latency.json 2>$F/latency.err < /dev/null; tail -2 $F/latency.json | cut -c1-300
timeout 200 python tools/bench_skin.py > $F/bench_skin.json 2>/dev/null < /dev/null; cat $F/bench_skin.json
timeout 300 python tools/bench_general.py > $F/bench_general.json 2>/dev/null < /dev/null; cat $F/bench_general.json
# the general path: per-kernel statistics (300 betas at B = 256 and 2048, 32 betas at 4096) and the A/B against the
# vector-ALU form of the vertex block
{ for a in "smpl_b300" "smpl_b300 2048" "smpl_b32" "smpl_w12"; do echo "== $a"; timeout 300 bash tools/kstats_general.sh $a; done
  for a in "smpl_b300" "smpl_b32"; do echo "== $a SMPLFIT_GEN_MFMA=0"; SMPLFIT_GEN_MFMA=0 timeout 300 bash tools/kstats_general.sh $a; done; } > $F/kstats_general.txt 2>&1 < /dev/null
grep -A2 "^==" $F/kstats_general.txt | head -30
# matrix-pipe counters of the general accumulate kernel (300 betas, B = 256)
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmcgen && timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d /tmp/pmcgen -o p -- python $R/tools/bench_general.py smpl_b300 > /dev/null 2>&1 < /dev/null)
python - > $F/pmc_general.json <<'PY'
import csv, glob, json, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/pmcgen/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r'(k_[a-z_0-9]+)', r['Kernel_Name'])
        if m and m.group(1) in ('k_gen_accum_mfma', 'k_shape_solve', 'k_gen_lbs'):
            acc[m.group(1)][r['Counter_Name']].append(float(r['Counter_Value']))
res = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
for k, e in res.items():
    if e.get('SQ_BUSY_CU_CYCLES'):
        e['mfma_busy_frac_per_simd'] = e.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / e['SQ_BUSY_CU_CYCLES'] / 4
res['_note'] = 'rocprofv3 --pmc, tools/bench_general.py smpl_b300 (B = 256): per-launch averages'
print(json.dumps(res, indent=1))
PY
head -c 900 $F/pmc_general.json
SMPLFIT_LIB=build_ab/libwstamp.so timeout 200 python tools/wave_stamps.py 4096 > $F/wave_stamps_4096.txt 2>&1 < /dev/null; head -12 $F/wave_stamps_4096.txt
# round 6: the fused combine + solve kernel — phase stamps (debug build), event times against the two-kernel path, the
# overlap probe (pair-Gram under the residual pass), the result gather (world-1 RCCL: overlapped / in line / none)
{ for a in "4096 smpl" "16384 smpl" "4096 smplx"; do SMPLFIT_LIB=build_ab/libsstamp.so timeout 200 python tools/solve_stamps.py $a 2>&1 < /dev/null | grep -v amdgpu.ids; done; } > $F/solve_stamps.txt; head -9 $F/solve_stamps.txt
{ timeout 300 python tools/solve_ab_time.py smpl < /dev/null; timeout 300 python tools/solve_ab_time.py smplx 2048 4096 8192 < /dev/null; } 2>&1 | grep -v amdgpu.ids > $F/solve_ab_time.txt; cat $F/solve_ab_time.txt
timeout 200 python tools/overlap_probe.py smpl 4096 2>&1 < /dev/null | grep -v amdgpu.ids > $F/overlap_probe.txt; cat $F/overlap_probe.txt
timeout 300 python bench.py --steps 50 --warmup 5 --collective-always --no-cpu-baseline > $F/bench_collective.json 2>/dev/null < /dev/null; python -c "
import json
d=[json.loads(l) for l in open('$F/bench_collective.json') if l.startswith('{')][0]; print('collective-always', d['value'], d['ms_per_step'], json.dumps(d['multi_gpu'])[:900])"
# the general path's fixtures with the fp64 arbiter line, and the differentiable fit's rate
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -k "general_goldens" 2>&1 < /dev/null | grep "general\|passed\|failed" > $F/general_path_parity.txt; tail -4 $F/general_path_parity.txt
# the neighbour-fault reproducer of round 6 (tools/ubench/lds_neighbour_r6.hip, built here: build_ab/)
[ -x build_ab/lds_neighbour_r6 ] && timeout 300 build_ab/lds_neighbour_r6 > $F/ubench_lds_neighbour.txt 2>&1 < /dev/null; tail -3 $F/ubench_lds_neighbour.txt
