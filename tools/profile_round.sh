#!/bin/bash
# rocprofv3 evidence for profiles/ (run on the GPU box through gpurun): kernel-trace stats of the default
# bench command with one chunk (the default for the SMPL-shaped model: 4096-instance launches) and with two chunks,
# then the HBM traffic counters (separate --pmc passes, kernel-trace only) and the matrix-pipe counters.
# Everything lands in gpurun_out/prof_<tag>/ ; tools/profile_collect.py turns it into the files of profiles/.
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs"
SMPLFIT_CHUNKS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace1 -o t -- $B > $OUT/bench_chunks1.json 2>/dev/null
SMPLFIT_CHUNKS=2 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace2 -o t -- $B > $OUT/bench_chunks2_profiled.json 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  SMPLFIT_CHUNKS=1 timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > $OUT/pmc_$c.log 2>&1
done
SMPLFIT_CHUNKS=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/pmc_MFMA -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > $OUT/pmc_MFMA.log 2>&1
cd $R
python tools/profile_collect.py $OUT $TAG
# the default line LAST, with this build's traffic figures in place (bench.py reads profiles/pmc_traffic.json)
cp $OUT/out/pmc_traffic.json profiles/pmc_traffic.json
python bench.py --steps 20 --warmup 5 > $OUT/out/${TAG}_bench_default.json 2>/dev/null
# keep the merge-back small: the raw traces stay on the box
rm -rf $OUT/trace1 $OUT/trace2 $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_MFMA
