#!/bin/bash
# average latency per instruction class of the batch-major kernels: LEVEL counters / instruction counts
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp; export SMPLFIT_BM=1 SMPLFIT_CHUNKS=1
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmclat_$name -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmclat_$name.log 2>&1; }
run a SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY
python $R/tools/pmc_summary_bm.py $R/gpurun_out/pmclat_
