#!/bin/bash
# PMC passes for the batch-major kernels
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp; export SMPLFIT_BM=1 SMPLFIT_CHUNKS=1
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmcbm_$name -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmcbm_$name.log 2>&1; }
run a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run b SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_SMEM
run c SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INST_LEVEL_LDS
python $R/tools/pmc_summary_bm.py $R/gpurun_out/pmcbm_
