#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/c2_tests.log 2>&1; echo "tests rc=$?" | tee gpurun_out/c2_rc.txt
tail -8 gpurun_out/c2_tests.log
R=$PWD; cd /tmp; export TMPDIR=/tmp
SMPLFIT_CHUNKS=1 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/c2_trace1 -- python $R/tools/ab_fit.py smpl 4096 > $R/gpurun_out/c2_trace1.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/c2_trace2 -- python $R/tools/ab_fit.py smpl 4096 > $R/gpurun_out/c2_trace2.log 2>&1
cd $R
python tools/trace_summary.py gpurun_out/c2_trace1 | tee gpurun_out/c2_trace1_summary.txt
python tools/trace_summary.py gpurun_out/c2_trace2 | tee gpurun_out/c2_trace2_summary.txt
rm -rf gpurun_out/c2_trace1 gpurun_out/c2_trace2
for ch in 2 3 4; do for kb in 0 84; do
SMPLFIT_CHUNKS=$ch SMPLFIT_GEMM_LDS_KB=$kb python tools/ab_fit.py smpl 4096 2>/dev/null | tail -1 | tee -a gpurun_out/c2_ab.jsonl
done; done
SMPLFIT_CHUNKS=1 SMPLFIT_GEMM_LDS_KB=84 python tools/ab_fit.py smpl 4096 2>/dev/null | tail -1 | tee -a gpurun_out/c2_ab.jsonl
