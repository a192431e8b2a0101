#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_evidence.py -m gpu -q -x -k "gemm_split or statistics or primitives" -s 2>&1 | grep -E "gemm\]|passed|failed|Error|assert" | head -20
python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3
for mode in f32 bf16x3; do for ch in 1 2; do
SMPLFIT_GEMM=$mode SMPLFIT_CHUNKS=$ch python tools/ab_fit.py smpl 4096 2>/dev/null | tail -1 | tee -a gpurun_out/c3_ab.jsonl
done; done
SMPLFIT_CHUNKS=3 python tools/ab_fit.py smpl 4096 2>/dev/null | tail -1 | tee -a gpurun_out/c3_ab.jsonl
python tools/ab_fit.py smpl 32768 2>/dev/null | tail -1 | tee -a gpurun_out/c3_ab.jsonl
