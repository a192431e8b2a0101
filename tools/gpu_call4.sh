#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "full_size_properties and batch-major and smpl-4096" 2>&1 | grep -E "^E|assert|Error|passed|failed" | head -30
for ch in 2 3; do for kb in 0 84; do
SMPLFIT_CHUNKS=$ch SMPLFIT_GEMM_LDS_KB=$kb python tools/ab_fit.py smpl 4096 2>/dev/null | tail -1 | tee -a gpurun_out/c4_ab.jsonl
done; done
