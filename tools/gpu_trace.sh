#!/bin/bash
# kernel timeline of the default fit with N chunks: tools/gpu_trace.sh <tag> <chunks> [extra env, e.g. TRACE_B=32]
tag=$1; ch=$2; shift 2
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/trace_$tag; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for e in "$@"; do export "$e"; done
SMPLFIT_CHUNKS=$ch rocprofv3 --kernel-trace --output-format csv -d $OUT/raw -o t -- python $R/tools/fit_only.py ${TRACE_B:-4096} 8 > $OUT/log.txt 2>&1
cd $R; python tools/trace_timeline.py $OUT/raw -2 > $OUT/timeline.txt; rm -rf $OUT/raw; head -3 $OUT/timeline.txt
