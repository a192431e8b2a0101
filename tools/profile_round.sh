#!/bin/bash
# rocprofv3 evidence for profiles/: kernel stats (1 chunk and default), HBM traffic counters (1 chunk)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
SMPLFIT_CHUNKS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_i1 -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_i1.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_i2 -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_i2.json 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
SMPLFIT_CHUNKS=1 timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmci_$c -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmci_$c.log 2>&1
done
cat $R/gpurun_out/prof_i1.json | cut -c1-200
