#!/bin/bash
# SQ issue / wait counters of the batch-major kernels and the split-bf16 GEMMs for one library build:
#   tools/pmc_sq.sh <tag> [lib.so|-] [smpl|smplx]     -> gpurun_out/pmc_sq_<tag>.json  (per-launch averages)
TAG=$1; LIB=$2; KIND=${3:-smpl}
[ "$LIB" = "-" ] && LIB=
R=$GRAFT_REPO_ROOT; OUT=/tmp/pmcsq_$TAG; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
[ -n "$LIB" ] && export SMPLFIT_LIB=$R/$LIB
export SMPLFIT_CHUNKS=1
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
P2="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT"
P3="GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_FLAT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"
P4="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/p$i -o p -- python $R/tools/ab_fit.py $KIND 4096 > $OUT/p$i.log 2>&1 < /dev/null
done
python - $OUT $TAG <<'PY'
import csv, glob, json, collections, re, sys, os
out, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f'{out}/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r'(k_[a-z_0-9]+)', r['Kernel_Name'])
        k = m.group(1) if m else r['Kernel_Name'][:40]
        if k in ('k_residual_bm', 'k_lbs_partsum_bm', 'k_pair_gram_bm', 'k_posedirs_gemm_bf16x3', 'k_posedirs_gemm_bf16x3_tiled'):
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
res = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
res['_launches'] = {k: len(next(iter(d.values()))) for k, d in acc.items()}
json.dump(res, open(os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out', f'pmc_sq_{tag}.json'), 'w'), indent=1)
for k, d in res.items():
    print(k, json.dumps(d))
PY
tail -3 $OUT/p1.log
