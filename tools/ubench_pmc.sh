#!/bin/bash
# Hardware counters of the fused posedirs-GEMM + residual micro-benchmark (tools/ubench/fused_tile.hip) beside the same
# two loops unfused, on the GPU box:   bash tools/ubench_pmc.sh   -> gpurun_out/ubench_fused_pmc.json
# One rocprofv3 pass per counter group (kernel trace only beside the counters); every launch of the benchmark is kept
# apart by (kernel, grid, workgroup) so that the variants the program times one after the other are not averaged together.
R=$GRAFT_REPO_ROOT; OUT=/tmp/ubpmc; rm -rf $OUT; mkdir -p $OUT $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $R/tools/ubench/fused_tile.hip -o /tmp/fused_tile 2> $OUT/build.log || { tail $OUT/build.log; exit 1; }
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
P2="SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
P3="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"
P4="FETCH_SIZE"
P5="WRITE_SIZE"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/p$i -o p -- /tmp/fused_tile > $OUT/p$i.log 2>&1 < /dev/null
done
python - $OUT <<'PY'
import csv, glob, json, collections, re, sys, os
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f'{out}/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r'^void ', '', r['Kernel_Name']).split('(')[0]
        key = f"{n} grid {r.get('Grid_Size', '?')} wg {r.get('Workgroup_Size', '?')}"
        acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
res = {}
for k, d in sorted(acc.items()):
    e = {c: sum(v) / len(v) for c, v in d.items()}
    e['_launches'] = len(next(iter(d.values())))
    if 'FETCH_SIZE' in e and 'WRITE_SIZE' in e:
        e['hbm_MB'] = (2 * e['FETCH_SIZE'] + e['WRITE_SIZE']) * 1024 / 1e6   # the guide's gfx950 correction, as pmc_traffic.json
    if e.get('SQ_BUSY_CU_CYCLES'):
        for c in ('SQ_VALU_MFMA_BUSY_CYCLES',):
            if c in e: e['mfma_busy_frac'] = e[c] / e['SQ_BUSY_CU_CYCLES'] / 4   # per SIMD
    if e.get('SQ_WAVE_CYCLES'):
        for c, nm in (('SQ_WAIT_INST_LDS', 'wait_lds_frac'), ('SQ_WAIT_ANY', 'wait_any_frac'), ('SQ_ACTIVE_INST_VALU', 'valu_active_frac')):
            if c in e: e[nm] = e[c] / e['SQ_WAVE_CYCLES'] * (4 if c.startswith('SQ_ACTIVE') else 1)
    if e.get('SQ_LDS_IDX_ACTIVE'):
        e['lds_conflict_frac'] = e.get('SQ_LDS_BANK_CONFLICT', 0) / e['SQ_LDS_IDX_ACTIVE']
    res[k] = e
json.dump(res, open(os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out', 'ubench_fused_pmc.json'), 'w'), indent=1)
for k, e in res.items():
    print(k[:110]); print('   ', {c: (round(v, 3) if v < 100 else int(v)) for c, v in e.items() if c.startswith(('hbm', 'mfma', 'wait', 'valu', 'lds', '_l'))})
PY
