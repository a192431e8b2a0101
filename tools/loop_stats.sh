#!/bin/bash
# Statistics of the innermost (last) loop of the batch-major vertex kernels as hipcc compiles them (no GPU):
#   tools/loop_stats.sh [extra -D flags]      -> /tmp/dev_loop.s holds the device assembly
set -e
cd /tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -DSMPLFIT_BUILD_ID='"x"' "$@" --cuda-device-only -S /root/repo/smplfitter_amd/csrc/smplfit_hip.hip -o /tmp/dev_loop.s 2>&1 | grep -v "hip-link" | grep -A3 error || true
for k in "16k_lbs_partsum_bmILi10ELi4ELb0ELb0E" "16k_lbs_partsum_bmILi11ELi4ELb0ELb0E" "16k_lbs_partsum_bmILi10ELi4ELb1ELb0E" "13k_residual_bmILi10E" "13k_residual_bmILi11E"; do
  awk -v k="$k" 'index($0, "_ZN12_GLOBAL__N_1" k) == 1 && /:/ {p=1} p{print} /\.Lfunc_end[0-9]+:/{if(p){exit}}' dev_loop.s > k_$k.s
  # the vertex loop = the longest innermost loop of the kernel
  L=$(awk '/Inner Loop/{s=NR} /s_cbranch_scc/{if(s && NR-s>best){best=NR-s; bl=s} s=0} END{print bl}' k_$k.s)
  awk -v l=$L 'NR>=l' k_$k.s | awk '/s_cbranch_scc/{print; exit} {print}' > kl_$k.s
  echo "$k: loop lines $(wc -l < kl_$k.s) | vmcnt waits [$(grep -o 'vmcnt([0-9]*)' kl_$k.s | tr '\n' ' ')] | vmov $(grep -c 'v_mov_b32\|v_pk_mov' kl_$k.s) | scratch $(grep -c scratch_ k_$k.s) | valu $(grep -c '^\sv_' kl_$k.s) ds $(grep -c 'ds_read' kl_$k.s) sload $(grep -c 's_load' kl_$k.s) vmem $(grep -c 'global_load\|global_store' kl_$k.s) lgkm-waits $(grep -c 'lgkmcnt' kl_$k.s) | vgpr $(grep "$k" dev_loop.s | grep -o 'num_vgpr, [0-9]*' | head -1)"
done
