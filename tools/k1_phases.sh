#!/bin/bash
# K1 (k_joint_stage) per mode: rotations only (part_rotations entry point), prologue only (shape-solve
# entry point), both (inside a fit)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
cat > /tmp/k1.py <<'PY'
import sys, os
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import numpy as np, torch
from smplfitter_amd import synth
from smplfitter_amd.pt import BodyFitter, BodyModel
dev = torch.device('cuda:0'); root = synth.ensure_model_root(kinds=('smpl',))
m = BodyModel('smpl', 'neutral', model_root=f'{root}/smpl', num_betas=10, device=dev); f = BodyFitter(m)
B = 4096; rs = np.random.RandomState(1); t = lambda a: torch.from_numpy(a.astype(np.float32)).to(dev)
fw = m(t(rs.randn(B, 72) * 0.1), t(rs.randn(B, 10) * 0.5), t(rs.randn(B, 3)))
mode = sys.argv[1]
for _ in range(6):
    if mode == 'rot':
        f._part_rotations(fw['vertices'], fw['joints'])
    elif mode == 'pro':
        f._shape_solve(fw['orientations'], fw['vertices'], fw['joints'], want_mesh=False)
    else:
        f.fit(fw['vertices'], fw['joints'], num_iter=1)
torch.cuda.synchronize()
PY
for mode in rot pro fit; do
rm -rf $R/gpurun_out/k1_$mode
SMPLFIT_CHUNKS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/k1_$mode -- python /tmp/k1.py $mode > /dev/null 2>&1
f=$(find $R/gpurun_out/k1_$mode -name "*kernel_stats.csv" | head -1)
echo $mode $(grep k_joint_stage $f | awk -F, '{print "calls", $2, "avg_us", $4/1000}')
done
