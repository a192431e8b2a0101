cd /tmp && export TMPDIR=/tmp
cat > /tmp/c3.py <<'PY'
import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from smplfitter_amd import synth
from smplfitter_amd.pt import BodyFitter, BodyModel
dev = torch.device('cuda:0')
root = synth.ensure_model_root(kinds=('smplx',))
m = BodyModel('smplx', 'neutral', model_root=f'{root}/smplx', num_betas=10, device=dev)
f = BodyFitter(m)
rs = np.random.RandomState(42); B = 4096; J = m.num_joints
t = lambda a: torch.from_numpy(a.astype(np.float32)).to(dev)
fw = m(t(rs.randn(B, 3 * J) * 0.1), t(rs.randn(B, 10) * 0.5), t(rs.randn(B, 3)))
for _ in range(6):
    f.fit(fw['vertices'], fw['joints'], num_iter=3)
torch.cuda.synchronize()
PY
SMPLFIT_CHUNKS=1 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_c3 -- python /tmp/c3.py > /dev/null 2>&1
python - <<'PY'
import csv,glob,os
f=sorted(glob.glob('/root/repo/gpurun_out/prof_c3/**/*kernel_stats.csv',recursive=True), key=os.path.getmtime)[-1]
for r in list(csv.DictReader(open(f)))[:14]:
    print(r['Name'][:62], r['Calls'], round(float(r['AverageNs'])/1e3,1), r['Percentage'])
PY
