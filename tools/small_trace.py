"""Kernel timeline of ONE fit call at a small batch (rocprofv3 --kernel-trace -- python tools/small_trace.py B)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from smplfitter_amd import synth
from smplfitter_amd.pt import BodyFitter, BodyModel
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device('cuda:0')
root = synth.ensure_model_root(kinds=('smpl',))
model = BodyModel('smpl', 'neutral', model_root=f'{root}/smpl', num_betas=10, device=dev)
J = model.num_joints
rs = np.random.RandomState(1)
pose = torch.from_numpy((rs.randn(B, 3 * J) * 0.1).astype(np.float32)).to(dev)
betas = torch.from_numpy((rs.randn(B, 10) * 0.5).astype(np.float32)).to(dev)
trans = torch.from_numpy(rs.randn(B, 3).astype(np.float32)).to(dev)
fw = model(pose, betas, trans)
tv, tj = fw['vertices'].contiguous(), fw['joints'].contiguous()
f = BodyFitter(model)
for _ in range(30):
    f.fit(tv, tj, num_iter=3, beta_regularizer=1.0, requested_keys=['pose_rotvecs', 'shape_betas', 'trans'])
torch.cuda.synchronize()
