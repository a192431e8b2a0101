"""N default fits of a batch and nothing else (for counter passes): python tools/fit_only.py <B> <nfits> [smpl|smplx] [weights|kid|nojoints]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from smplfitter_amd import synth
from smplfitter_amd.pt import BodyFitter, BodyModel

B, n = int(sys.argv[1]), int(sys.argv[2])
kind = sys.argv[3] if len(sys.argv) > 3 else 'smpl'
dev = torch.device('cuda:0')
root = synth.ensure_model_root(kinds=(kind,))
model = BodyModel(kind, 'neutral', model_root=f'{root}/{kind}', num_betas=10, device=dev)
fitter = BodyFitter(model, enable_kid=len(sys.argv) > 4 and sys.argv[4] == 'kid')
J = model.num_joints
rs = np.random.RandomState(42)
t = lambda a: torch.from_numpy(a.astype(np.float32)).to(dev)
fw = model(t(rs.randn(B, 3 * J) * 0.1), t(rs.randn(B, 10) * 0.5), t(rs.randn(B, 3)))
tv, tj = fw['vertices'].contiguous(), fw['joints'].contiguous()
ws = torch.empty(model._native(dev, kid=len(sys.argv) > 4 and sys.argv[4] == 'kid').workspace_bytes(B), dtype=torch.uint8, device=dev)
kw = {}
if len(sys.argv) > 4 and sys.argv[4] == 'weights':
    kw = dict(vertex_weights=torch.rand(B, model.num_vertices, device=dev) + 0.5, joint_weights=torch.rand(B, J, device=dev) + 0.5)
torch.cuda.synchronize()
for _ in range(n):
    fitter.fit(tv, None if (len(sys.argv) > 4 and sys.argv[4] == 'nojoints') else tj, **kw, num_iter=3, beta_regularizer=1.0, requested_keys=['pose_rotvecs', 'shape_betas', 'trans'], _workspace=ws)
torch.cuda.synchronize()
