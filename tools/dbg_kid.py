import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smplfitter_amd import synth
from smplfitter_amd.pt import BodyFitter, BodyModel
dev = torch.device('cuda:0')
root = synth.ensure_model_root(kinds=('smpl',))
m = BodyModel('smpl', 'neutral', model_root=f'{root}/smpl', num_betas=10, device=dev)
J = m.num_joints; B = 8
rs = np.random.RandomState(3)
pose = torch.from_numpy((rs.randn(B, 3 * J) * 0.1).astype(np.float32)).to(dev)
betas = torch.from_numpy((rs.randn(B, 10) * 0.5).astype(np.float32)).to(dev)
trans = torch.from_numpy(rs.randn(B, 3).astype(np.float32)).to(dev)
kid = torch.from_numpy((rs.rand(B) * 0.5).astype(np.float32)).to(dev)
fw = m(pose, betas, trans, kid_factor=kid)
tv, tj = fw['vertices'].contiguous(), fw['joints'].contiguous()
f = BodyFitter(m, enable_kid=True)
for it, fa in ((1, False), (1, True), (3, True)):
    r = f.fit(tv, tj, num_iter=it, beta_regularizer=1.0, final_adjust_rots=fa, requested_keys=['pose_rotvecs', 'shape_betas', 'trans', 'kid_factor'])
    fw2 = m(r['pose_rotvecs'], r['shape_betas'], r['trans'], kid_factor=r['kid_factor'])
    print(os.environ.get('SMPLFIT_BM', '1'), it, fa, 'roundtrip max', float((fw2['vertices'] - tv).norm(dim=-1).max()), 'betas', r['shape_betas'][0, :4].tolist(), 'kid', r['kid_factor'][:3].tolist())
