import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from smplfitter_amd import synth
from smplfitter_amd.pt import BodyFitter, BodyModel
dev = torch.device('cuda:0')
root = synth.ensure_model_root(kinds=('smpl',))
m = BodyModel('smpl', 'neutral', model_root=f'{root}/smpl', num_betas=10, device=dev)
f = BodyFitter(m)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rs = np.random.RandomState(42)
pose = torch.from_numpy((rs.randn(B, 72) * 0.1).astype(np.float32)).to(dev)
betas = torch.from_numpy((rs.randn(B, 10) * 0.5).astype(np.float32)).to(dev)
trans = torch.from_numpy(rs.randn(B, 3).astype(np.float32)).to(dev)
fw = m(pose, betas, trans)
fw2 = m(pose, betas, trans)
print('forward deterministic:', torch.equal(fw['vertices'], fw2['vertices']))
tv, tj = fw['vertices'].contiguous(), fw['joints'].contiguous()
rs_ = [f.fit(tv, tj, num_iter=3, beta_regularizer=1.0) for _ in range(4)]
torch.cuda.synchronize()
for i in range(1, 4):
    d = (rs_[i]['pose_rotvecs'] - rs_[0]['pose_rotvecs']).abs().amax(1)
    bad = torch.nonzero(d > 0).flatten().cpu().numpy()
    print(f'run {i}: {len(bad)} instances differ', bad[:20], 'max', float(d.max()))
for it in (1, 2):
    a = f.fit(tv, tj, num_iter=it, beta_regularizer=1.0, final_adjust_rots=False)
    b = f.fit(tv, tj, num_iter=it, beta_regularizer=1.0, final_adjust_rots=False)
    d = (a['pose_rotvecs'] - b['pose_rotvecs']).abs().amax(1)
    print(f'num_iter={it} nfa: differ', int((d > 0).sum()), torch.nonzero(d > 0).flatten().cpu().numpy()[:20])
