"""Which workspace region differs between two identical fit calls?  (replicates carve() of smplfit_hip.hip)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from smplfitter_amd import synth
from smplfitter_amd.pt import BodyFitter, BodyModel
dev = torch.device('cuda:0')
root = synth.ensure_model_root(kinds=('smpl',))
m = BodyModel('smpl', 'neutral', model_root=f'{root}/smpl', num_betas=10, device=dev)
f = BodyFitter(m)
B = 4096
rs = np.random.RandomState(42)
pose = torch.from_numpy((rs.randn(B, 72) * 0.1).astype(np.float32)).to(dev)
betas = torch.from_numpy((rs.randn(B, 10) * 0.5).astype(np.float32)).to(dev)
trans = torch.from_numpy(rs.randn(B, 3).astype(np.float32)).to(dev)
fw = m(pose, betas, trans)
tv, tj = fw['vertices'].contiguous(), fw['joints'].contiguous()
h = m._native(dev)
info = h.info
Vp, J, S = info.padded_vertices, info.num_joints, 10
NE = S * (S + 1) // 2 + S + 3 * S + 3
ngroups = len(h.table('vertex_groups')) // 5
def au(x, a): return (x + a - 1) // a * a
def regions(Bc):
    Mp = au(Bc, 128); Kp = 208; jds = 52
    r = [('tvs', Bc*3*Vp*4), ('vws', Bc*Vp*4), ('vposed', Mp*3*Vp*4), ('rp', Mp*Kp*4), ('mean', Bc*12), ('tjc', Bc*J*12),
         ('psum', Bc*J*16*4), ('G', Bc*J*36), ('jd', Bc*J*jds*4), ('pext', Bc*J*3*(S+1)*4), ('gramj', Bc*(NE+1)*4),
         ('gramv', Bc*(NE+1)*8), ('beta', Bc*S*4), ('trans', Bc*12), ('jb', Bc*J*16), ('jbT', Mp*J*16), ('rjoints', Bc*J*12), ('rverts', Bc*3*Vp*4),
         ('tjreg', Bc*J*12), ('rjreg', Bc*J*12), ('mbj', Bc*J*12), ('scale', Bc*4), ('regref', Bc*S*4), ('cen', (Bc+1)*(S*S+S)*8),
         ('vextra', Bc*32*4), ('beta_out', Bc*S*4), ('tjs', Bc*J*12), ('vpT', Mp*3*Vp*4), ('tT', Mp*3*Vp*4),
         ('psumP', ngroups*16*Mp*4), ('resP', ngroups*52*Mp*4), ('gramP', 32*NE*Mp*4), ('jdT', Mp*au(J*jds, 64)*4)]
    out, off = [], 0
    for n, sz in r:
        out.append((n, off, sz)); off = au(off + sz, 256)
    return out, off
nbytes = h.workspace_bytes(B)
ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
kw = dict(num_iter=int(os.environ.get('NI', 2)), beta_regularizer=1.0, final_adjust_rots=os.environ.get('FA', '0') == '1', _workspace=ws)
snaps = []
for rep in range(6):
    r = f.fit(tv, tj, **kw); torch.cuda.synchronize()
    snaps.append((ws.clone(), r['pose_rotvecs'].clone()))
chunks = 2 if os.environ.get('SMPLFIT_CHUNKS', '2') != '1' else 1
Bc = B // chunks
regs, per = regions(Bc)
print('workspace', nbytes, 'per chunk', per, 'x', chunks)
for rep in range(1, 6):
    d = (snaps[rep][0] != snaps[0][0])
    if not bool(d.any()):
        print(f'rep {rep}: identical'); continue
    print(f'rep {rep}: pose differs for', int(((snaps[rep][1] - snaps[0][1]).abs().amax(1) > 0).sum()), 'instances')
    for c in range(chunks):
        for n, off, sz in regs:
            seg = d[c*per + off: c*per + off + sz]
            k = int(seg.sum())
            if k:
                idx = torch.nonzero(seg).flatten()
                print(f'   chunk {c} {n:8s}: {k} bytes differ, first at +{int(idx[0])} last +{int(idx[-1])} (elem {int(idx[0])//4})')

# ---- read-before-write hunt: poison one region at a time with NaN bytes before the call
print('== poison test')
kw['_workspace'] = ws
ref = f.fit(tv, tj, **kw); torch.cuda.synchronize()
ref = {k: v.clone() for k, v in ref.items()}
for c in range(chunks):
    for n, off, sz in regs:
        ws[c*per + off: c*per + off + sz] = 0xFF
        r = f.fit(tv, tj, **kw); torch.cuda.synchronize()
        bad = int(((r['pose_rotvecs'] != ref['pose_rotvecs']).any(1) | (r['shape_betas'] != ref['shape_betas']).any(1)).sum())
        if bad:
            print(f'   poisoning chunk {c} {n:8s} changes {bad} instances; nan: {bool(torch.isnan(r["pose_rotvecs"]).any())}')
        r = f.fit(tv, tj, **kw); torch.cuda.synchronize()   # restore a clean state
print('== all poisoned')
ws[:] = 0xFF
r = f.fit(tv, tj, **kw); torch.cuda.synchronize()
print('   differs from ref:', int((r['pose_rotvecs'] != ref['pose_rotvecs']).any(1).sum()), 'nan', bool(torch.isnan(r['pose_rotvecs']).any()))
