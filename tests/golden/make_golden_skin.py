"""Generate ``golden_<kind>.npz (kind = smpl_w6, smplx_w6, smpl_rnd)`` by running the REFERENCE (build container only, like make_golden.py).

Skinning variants of the synthetic models (``synth.make_model_arrays``):

* ``smpl_w6`` / ``smplx_w6``: SIX non-zero skinning weights per vertex.  The reference blends with the dense (V, J)
  weight matrix and has no cap on the non-zeros (pt/bodyfitter.py:1000-1003, pt/bodymodel.py:80-93); the library stores
  such models with eight (joint, weight) pairs per vertex (KW = 8 instantiations of the wave-per-instance kernels).
* ``smpl_rnd``: four weights, the three minor ones on RANDOM joints — joint sets (and joint pairs) the distance-based
  construction never produces: pins the vertex pieces / cells / residual segments of the batch-major tables.

B = 8; the grid of make_golden.py's non-default kinds (joints given / omitted, weights, final adjustment) plus the kid
unknown and the forward pins.

Usage:  python tests/golden/make_golden_skin.py
"""

import os
import os.path as osp
import sys

import numpy as np
import torch

HERE = osp.dirname(osp.abspath(__file__))
sys.path.insert(0, osp.join(HERE, '..', '..'))
sys.path.insert(0, '/root/reference/src')

import smplfitter.pt as ref  # noqa: E402
from smplfitter_amd import synth  # noqa: E402

B = 8
KINDS = ('smpl_w6', 'smplx_w6', 'smpl_rnd')
GRID = [(3, 1.0, True, False, True), (1, 0.0, True, False, False), (3, 1.0, True, True, True),
        (2, 1.0, False, False, True), (3, 0.0, False, True, False)]


def cfg_name(num_iter, beta_reg, joints, weights, final):
    return f'it{num_iter}_reg{int(beta_reg)}_{"j" if joints else "nj"}_{"w" if weights else "nw"}_{"fa" if final else "nfa"}'


def make_kind(kind, root):
    arrs = synth.make_model_arrays(kind, seed=0)
    base = 'smplx' if kind.startswith('smplx') else 'smpl'
    model = ref.BodyModel(base, 'neutral', model_root=f'{root}/{kind}', num_betas=10)
    fitter = ref.BodyFitter(model)
    J = model.num_joints
    rs = np.random.RandomState(2468)
    pose = (rs.randn(B, 3 * J) * 0.1).astype(np.float32)
    betas = (rs.randn(B, 10) * 0.5).astype(np.float32)
    trans = rs.randn(B, 3).astype(np.float32)
    vw = rs.uniform(0.5, 1.5, size=(B, model.num_vertices)).astype(np.float32)
    jw = rs.uniform(0.5, 1.5, size=(B, J)).astype(np.float32)
    out = dict(pose=pose, betas=betas, trans=trans, vertex_weights=vw, joint_weights=jw,
               model_sha256=np.array(synth.model_sha256(arrs)),
               skin_nnz=np.array(int((arrs['weights'] != 0).sum(1).max())))
    with torch.no_grad():
        fw = model(torch.from_numpy(pose), torch.from_numpy(betas), torch.from_numpy(trans))
        tv, tj = fw['vertices'], fw['joints']
        out['target_vertices'], out['target_joints'] = tv.numpy(), tj.numpy()
        out['fwd_joints'], out['fwd_orientations'] = tj.numpy(), fw['orientations'].numpy()
        for num_iter, reg, joints, weights, final in GRID:
            r = fitter.fit(tv, tj if joints else None,
                           vertex_weights=torch.from_numpy(vw) if weights else None,
                           joint_weights=torch.from_numpy(jw) if (weights and joints) else None,
                           num_iter=num_iter, beta_regularizer=reg, final_adjust_rots=final,
                           requested_keys=['pose_rotvecs', 'shape_betas', 'trans'])
            c = cfg_name(num_iter, reg, joints, weights, final)
            for k in ('pose_rotvecs', 'shape_betas', 'trans', 'orientations'):
                out[f'fit.{c}.{k}'] = r[k].numpy()
        kid = (rs.randn(B) * 0.3).astype(np.float32)
        out['kid'] = kid
        fwk = model(torch.from_numpy(pose), torch.from_numpy(betas), torch.from_numpy(trans), kid_factor=torch.from_numpy(kid))
        out['kid.target_vertices'], out['kid.target_joints'] = fwk['vertices'].numpy(), fwk['joints'].numpy()
        r = ref.BodyFitter(model, enable_kid=True).fit(fwk['vertices'], fwk['joints'], num_iter=3, beta_regularizer=1.0,
                                                       requested_keys=['pose_rotvecs', 'shape_betas', 'trans'])
        for k in ('pose_rotvecs', 'shape_betas', 'trans', 'kid_factor'):
            out[f'kidfit.a.{k}'] = r[k].numpy()
    path = osp.join(HERE, f'golden_{kind}.npz')
    np.savez_compressed(path, **out)
    print(path, f'{os.path.getsize(path) / 1e6:.2f} MB', len(out), 'arrays')


def main():
    torch.set_num_threads(8)
    root = synth.ensure_model_root(kinds=KINDS, seed=0)
    for kind in KINDS:
        make_kind(kind, root)


if __name__ == '__main__':
    main()
