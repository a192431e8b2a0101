"""Generate ``golden_smplxfat.npz`` by running the REFERENCE (build container only, like make_golden.py).

The fat-part SMPL-X-shaped fixture of SURVEY.md Appendix C: same 55-joint tree as the ``smplx`` fixture but
with 3.5 cm finger / face parts (``synth.make_model_arrays('smplx_fat')``), on which the bone parts' twist
is well conditioned, so ``pose_rotvecs`` of two correct fp32 implementations agree to ~1e-4 and a tight
pose gate is meaningful (the thin ``smplx`` fixture is judged on vertices only).  B = 8.

Usage:  python tests/golden/make_golden_fat.py
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


def main():
    torch.set_num_threads(8)
    root = synth.ensure_model_root(kinds=('smplx_fat',), seed=0)
    arrs = synth.make_model_arrays('smplx_fat', seed=0)
    model = ref.BodyModel('smplx', 'neutral', model_root=f'{root}/smplx_fat', num_betas=10)
    fitter = ref.BodyFitter(model)
    J = model.num_joints
    rs = np.random.RandomState(4321)
    pose = (rs.randn(B, 3 * J) * 0.1).astype(np.float32)
    betas = (rs.randn(B, 10) * 0.5).astype(np.float32)
    trans = rs.randn(B, 3).astype(np.float32)
    out = dict(pose=pose, betas=betas, trans=trans, model_sha256=np.array(synth.model_sha256(arrs)))
    with torch.no_grad():
        fw = model(torch.from_numpy(pose), torch.from_numpy(betas), torch.from_numpy(trans))
        tv, tj = fw['vertices'], fw['joints']
        out['target_vertices'], out['target_joints'] = tv.numpy(), tj.numpy()
        out['fwd_joints'], out['fwd_orientations'] = tj.numpy(), fw['orientations'].numpy()
        for name, kw in (('it3_reg1_j_nw_fa', dict(num_iter=3, beta_regularizer=1.0)),
                         ('it1_reg0_j_nw_nfa', dict(num_iter=1, beta_regularizer=0.0, final_adjust_rots=False)),
                         ('it2_reg1_j_nw_fa', dict(num_iter=2, beta_regularizer=1.0))):
            r = fitter.fit(tv, tj, requested_keys=['pose_rotvecs', 'shape_betas', 'trans'], **kw)
            for k in ('pose_rotvecs', 'shape_betas', 'trans', 'orientations'):
                out[f'fit.{name}.{k}'] = r[k].numpy()
    path = osp.join(HERE, 'golden_smplxfat.npz')
    np.savez_compressed(path, **out)
    print(path, f'{os.path.getsize(path) / 1e6:.2f} MB', len(out), 'arrays')


if __name__ == '__main__':
    main()
