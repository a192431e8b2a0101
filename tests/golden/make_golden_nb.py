"""Generate ``golden_nb_smpl.npz`` by running the REFERENCE (build container only, like make_golden.py):
fits and forward passes with num_betas = 6 (the ``smpl`` fixture) and 13 (``smpl_b16``: the same construction
with 16 shape directions, synth.make_model_arrays('smpl_b16')) — neither is a count the kernels are
instantiated for: the library pads the shape unknowns up to 10 / 16 — without and with the kid blend shape.
B = 8; the targets are forward passes of each model at random parameters.

Usage:  python tests/golden/make_golden_nb.py
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
NB_DIR = {6: 'smpl', 13: 'smpl_b16'}


def main():
    torch.set_num_threads(8)
    root = synth.ensure_model_root(kinds=('smpl', 'smpl_b16'), seed=0)
    out = {}
    rs = np.random.RandomState(77)
    with torch.no_grad():
        for nb in (6, 13):
            model = ref.BodyModel('smpl', 'neutral', model_root=f'{root}/{NB_DIR[nb]}', num_betas=nb)
            assert model.num_betas == nb
            J = model.num_joints
            pose = (rs.randn(B, 3 * J) * 0.1).astype(np.float32)
            betas = (rs.randn(B, nb) * 0.5).astype(np.float32)
            trans = rs.randn(B, 3).astype(np.float32)
            fw = model(torch.from_numpy(pose), torch.from_numpy(betas), torch.from_numpy(trans))
            out[f'nb{nb}.pose'], out[f'nb{nb}.betas'], out[f'nb{nb}.trans'] = pose, betas, trans
            out[f'nb{nb}.fwd_vertices_every_50th'] = fw['vertices'].numpy()[:, ::50]
            out[f'nb{nb}.fwd_joints'] = fw['joints'].numpy()
            # targets: another forward pass of the same model (full 10 / 16 betas' worth of shape is not needed)
            p2 = (rs.randn(B, 3 * J) * 0.1).astype(np.float32)
            b2 = (rs.randn(B, nb) * 0.5).astype(np.float32)
            t2 = rs.randn(B, 3).astype(np.float32)
            fw2 = model(torch.from_numpy(p2), torch.from_numpy(b2), torch.from_numpy(t2))
            tv, tj = fw2['vertices'], fw2['joints']
            out[f'nb{nb}.target_vertices'], out[f'nb{nb}.target_joints'] = tv.numpy(), tj.numpy()
            for kid in (False, True):
                fitter = ref.BodyFitter(model, enable_kid=kid)
                for name, kw in (('it3_reg1', dict(num_iter=3, beta_regularizer=1.0)),
                                 ('it2_reg0', dict(num_iter=2, beta_regularizer=0.0, final_adjust_rots=False))):
                    keys = ['pose_rotvecs', 'shape_betas', 'trans'] + (['kid_factor'] if kid else [])
                    r = fitter.fit(tv, tj, requested_keys=keys, **kw)
                    for k in keys:
                        out[f'nb{nb}.kid{int(kid)}.{name}.{k}'] = r[k].numpy()
    path = osp.join(HERE, 'golden_nb_smpl.npz')
    np.savez_compressed(path, **out)
    print(path, f'{os.path.getsize(path) / 1e6:.2f} MB', len(out), 'arrays')


if __name__ == '__main__':
    main()
