"""Generate ``golden_general.npz`` by running the REFERENCE (build container only, like make_golden.py): models the
library serves on its GENERAL path (``smplfit_info.vertex_path == SMPLFIT_PATH_GENERAL``):

* ``smpl_b32``  — 32 betas (``num_betas=32``): more shape unknowns than the templated kernels' 16;
* ``smpl_b300`` — ``num_betas=None``: every column of a file with 300 shape directions, what ``BodyModel('smpl')``
  does with the official file (common.py:223, 381-385).  The reference routes it through ``_fit_shape_general``
  (``gram_supported`` is False: 72 x 6890 x 300 > 2**26, pt/bodyfitter.py:202, 1104-1319);
* ``smpl_w12``  — twelve non-zero skinning weights per vertex (more than the eight pairs the other kernels store; the
  reference blends with the dense (V, J) matrix, pt/bodyfitter.py:1000-1003).

B = 8.  Per model: the default fit (3 iterations, joints given), a joints-omitted fit without the final adjustment, a
fit with vertex + joint weights, the configuration ``BodyConverter.convert`` runs (``enable_kid`` fitter, one iteration,
no ridge, joints omitted, pt/bodyconverter.py:74-88) and forward pins (for ``smpl_b300`` also with 10 of the 300 betas
given, as the README's conversion example does).

Usage:  python tests/golden/make_golden_general.py
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
KINDS = {'smpl_b32': 32, 'smpl_b300': None, 'smpl_w12': 10}


def make_kind(kind, num_betas, root, out):
    arrs = synth.make_model_arrays(kind, seed=0)
    model = ref.BodyModel('smpl', 'neutral', model_root=f'{root}/{kind}', num_betas=num_betas)
    S, J = model.num_betas, model.num_joints
    rs = np.random.RandomState(1357)
    pose = (rs.randn(B, 3 * J) * 0.1).astype(np.float32)
    betas = (rs.randn(B, S) * (0.5 if S <= 32 else 0.15)).astype(np.float32)
    trans = rs.randn(B, 3).astype(np.float32)
    vw = rs.uniform(0.5, 1.5, size=(B, model.num_vertices)).astype(np.float32)
    jw = rs.uniform(0.5, 1.5, size=(B, J)).astype(np.float32)
    pre = kind + '.'
    out.update({pre + 'pose': pose, pre + 'betas': betas, pre + 'trans': trans, pre + 'vertex_weights': vw,
                pre + 'joint_weights': jw, pre + 'model_sha256': np.array(synth.model_sha256(arrs)),
                pre + 'num_betas': np.array(S), pre + 'skin_nnz': np.array(int((arrs['weights'] != 0).sum(1).max()))})
    t = torch.from_numpy
    with torch.no_grad():
        fw = model(t(pose), t(betas), t(trans))
        tv, tj = fw['vertices'], fw['joints']
        out[pre + 'target_vertices'], out[pre + 'target_joints'] = tv.numpy(), tj.numpy()
        out[pre + 'fwd_orientations'] = fw['orientations'].numpy()
        if S > 10:  # fewer betas given than the model has (bodymodel.py:258-264)
            fw10 = model(t(pose), t(betas[:, :10].copy()), t(trans))
            out[pre + 'fwd10_vertices_every_50th'] = fw10['vertices'].numpy()[:, ::50]
            out[pre + 'fwd10_joints'] = fw10['joints'].numpy()
        fitter = ref.BodyFitter(model)
        keys = ['pose_rotvecs', 'shape_betas', 'trans']
        cases = {
            'it3_reg1_j_nw_fa': dict(target_joints=tj, num_iter=3, beta_regularizer=1.0),
            'it2_reg0_nj_nw_nfa': dict(target_joints=None, num_iter=2, beta_regularizer=0.0, final_adjust_rots=False),
            'it2_reg1_j_w_fa': dict(target_joints=tj, vertex_weights=t(vw), joint_weights=t(jw), num_iter=2, beta_regularizer=1.0),
        }
        for name, kw in cases.items():
            r = fitter.fit(tv, requested_keys=keys, **kw)
            for k in keys + ['orientations']:
                out[f'{pre}fit.{name}.{k}'] = r[k].numpy()
        r = ref.BodyFitter(model, enable_kid=True).fit(
            tv, num_iter=1, beta_regularizer=0.0, final_adjust_rots=False, kid_regularizer=1e9,
            requested_keys=keys + ['kid_factor'])
        for k in keys + ['kid_factor']:
            out[f'{pre}fit.conv.{k}'] = r[k].numpy()
    print(kind, 'S =', S, 'gram_supported =', fitter.gram_supported, flush=True)


def main():
    torch.set_num_threads(8)
    root = synth.ensure_model_root(kinds=tuple(KINDS), seed=0)
    out = {}
    for kind, nb in KINDS.items():
        make_kind(kind, nb, root, out)
    path = osp.join(HERE, 'golden_general.npz')
    np.savez_compressed(path, **out)
    print(path, f'{os.path.getsize(path) / 1e6:.2f} MB', len(out), 'arrays')


if __name__ == '__main__':
    main()
