"""Generate ``golden_general_opts.npz`` by running the REFERENCE (build container only): the options of ``fit`` that add
unknowns or couple the batch — ``scale_target`` / ``scale_fit`` (pt/bodyfitter.py:1170-1175, 434-519), ``share_beta``
(pt/lstsq.py) and both together — on two models of the GENERAL path (``smpl_b32``: 32 betas; ``smpl_w12``: twelve
skinning weights per vertex).  Inputs come from ``golden_general.npz`` (``make_golden_general.py``) through the same
helpers as the fixtures of the other paths (``tests/util.py``: SCALE_CASES, SHARE_CASES, SHARE_SCALE_CASES); only
outputs are stored here.

Usage:  python tests/golden/make_golden_general_opts.py
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

sys.path.insert(0, osp.join(HERE, '..'))
import util  # noqa: E402


def main():
    torch.set_num_threads(8)
    gg = dict(np.load(osp.join(HERE, 'golden_general.npz')))
    root = synth.ensure_model_root(kinds=tuple(util.GENERAL_OPT_KINDS), seed=0)
    T = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
    out = {}
    for kind, cases in util.GENERAL_OPT_KINDS.items():
        nb = util.GENERAL_KINDS[kind]
        g = util.general_view(gg, kind)
        model = ref.BodyModel('smpl', 'neutral', model_root=f'{root}/{kind}', num_betas=nb)
        fitter, kfitter = ref.BodyFitter(model), ref.BodyFitter(model, enable_kid=True)
        om = util.general_oracle(root, kind)
        pre = kind + '.'
        with torch.no_grad():
            for case in cases['scale']:
                kid_fit, tv, kw = util.scale_inputs(g, case)
                kwt = {k: (T(v) if isinstance(v, np.ndarray) else v) for k, v in kw.items()}
                r = (kfitter if kid_fit else fitter).fit(
                    T(tv), requested_keys=['pose_rotvecs', 'shape_betas', 'trans', 'scale_corr'], **kwt)
                for k in ('pose_rotvecs', 'shape_betas', 'trans', 'kid_factor', 'orientations', 'scale_corr'):
                    if k in r:
                        out[f'{pre}scale.{case}.{k}'] = r[k].numpy()
            for case in cases['share']:
                kid_fit, tv, kw = util.share_inputs(g, om, case)
                out[f'{pre}share.{case}.target_vertices_sub'] = tv[:, ::300]
                kwt = {k: (T(v) if isinstance(v, np.ndarray) else v) for k, v in kw.items()}
                r = (kfitter if kid_fit else fitter).fit(
                    T(tv), share_beta=True, requested_keys=['pose_rotvecs', 'shape_betas', 'trans'], **kwt)
                for k in ('pose_rotvecs', 'shape_betas', 'trans', 'kid_factor', 'orientations'):
                    if k in r:
                        out[f'{pre}share.{case}.{k}'] = r[k].numpy()
            for case in cases['sharescale']:
                kid_fit, tv, kw = util.share_scale_inputs(g, om, case)
                out[f'{pre}sharescale.{case}.target_vertices_sub'] = tv[:, ::300]
                kwt = {k: (T(v) if isinstance(v, np.ndarray) else v) for k, v in kw.items()}
                r = (kfitter if kid_fit else fitter).fit(
                    T(tv), share_beta=True,
                    requested_keys=['pose_rotvecs', 'shape_betas', 'trans', 'scale_corr'], **kwt)
                for k in ('pose_rotvecs', 'shape_betas', 'trans', 'kid_factor', 'orientations', 'scale_corr'):
                    if k in r:
                        out[f'{pre}sharescale.{case}.{k}'] = r[k].numpy()
        print(kind, 'done', flush=True)
    path = osp.join(HERE, 'golden_general_opts.npz')
    np.savez_compressed(path, **out)
    print(path, f'{os.path.getsize(path) / 1e3:.1f} kB', len(out), 'arrays')


if __name__ == '__main__':
    main()
