"""Golden vectors for the entry points added after the first fixture set (known-shape fits, scale and
translation alignment), again produced by running the REFERENCE itself in the build container.

Inputs are taken from ``golden_<kind>.npz`` (made by ``make_golden.py``) so that the two fixture files
share targets; only outputs are stored here:

* ``knownshape.<case>.{pose_rotvecs,trans,orientations[,scale_corr]}`` for ``fit_with_known_shape``
  (reference pt/bodyfitter.py:655-838) over cases covering ``num_iter``, joints given / omitted,
  weights, ``scale_fit``, ``kid_factor``, ``initial_pose_rotvecs`` and ``final_adjust_rots``;
* ``scaletrans.<case>.{scale,trans}`` for the module-level ``fit_scale_and_translation``
  (pt/bodyfitter.py:1628-1681);
* ``warm.<case>.*`` for ``fit`` with ``initial_pose_rotvecs / initial_shape_betas / initial_kid_factor``
  (:363-382), including BodyFlipper's configuration (pt/bodyflipper.py:71-81);
* ``scale.<case>.*`` for ``fit(scale_target=True)`` / ``fit(scale_fit=True)`` (:1170-1175, :434-519);
* ``share.<case>.*`` for ``fit(share_beta=True)`` (pt/lstsq.py) on a batch of one shape in several
  poses (targets rebuilt by the tests with the repo's numpy forward; ``target_vertices_sub`` pins them);
* ``flip.*`` for ``BodyFlipper`` (mirror joint permutation, ``flip_vertices`` sampled every 50th vertex,
  ``naive_flip_rotvecs``, ``flip`` results) on the synthetic mirror / transfer files.

Usage:  python tests/golden/make_golden_ext.py
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
from smplfitter.pt.bodyfitter import fit_scale_and_translation  # noqa: E402
from smplfitter_amd import synth  # noqa: E402

sys.path.insert(0, osp.join(HERE, '..'))
from util import (KNOWN_SHAPE_CASES, SCALE_CASES, SCALE_TRANS_CASES, SHARE_CASES, WARM_CASES,  # noqa: E402
                  known_shape_inputs, load_md, make_oracle, scale_inputs, scale_trans_inputs, share_inputs,
                  warm_inputs)

def main():
    torch.set_num_threads(8)
    root = synth.ensure_model_root(kinds=('smpl', 'smplx'), seed=0)
    T = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
    for kind in ('smpl', 'smplx'):
        g = dict(np.load(osp.join(HERE, f'golden_{kind}.npz')))
        model = ref.BodyModel(kind, 'neutral', model_root=f'{root}/{kind}', num_betas=10)
        fitter = ref.BodyFitter(model)
        out = {}
        with torch.no_grad():
            for case in KNOWN_SHAPE_CASES:
                if kind != 'smpl' and case not in ('a', 'b', 'c'):
                    continue
                betas, tv, kw = known_shape_inputs(g, case)
                kwt = {k: (T(v) if isinstance(v, np.ndarray) else v) for k, v in kw.items()}
                if kw['scale_fit']:
                    # the reference's scale branch broadcasts a (B,) scale against (B,3) means
                    # (pt/bodyfitter.py:1675-1676) and only works for B == 1: run instance by instance
                    rows = []
                    for b in range(tv.shape[0]):
                        kb = {k: (v[b:b + 1] if isinstance(v, torch.Tensor) else v) for k, v in kwt.items()}
                        rows.append(fitter.fit_with_known_shape(
                            T(betas)[b:b + 1], T(tv)[b:b + 1], requested_keys=['pose_rotvecs'], **kb))
                    r = {k: torch.cat([x[k].reshape(1, *x[k].shape[1:]) for x in rows]) for k in rows[0]}
                else:
                    r = fitter.fit_with_known_shape(T(betas), T(tv), requested_keys=['pose_rotvecs'], **kwt)
                for k in ('pose_rotvecs', 'trans', 'orientations', 'scale_corr'):
                    if k in r:
                        out[f'knownshape.{case}.{k}'] = r[k].numpy()
            kfitter = ref.BodyFitter(model, enable_kid=True)
            for case in WARM_CASES:
                if kind != 'smpl' and case not in ('a', 'c'):
                    continue
                kid_fit, tv, kw = warm_inputs(g, case)
                kwt = {k: (T(v) if isinstance(v, np.ndarray) else v) for k, v in kw.items()}
                r = (kfitter if kid_fit else fitter).fit(
                    T(tv), requested_keys=['pose_rotvecs', 'shape_betas', 'trans'], **kwt)
                for k in ('pose_rotvecs', 'shape_betas', 'trans', 'kid_factor', 'orientations'):
                    if k in r:
                        out[f'warm.{case}.{k}'] = r[k].numpy()
            for case in SCALE_CASES:  # the scale unknown of the last shape solve
                if kind != 'smpl' and case not in ('a', 'b'):
                    continue
                kid_fit, tv, kw = scale_inputs(g, case)
                kwt = {k: (T(v) if isinstance(v, np.ndarray) else v) for k, v in kw.items()}
                r = (kfitter if kid_fit else fitter).fit(
                    T(tv), requested_keys=['pose_rotvecs', 'shape_betas', 'trans', 'scale_corr'], **kwt)
                for k in ('pose_rotvecs', 'shape_betas', 'trans', 'kid_factor', 'orientations', 'scale_corr'):
                    if k in r:
                        out[f'scale.{case}.{k}'] = r[k].numpy()
            # share_beta: the targets come from the repo's own numpy forward (pinned by golden_<kind>.npz)
            _, md_ = load_md(root, kind, g)
            om_, _ = make_oracle(md_, kind)
            for case in SHARE_CASES:
                if kind != 'smpl' and case != 'a':
                    continue
                kid_fit, tv, kw = share_inputs(g, om_, case)
                out[f'share.{case}.target_vertices_sub'] = tv[:, ::300]
                kwt = {k: (T(v) if isinstance(v, np.ndarray) else v) for k, v in kw.items()}
                r = (kfitter if kid_fit else fitter).fit(
                    T(tv), share_beta=True, requested_keys=['pose_rotvecs', 'shape_betas', 'trans'], **kwt)
                for k in ('pose_rotvecs', 'shape_betas', 'trans', 'kid_factor', 'orientations'):
                    if k in r:
                        out[f'share.{case}.{k}'] = r[k].numpy()
            # BodyFlipper (pt/bodyflipper.py) on the synthetic mirror / transfer files of
            # synth.write_transfer_files
            os.environ['DATA_ROOT'] = synth.write_transfer_files(
                os.getenv('SMPLFIT_SYNTH_DATA', '/tmp/smplfit_synth_data_seed0'))
            flipper = ref.BodyFlipper(model)
            out['flip.mirror_inds_joints'] = flipper.mirror_inds_joints.numpy()
            out['flip.vertices_sub'] = flipper.flip_vertices(T(g['target_vertices'])).numpy()[:, ::50]
            out['flip.naive_rotvecs'] = flipper.naive_flip_rotvecs(T(g['pose'])).numpy()
            for tag, kidf, ni in (('a', None, 1), ('b', None, 3), ('c', g.get('kid'), 2)):
                if kidf is None and tag == 'c':
                    continue
                r = flipper.flip(T(g['pose']), T(g['betas']), T(g['trans']),
                                 kid_factor=None if kidf is None else T(kidf), num_iter=ni)
                for k in ('pose_rotvecs', 'shape_betas', 'trans', 'kid_factor'):
                    if r.get(k) is not None:
                        out[f'flip.{tag}.{k}'] = r[k].numpy()
            if kind == 'smpl':
                tv, tj, rv, rj, vw, jw = scale_trans_inputs(g)
                for case, (uj, uw, sc) in SCALE_TRANS_CASES.items():
                    ss, ts = [], []
                    for b in range(tv.shape[0]):  # B == 1 calls (see above)
                        sl = slice(b, b + 1)
                        s, t = fit_scale_and_translation(
                            T(tv[sl]), T(rv[sl]), T(tj[sl]) if uj else None, T(rj[sl]) if uj else None,
                            T(vw[sl]) if uw else None, T(jw[sl]) if (uw and uj) else None, scale=sc)
                        ts.append(t.numpy())
                        if s is not None:
                            ss.append(s.numpy())
                    out[f'scaletrans.{case}.trans'] = np.concatenate(ts)
                    if ss:
                        out[f'scaletrans.{case}.scale'] = np.concatenate(ss)
        path = osp.join(HERE, f'golden_ext_{kind}.npz')
        np.savez_compressed(path, **out)
        print(path, f'{os.path.getsize(path) / 1e3:.1f} kB', len(out), 'arrays')


if __name__ == '__main__':
    main()
