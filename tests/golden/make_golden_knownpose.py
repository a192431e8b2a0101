"""Golden vectors for ``fit_with_known_pose`` with the options of the general shape solve (share_beta,
scale_target / scale_fit, ridge references; reference pt/bodyfitter.py:552-653 -> :1104-1319), produced
by running the REFERENCE itself in the build container on the inputs of ``golden_<kind>.npz``.

Stored: ``knownpose.<case>.{shape_betas,trans[,kid_factor][,scale_corr]}`` for ``util.KNOWN_POSE_CASES``,
``sharescale.<case>.*`` for ``fit(share_beta=True, scale_target / scale_fit=True)`` (``util.SHARE_SCALE_CASES``:
all-shared solves, then the partially shared last solve, pt/lstsq.py:50-90) and ``sharewarm.a.*``: ``fit(share_beta=True)`` on the warm-start inputs of ``util.WARM_CASES['a']`` (the
all-shared solve of the reference ignores the ridge reference the warm start hands it, pt/lstsq.py:45-47).

Usage:  python tests/golden/make_golden_knownpose.py
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
from util import (KNOWN_POSE_CASES, SHARE_SCALE_CASES, known_pose_inputs, load_md, make_oracle,  # noqa: E402
                  share_scale_inputs, warm_inputs)


def main():
    torch.set_num_threads(8)
    root = synth.ensure_model_root(kinds=('smpl', 'smplx'), seed=0)
    T = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
    for kind in ('smpl', 'smplx'):
        g = dict(np.load(osp.join(HERE, f'golden_{kind}.npz')))
        model = ref.BodyModel(kind, 'neutral', model_root=f'{root}/{kind}', num_betas=10)
        fitters = {False: ref.BodyFitter(model), True: ref.BodyFitter(model, enable_kid=True)}
        out = {}
        with torch.no_grad():
            for case in KNOWN_POSE_CASES:
                if kind != 'smpl' and case not in ('a', 'b', 'd', 'h'):
                    continue
                kid_fit, pose, tv, kw = known_pose_inputs(g, case)
                kwt = {k: (T(v) if isinstance(v, np.ndarray) else v) for k, v in kw.items()}
                r = fitters[kid_fit].fit_with_known_pose(T(pose), T(tv), **kwt)
                for k in ('shape_betas', 'trans', 'kid_factor', 'scale_corr'):
                    if r.get(k) is not None:
                        out[f'knownpose.{case}.{k}'] = r[k].numpy()
            # share_beta with a scale unknown; targets from the repo's numpy forward (pinned by golden_<kind>.npz)
            om_, _ = make_oracle(load_md(root, kind, g)[1], kind)
            for case in SHARE_SCALE_CASES:
                if kind != 'smpl' and case != 'a':
                    continue
                kid_fit, tv, kw = share_scale_inputs(g, om_, case)
                kwt = {k: (T(v) if isinstance(v, np.ndarray) else v) for k, v in kw.items()}
                r = fitters[kid_fit].fit(T(tv), share_beta=True,
                                         requested_keys=['pose_rotvecs', 'shape_betas', 'trans'], **kwt)
                for k in ('pose_rotvecs', 'shape_betas', 'trans', 'kid_factor', 'scale_corr'):
                    if r.get(k) is not None:
                        out[f'sharescale.{case}.{k}'] = r[k].numpy()
            if kind == 'smpl':
                _, tv, kw = warm_inputs(g, 'a')
                kwt = {k: (T(v) if isinstance(v, np.ndarray) else v) for k, v in kw.items()}
                r = fitters[False].fit(T(tv), share_beta=True,
                                       requested_keys=['pose_rotvecs', 'shape_betas', 'trans'], **kwt)
                for k in ('pose_rotvecs', 'shape_betas', 'trans'):
                    out[f'sharewarm.a.{k}'] = r[k].numpy()
        path = osp.join(HERE, f'golden_kp_{kind}.npz')
        np.savez_compressed(path, **out)
        print(path, f'{os.path.getsize(path) / 1e3:.1f} kB', len(out), 'arrays')


if __name__ == '__main__':
    main()
