"""Drop-in counterpart of ``smplfitter.pt`` for the ``BodyFitter.fit`` hot path on MI355X.

``BodyModel`` / ``BodyFitter`` keep the reference's names, constructor arguments, ``forward`` / ``fit``
signatures and result keys (reference src/smplfitter/pt/__init__.py).  ``get_cached_fit_fn`` mirrors
pt/__init__.py:58-132 without ``torch.jit.script`` (a ctypes-backed module is not scriptable).
"""

from __future__ import annotations

import functools
from typing import Optional

import torch

from .bodymodel import BodyModel
from .bodyfitter import BodyFitter
from .bodyconverter import BodyConverter
from .bodyflipper import BodyFlipper
from . import ops  # noqa: F401  (registers torch.ops.smplfitter_amd.fit / .forward)

__all__ = ['BodyModel', 'BodyFitter', 'BodyConverter', 'BodyFlipper', 'get_cached_body_model',
           'get_cached_fit_fn']


@functools.lru_cache()
def get_cached_body_model(model_name='smpl', gender='neutral', model_root=None):
    return BodyModel(model_root=model_root, gender=gender, model_name=model_name)


@functools.lru_cache()
def get_cached_fit_fn(
    body_model_name='smpl',
    gender='neutral',
    num_betas=10,
    enable_kid=False,
    requested_keys=('pose_rotvecs', 'shape_betas', 'trans'),
    beta_regularizer=1.0,
    beta_regularizer2=0.0,
    num_iter=3,
    vertex_subset=None,
    joint_regressor_post_lbs=None,
    share_beta=False,
    final_adjust_rots=True,
    scale_target=False,
    scale_fit=False,
    scale_regularizer=0.0,
    kid_regularizer=None,
    device='cuda',
    model_root=None,
):
    body_model = BodyModel(
        gender=gender, model_name=body_model_name, num_betas=num_betas, model_root=model_root,
        vertex_subset=vertex_subset, joint_regressor_post_lbs=joint_regressor_post_lbs, device=device,
    )
    fitter = BodyFitter(body_model, enable_kid=enable_kid)

    def wrapped(verts, joints=None, vertex_weights=None, joint_weights=None):
        V, J = body_model.num_vertices, body_model.num_joints
        r = lambda t, *s: None if t is None else t.reshape(-1, *s)  # noqa: E731
        res = fitter.fit(
            r(verts, V, 3), target_joints=r(joints, J, 3), vertex_weights=r(vertex_weights, V),
            joint_weights=r(joint_weights, J), num_iter=num_iter, beta_regularizer=beta_regularizer,
            beta_regularizer2=beta_regularizer2, scale_regularizer=scale_regularizer,
            kid_regularizer=kid_regularizer, share_beta=share_beta,
            final_adjust_rots=final_adjust_rots, scale_target=scale_target, scale_fit=scale_fit,
            requested_keys=list(requested_keys),
        )
        return {k: v.view(*verts.shape[:-2], *v.shape[1:]) for k, v in res.items()}

    return wrapped
