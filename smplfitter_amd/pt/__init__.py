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
from . import ops  # noqa: F401  (registers torch.ops.smplfitter_amd.fit / .forward)

__all__ = ['BodyModel', 'BodyFitter', 'BodyConverter', 'get_cached_body_model', 'get_cached_fit_fn']


@functools.lru_cache()
def get_cached_body_model(model_name='smpl', gender='neutral', model_root=None):
    """One shared ``BodyModel`` per (model, gender, root) — do not modify it in place."""
    return BodyModel(model_name, gender, model_root)


@functools.lru_cache()
def get_cached_fit_fn(body_model_name='smpl', gender='neutral', num_betas=10, enable_kid=False,
                      requested_keys=('pose_rotvecs', 'shape_betas', 'trans'), beta_regularizer=1.0,
                      beta_regularizer2=0.0, num_iter=3, vertex_subset=None, joint_regressor_post_lbs=None,
                      share_beta=False, final_adjust_rots=True, scale_target=False, scale_fit=False,
                      scale_regularizer=0.0, kid_regularizer=None, device='cuda', model_root=None):
    """A cached fitting closure with the reference's keyword list (pt/__init__.py:58-132): builds the
    model + fitter once, accepts inputs with any leading batch shape, returns results in that shape.
    (No ``torch.jit.script``: a ctypes-backed module is not scriptable.)"""
    options = dict(num_iter=num_iter, beta_regularizer=beta_regularizer, beta_regularizer2=beta_regularizer2,
                   scale_regularizer=scale_regularizer, kid_regularizer=kid_regularizer, share_beta=share_beta,
                   final_adjust_rots=final_adjust_rots, scale_target=scale_target, scale_fit=scale_fit)
    model = BodyModel(body_model_name, gender, model_root, num_betas, vertex_subset=vertex_subset,
                      joint_regressor_post_lbs=joint_regressor_post_lbs, device=device)
    fitter = BodyFitter(model, enable_kid=enable_kid)
    keys = list(requested_keys)

    def fit_fn(verts, joints=None, vertex_weights=None, joint_weights=None):
        lead = verts.shape[:-2]
        flat = lambda x, *tail: None if x is None else x.reshape(-1, *tail)  # noqa: E731
        out = fitter.fit(flat(verts, model.num_vertices, 3), flat(joints, model.num_joints, 3),
                         flat(vertex_weights, model.num_vertices), flat(joint_weights, model.num_joints),
                         requested_keys=keys, **options)
        return {name: value.reshape(*lead, *value.shape[1:]) for name, value in out.items()}

    return fit_fn
