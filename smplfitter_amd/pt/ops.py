"""``torch.library`` registration of the two hot entry points, ``smplfitter_amd::fit`` and
``smplfitter_amd::forward``, so that ``torch.compile`` / ``torch.export`` see ONE opaque operator with
a shape function instead of tracing into ctypes (SURVEY.md §8b caveat: the reference recommends
``torch.jit.script(fitter)``; a ctypes-backed module is not scriptable, a registered operator is
compilable).  ``BodyFitter.fit`` and ``BodyModel.forward`` route through these operators while a
compiler is tracing and call the C-ABI directly otherwise.

The operators take a ``model_id`` (an integer naming a live ``BodyModel``, see ``register_model``)
because operator arguments must be tensors or scalars.  They are not differentiable, as the HIP path.
"""

from __future__ import annotations

import itertools
import weakref
from typing import List, Optional

import torch
from torch.library import custom_op

_models: 'weakref.WeakValueDictionary[int, torch.nn.Module]' = weakref.WeakValueDictionary()
_ids = itertools.count(1)


def register_model(model) -> int:
    """Give ``model`` (a ``BodyModel``) an id usable as the ``model_id`` operator argument."""
    mid = getattr(model, '_model_id', None)
    if mid is None:
        mid = next(_ids)
        model._model_id = mid
    _models[mid] = model
    return mid


def _model(model_id: int):
    m = _models.get(model_id)
    if m is None:
        raise RuntimeError(f'smplfitter_amd: no live BodyModel with id {model_id}')
    return m


def _fitter(model_id: int, enable_kid: bool):
    """The operator's BodyFitter lives ON the model (``model._op_fitters``): no module-level strong reference
    keeps a model and its per-device handles (19-66 MB each) alive after the user dropped it."""
    from .bodyfitter import BodyFitter

    m = _model(model_id)
    cache = m.__dict__.setdefault('_op_fitters', {})
    f = cache.get(enable_kid)
    if f is None:
        f = BodyFitter(m, enable_kid=enable_kid)
        cache[enable_kid] = f
    return f


def _model_device(model_id: int):
    return _model(model_id).v_template.device


@custom_op('smplfitter_amd::fit', mutates_args=())
def fit(
    model_id: int, enable_kid: bool, target_vertices: torch.Tensor, target_joints: Optional[torch.Tensor],
    vertex_weights: Optional[torch.Tensor], joint_weights: Optional[torch.Tensor], num_iter: int,
    beta_regularizer: float, beta_regularizer2: float, kid_regularizer: float, final_adjust_rots: bool,
    initial_pose_rotvecs: Optional[torch.Tensor], initial_shape_betas: Optional[torch.Tensor],
    initial_kid_factor: Optional[torch.Tensor], share_beta: bool, scale_mode: int, scale_regularizer: float,
) -> List[torch.Tensor]:
    """[pose_rotvecs (B,3J), shape_betas (B,S), trans (B,3), kid_factor (B) (zeros without enable_kid),
    orientations (B,J,3,3), relative_orientations (B,J,3,3), scale_corr (B) (empty without a scale option)]."""
    r = _fitter(model_id, enable_kid)._fit_direct(
        target_vertices, target_joints, vertex_weights, joint_weights, num_iter, beta_regularizer,
        beta_regularizer2, kid_regularizer, final_adjust_rots, initial_pose_rotvecs, initial_shape_betas,
        initial_kid_factor, None, share_beta, scale_mode, scale_regularizer)
    kid = r['kid_factor'] if enable_kid else r['trans'].new_zeros((r['trans'].shape[0],))
    scale = r['scale_corr'] if scale_mode else r['trans'].new_zeros((0,))
    return [r['pose_rotvecs'], r['shape_betas'], r['trans'], kid, r['orientations'],
            r['relative_orientations'], scale]


@fit.register_fake
def _(model_id, enable_kid, target_vertices, target_joints, vertex_weights, joint_weights, num_iter,
      beta_regularizer, beta_regularizer2, kid_regularizer, final_adjust_rots, initial_pose_rotvecs,
      initial_shape_betas, initial_kid_factor, share_beta, scale_mode, scale_regularizer):
    m = _model(model_id)
    B, J, S = target_vertices.shape[0], m.num_joints, m.num_betas
    dev = _model_device(model_id)  # the real operator returns tensors on the MODEL's device
    new = lambda *s: target_vertices.new_empty(s, dtype=torch.float32, device=dev)  # noqa: E731
    return [new(B, 3 * J), new(B, S), new(B, 3), new(B), new(B, J, 3, 3), new(B, J, 3, 3),
            new(B if scale_mode else 0)]


@custom_op('smplfitter_amd::forward', mutates_args=())
def forward(
    model_id: int, pose_rotvecs: Optional[torch.Tensor], shape_betas: Optional[torch.Tensor],
    trans: Optional[torch.Tensor], kid_factor: Optional[torch.Tensor], rel_rotmats: Optional[torch.Tensor],
    glob_rotmats: Optional[torch.Tensor], return_vertices: bool,
) -> List[torch.Tensor]:
    """[joints (B,J,3), orientations (B,J,3,3), vertices (B,V,3) or an empty tensor]."""
    r = _model(model_id)._forward_direct(pose_rotvecs, shape_betas, trans, kid_factor, rel_rotmats,
                                         glob_rotmats, return_vertices)
    v = r['vertices'] if return_vertices else r['joints'].new_empty((0,))
    return [r['joints'], r['orientations'], v]


@forward.register_fake
def _(model_id, pose_rotvecs, shape_betas, trans, kid_factor, rel_rotmats, glob_rotmats, return_vertices):
    m = _model(model_id)
    first = next(a for a in (pose_rotvecs, shape_betas, trans, rel_rotmats, glob_rotmats) if a is not None)
    B, J, V = first.shape[0], m.num_joints, m.num_vertices
    dev = _model_device(model_id)
    new = lambda *s: first.new_empty(s, dtype=torch.float32, device=dev)  # noqa: E731
    return [new(B, J, 3), new(B, J, 3, 3), new(B, V, 3) if return_vertices else new(0)]
