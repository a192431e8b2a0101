"""``BodyFitter`` — same surface as ``smplfitter.pt.BodyFitter`` (reference
src/smplfitter/pt/bodyfitter.py:15-549); ``fit`` in its default configuration runs entirely in the HIP
kernels behind ``smplfit_fit_f32``.  Options the kernels do not implement raise
``NotImplementedError`` — they never silently compute something else.
"""

from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from .. import _lib
from .bodymodel import BodyModel, _ptr


def _share_callback(ws: torch.Tensor, group):
    """``smplfit_share_allreduce_fn`` for a ``share_beta`` batch sharded over ``group`` (a null callback
    without one) and the list an exception raised inside it is parked in.  ``sums`` lies inside the
    workspace tensor ``ws``; the collective is ordered after the kernels already enqueued on the current
    stream (nccl = RCCL) or synchronous (gloo)."""
    failure: list = []
    if group is None:
        return _lib.ShareAllreduceFn(), failure

    def allreduce_sums(_user, sums_ptr, count, _stream):
        try:
            import torch.distributed as dist
            off = sums_ptr - ws.data_ptr()
            sums = ws.view(torch.uint8).reshape(-1)[off:off + 8 * count].view(torch.float64)
            if dist.get_backend(group) == 'gloo':  # host collective: stage explicitly
                staged = sums.cpu()
                dist.all_reduce(staged, op=dist.ReduceOp.SUM, group=group)
                sums.copy_(staged)
            else:
                dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
            return 0
        except BaseException as e:  # never unwind through the C frames
            failure.append(e)
            return 1

    return _lib.ShareAllreduceFn(allreduce_sums), failure


class BodyFitter(nn.Module):
    """Fits SMPL-family parameters (pose, shape, translation) to target vertices (and joints).

    Parameters:
        body_model: the :class:`BodyModel` to fit.
        enable_kid: adds the kid blend shape (AGORA) as one more shape unknown
            (reference pt/bodyfitter.py:52-58); results then carry ``kid_factor``.
    """

    def __init__(self, body_model: BodyModel, enable_kid: bool = False):
        super().__init__()
        self.body_model = body_model
        self.n_betas = body_model.shapedirs.shape[2]
        self.enable_kid = enable_kid
        self._torchfit = None  # tables of the differentiable restatement (pt/_autograd.py), built on first use
        self.is_smpl_family = body_model.model_name.startswith('smpl')

    def fit(
        self,
        target_vertices: torch.Tensor,
        target_joints: Optional[torch.Tensor] = None,
        vertex_weights: Optional[torch.Tensor] = None,
        joint_weights: Optional[torch.Tensor] = None,
        num_iter: int = 1,
        beta_regularizer: float = 1,
        beta_regularizer2: float = 0,
        scale_regularizer: float = 0,
        kid_regularizer: Optional[float] = None,
        share_beta: bool = False,
        final_adjust_rots: bool = True,
        scale_target: bool = False,
        scale_fit: bool = False,
        initial_pose_rotvecs: Optional[torch.Tensor] = None,
        initial_shape_betas: Optional[torch.Tensor] = None,
        initial_kid_factor: Optional[torch.Tensor] = None,
        requested_keys: Optional[list[str]] = None,
        _workspace: Optional[torch.Tensor] = None,
        share_beta_group=None,
    ) -> dict[str, torch.Tensor]:
        """Same arguments and returned keys as the reference's ``fit`` (pt/bodyfitter.py:283-549):
        ``shape_betas, trans, orientations, relative_orientations`` and ``pose_rotvecs`` when
        requested (default).  ``initial_pose_rotvecs / initial_shape_betas / initial_kid_factor``
        warm-start the fit (``smplfit_fit_warm_f32``; reference :363-382).

        ``share_beta_group`` (not in the reference): a ``torch.distributed`` process group over which
        a ``share_beta`` batch is sharded, one block of instances per rank — the summed normal
        equations of every shape solve are all-reduced over it, so all ranks get the shape of the
        WHOLE batch (``smplfitter_amd.dist.fit_sharded`` passes it)."""
        if requested_keys is None:
            requested_keys = ['pose_rotvecs']
        if scale_target and scale_fit:  # same check, same message as pt/bodyfitter.py:858-859
            raise ValueError('Only one of estim_scale_target and estim_scale_fit can be True')
        scale_mode = 1 if scale_target else 2 if scale_fit else 0
        if share_beta_group is not None:
            if not share_beta:
                raise ValueError('share_beta_group needs share_beta=True')
            if torch.compiler.is_compiling():
                raise NotImplementedError('a sharded share_beta fit cannot be traced (collective inside the fit)')
        if initial_kid_factor is not None and not self.enable_kid:
            raise NotImplementedError(
                'initial_kid_factor needs BodyFitter(enable_kid=True) on the HIP path')
        # the reference defaults the kid ridge weight to beta_regularizer (pt/bodyfitter.py:1235-1237)
        kid_reg = float(beta_regularizer if kid_regularizer is None else kid_regularizer)
        if torch.is_grad_enabled() and not torch.compiler.is_compiling() and any(
                t is not None and t.requires_grad for t in (target_vertices, target_joints, vertex_weights, joint_weights)):
            # gradients with respect to the targets / weights (reference: tests/pt/test_fitter_grad.py): the HIP kernels
            # have no backward pass, so THIS call — and only a call whose inputs require gradients — runs the algorithm
            # written with PyTorch operators (pt/_autograd.py) on the model's cuda device
            result = self._fit_differentiable(
                target_vertices, target_joints, vertex_weights, joint_weights, num_iter, beta_regularizer,
                beta_regularizer2, kid_regularizer, final_adjust_rots,
                unsupported=dict(share_beta=share_beta, scale_target=scale_target, scale_fit=scale_fit,
                                 initial_pose_rotvecs=initial_pose_rotvecs, initial_shape_betas=initial_shape_betas,
                                 initial_kid_factor=initial_kid_factor))
        elif torch.compiler.is_compiling():  # one opaque operator for torch.compile / export
            pose, betas, trans, kid, orient, rel, scale = torch.ops.smplfitter_amd.fit(
                self.body_model._model_id, self.enable_kid, target_vertices, target_joints,
                vertex_weights, joint_weights, int(num_iter), float(beta_regularizer),
                float(beta_regularizer2), kid_reg, bool(final_adjust_rots), initial_pose_rotvecs,
                initial_shape_betas, initial_kid_factor, bool(share_beta), scale_mode,
                float(scale_regularizer))
            result = dict(pose_rotvecs=pose, shape_betas=betas, trans=trans, orientations=orient,
                          relative_orientations=rel)
            if self.enable_kid:
                result['kid_factor'] = kid
            if scale_mode:
                result['scale_corr'] = scale
        else:
            result = self._fit_direct(target_vertices, target_joints, vertex_weights, joint_weights,
                                      num_iter, beta_regularizer, beta_regularizer2, kid_reg,
                                      final_adjust_rots, initial_pose_rotvecs, initial_shape_betas,
                                      initial_kid_factor, _workspace, share_beta, scale_mode,
                                      scale_regularizer, share_beta_group)
        # relative_orientations = parent^T @ global of the FINAL rotations (pt/bodyfitter.py:523-533);
        # returned always (the reference returns the pre-refinement ones when neither
        # 'relative_orientations' nor 'pose_rotvecs' is requested)
        if 'pose_rotvecs' not in requested_keys:
            result.pop('pose_rotvecs', None)
        return result

    def _fit_differentiable(self, target_vertices, target_joints, vertex_weights, joint_weights, num_iter,
                            beta_regularizer, beta_regularizer2, kid_regularizer, final_adjust_rots, unsupported):
        from ._autograd import TorchFit

        bm = self.body_model
        device = bm.v_template.device
        if device.type != 'cuda':
            raise RuntimeError("smplfitter_amd runs on MI355X only: move the model and the inputs to a 'cuda' (ROCm) device")
        bad = [k for k, v in unsupported.items() if v is not None and v is not False]
        if bad:
            raise NotImplementedError(f'the differentiable fit does not implement {", ".join(bad)}; detach the inputs to use the HIP path')
        if self._torchfit is None:
            self._torchfit = TorchFit(bm, enable_kid=self.enable_kid)
            # said once per fitter: a caller that feeds network outputs in training mode without detach() lands here
            import warnings

            warnings.warn('smplfitter_amd: inputs of BodyFitter.fit require gradients — this call runs the PyTorch '
                          'restatement (pt/_autograd.py), hundreds of times slower than the HIP kernels; detach() the '
                          'inputs or call under torch.no_grad() unless the gradients are wanted', RuntimeWarning, stacklevel=3)
        mv = lambda t: None if t is None else t.to(device)  # noqa: E731
        return self._torchfit.fit(mv(target_vertices), mv(target_joints), mv(vertex_weights), mv(joint_weights),
                                  num_iter=int(num_iter), beta_regularizer=float(beta_regularizer),
                                  beta_regularizer2=float(beta_regularizer2), kid_regularizer=kid_regularizer,
                                  final_adjust_rots=bool(final_adjust_rots))

    def _fit_direct(self, target_vertices, target_joints, vertex_weights, joint_weights, num_iter,
                    beta_regularizer, beta_regularizer2, kid_reg, final_adjust_rots,
                    initial_pose_rotvecs, initial_shape_betas, initial_kid_factor, _workspace,
                    share_beta=False, scale_mode=0, scale_regularizer=0.0, share_beta_group=None):
        """The C-ABI call behind ``fit`` (and behind the ``smplfitter_amd::fit`` operator): every result
        tensor, ``pose_rotvecs`` included."""
        bm = self.body_model
        device = bm.v_template.device
        for t in (target_vertices, target_joints, vertex_weights, joint_weights):
            if t is not None and t.requires_grad:
                raise NotImplementedError('the HIP fit kernels are not differentiable; detach the inputs')
        J, V, S = bm.num_joints, bm.num_vertices, self.n_betas
        if target_vertices.ndim != 3 or tuple(target_vertices.shape[1:]) != (V, 3):
            raise ValueError(f'target_vertices must have shape (batch, {V}, 3), got {tuple(target_vertices.shape)}')
        B = target_vertices.shape[0]
        prep = lambda t: None if t is None else t.to(device=device, dtype=torch.float32).contiguous()  # noqa: E731
        tv, tj, vw, jw = prep(target_vertices), prep(target_joints), prep(vertex_weights), prep(joint_weights)
        if tj is not None and tuple(tj.shape) != (B, J, 3):
            raise ValueError(f'target_joints must have shape ({B}, {J}, 3), got {tuple(tj.shape)}')
        if vw is not None and tuple(vw.shape) != (B, V):
            raise ValueError(f'vertex_weights must have shape ({B}, {V})')
        if jw is not None and tuple(jw.shape) != (B, J):
            raise ValueError(f'joint_weights must have shape ({B}, {J})')
        init_pose = None if initial_pose_rotvecs is None else prep(initial_pose_rotvecs.reshape(B, J * 3))
        init_betas = None if initial_shape_betas is None else prep(initial_shape_betas)[:, :S].contiguous()
        init_kid = None
        if initial_kid_factor is not None:
            init_kid = torch.as_tensor(initial_kid_factor, dtype=torch.float32, device=device).reshape(-1)
            init_kid = init_kid.expand(B).contiguous() if init_kid.numel() == 1 else init_kid.contiguous()
        pose = torch.empty((B, 3 * J), dtype=torch.float32, device=device)
        betas = torch.empty((B, S), dtype=torch.float32, device=device)
        trans = torch.empty((B, 3), dtype=torch.float32, device=device)
        orient = torch.empty((B, J, 3, 3), dtype=torch.float32, device=device)
        rel = torch.empty((B, J, 3, 3), dtype=torch.float32, device=device)
        kid = torch.empty((B,), dtype=torch.float32, device=device) if self.enable_kid else None
        scale = torch.empty((B,), dtype=torch.float32, device=device) if scale_mode else None
        if B == 0 and share_beta_group is not None:
            raise ValueError('every rank of a sharded share_beta fit needs at least one instance')
        if B > 0:
            h = bm._native(device, kid=self.enable_kid)
            ws = _workspace if _workspace is not None else bm._workspace(h, B, device)
            callback, failure = _share_callback(ws, share_beta_group)
            with torch.cuda.device(device):
                stream = torch.cuda.current_stream(device).cuda_stream
                args = _lib.FitArgs(
                    target_vertices=tv.data_ptr(), target_joints=tj.data_ptr() if tj is not None else None,
                    vertex_weights=vw.data_ptr() if vw is not None else None,
                    joint_weights=jw.data_ptr() if jw is not None else None, batch=B, num_iter=int(num_iter),
                    beta_regularizer=float(beta_regularizer), beta_regularizer2=float(beta_regularizer2),
                    kid_regularizer=kid_reg, final_adjust_rots=int(bool(final_adjust_rots)),
                    initial_pose_rotvecs=init_pose.data_ptr() if init_pose is not None else None,
                    initial_shape_betas=init_betas.data_ptr() if init_betas is not None else None,
                    num_initial_betas=0 if init_betas is None else init_betas.shape[1],
                    initial_kid_factor=init_kid.data_ptr() if init_kid is not None else None,
                    share_beta=int(bool(share_beta)), scale_mode=int(scale_mode),
                    scale_regularizer=float(scale_regularizer), pose_rotvecs=pose.data_ptr(),
                    shape_betas=betas.data_ptr(), trans=trans.data_ptr(),
                    kid_factor=kid.data_ptr() if kid is not None else None, orientations=orient.data_ptr(),
                    relative_orientations=rel.data_ptr(),
                    scale_corr=scale.data_ptr() if scale is not None else None, workspace=ws.data_ptr(),
                    workspace_bytes=ws.numel(), hip_stream=stream, share_allreduce=callback)
                rc = _lib.load().smplfit_fit_ex_f32(h.ptr, C.byref(args))
                if failure:
                    raise failure[0]
                _lib.check(rc)
        result = dict(pose_rotvecs=pose, shape_betas=betas, trans=trans, orientations=orient,
                      relative_orientations=rel)
        if self.enable_kid:
            result['kid_factor'] = kid
        if scale_mode:
            result['scale_corr'] = scale
        return result


    def fit_with_known_pose(
        self,
        pose_rotvecs: torch.Tensor,
        target_vertices: torch.Tensor,
        target_joints: Optional[torch.Tensor] = None,
        vertex_weights: Optional[torch.Tensor] = None,
        joint_weights: Optional[torch.Tensor] = None,
        beta_regularizer: float = 1,
        beta_regularizer2: float = 0,
        scale_regularizer: float = 0,
        kid_regularizer: Optional[float] = None,
        share_beta: bool = False,
        scale_target: bool = False,
        scale_fit: bool = False,
        beta_regularizer_reference: Optional[torch.Tensor] = None,
        kid_regularizer_reference: Optional[torch.Tensor] = None,
        requested_keys: Optional[list[str]] = None,
        share_beta_group=None,
    ) -> dict[str, torch.Tensor]:
        """Shape and translation (and possibly scale) for a known pose (reference
        pt/bodyfitter.py:552-653): global rotations by forward kinematics of ``pose_rotvecs`` (the HIP
        forward kernel), then one shape solve (``smplfit_shape_solve_ex_f32``) with the target mean added
        back.  ``share_beta`` ignores the ridge references, as the reference's all-shared solve does
        (pt/lstsq.py:45-47).  ``share_beta_group``: as in :meth:`fit`."""
        if scale_target and scale_fit:  # same check, same message as pt/bodyfitter.py:858-859
            raise ValueError('Only one of estim_scale_target and estim_scale_fit can be True')
        if share_beta_group is not None and not share_beta:
            raise ValueError('share_beta_group needs share_beta=True')
        if kid_regularizer_reference is not None and not self.enable_kid:
            kid_regularizer_reference = None  # the reference only reads it with enable_kid (:1235-1246)
        if torch.is_grad_enabled() and any(t is not None and isinstance(t, torch.Tensor) and t.requires_grad for t in
                                          (pose_rotvecs, target_vertices, target_joints, vertex_weights, joint_weights)):
            # (as fit_with_known_shape: never a silently non-differentiable result)
            raise NotImplementedError('fit_with_known_pose on the HIP path is not differentiable: detach the inputs '
                                      '(or call it under torch.no_grad())')
        bm = self.body_model
        B = target_vertices.shape[0]
        pose = pose_rotvecs.reshape(B, bm.num_joints * 3)
        G = bm(pose_rotvecs=pose, return_vertices=False)['orientations']
        kid_reg = float(beta_regularizer if kid_regularizer is None else kid_regularizer)
        r = self._shape_solve(G, target_vertices, target_joints, vertex_weights, joint_weights,
                              beta_regularizer, beta_regularizer2, kid_regularizer=kid_reg,
                              add_mean=True, want_mesh=False, share_beta=share_beta,
                              scale_mode=1 if scale_target else 2 if scale_fit else 0,
                              scale_regularizer=scale_regularizer, beta_ref=beta_regularizer_reference,
                              kid_ref=kid_regularizer_reference, share_beta_group=share_beta_group)
        parents = bm.kintree_parents_tensor[1:].to(G.device)
        parent_glob = torch.cat(
            [torch.eye(3, device=G.device).expand(B, 1, 3, 3), G.index_select(1, parents)], dim=1)
        out = dict(shape_betas=r['shape_betas'], trans=r['trans'], orientations=G,
                   relative_orientations=parent_glob.transpose(-1, -2) @ G)
        if self.enable_kid:
            out['kid_factor'] = r['kid_factor']
        if 'scale_corr' in r:
            out['scale_corr'] = r['scale_corr']
        return out

    def fit_with_known_shape(
        self,
        shape_betas: torch.Tensor,
        target_vertices: torch.Tensor,
        target_joints: Optional[torch.Tensor] = None,
        vertex_weights: Optional[torch.Tensor] = None,
        joint_weights: Optional[torch.Tensor] = None,
        kid_factor: Optional[torch.Tensor] = None,
        num_iter: int = 1,
        final_adjust_rots: bool = True,
        initial_pose_rotvecs: Optional[torch.Tensor] = None,
        scale_fit: bool = False,
        requested_keys: Optional[list[str]] = None,
    ) -> dict[str, torch.Tensor]:
        """Pose and translation (and, with ``scale_fit``, a per-instance scale) for known shape
        parameters (reference pt/bodyfitter.py:655-838), one C-ABI call
        (``smplfit_fit_known_shape_f32``).  Result keys as the reference: ``trans``, ``orientations``,
        ``scale_corr`` with ``scale_fit``, ``relative_orientations`` / ``pose_rotvecs`` when requested.

        The reference's ``scale_fit`` branch only runs for a batch of one (a ``(B,)`` scale is multiplied
        into ``(B,3)`` means, :1675-1676); here every instance gets the scale a batch-of-one call gives."""
        if requested_keys is None:
            requested_keys = ['pose_rotvecs']
        bm = self.body_model
        device = bm.v_template.device
        for name, arg in (('target_vertices', target_vertices), ('shape_betas', shape_betas)):
            if isinstance(arg, np.ndarray):
                raise TypeError(f"Expected torch.Tensor for '{name}', got numpy.ndarray.")
        if target_vertices.ndim != 3:
            raise ValueError(f'Expected batched target_vertices (B, V, 3), got {tuple(target_vertices.shape)}')
        if any(t is not None and t.requires_grad for t in
               (shape_betas, target_vertices, target_joints, initial_pose_rotvecs)):
            raise NotImplementedError('the HIP path is not differentiable')
        prep = lambda t: None if t is None else t.to(device=device, dtype=torch.float32).contiguous()  # noqa: E731
        tv, tj, vw, jw = prep(target_vertices), prep(target_joints), prep(vertex_weights), prep(joint_weights)
        B, J = tv.shape[0], bm.num_joints
        betas = prep(shape_betas)[:, :bm.num_betas].contiguous()
        kid = None
        if kid_factor is not None:
            kid = torch.as_tensor(kid_factor, dtype=torch.float32, device=device).reshape(-1)
            kid = kid.expand(B).contiguous() if kid.numel() == 1 else kid.contiguous()
        init = None if initial_pose_rotvecs is None else prep(initial_pose_rotvecs.reshape(B, J * 3))
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=device)  # noqa: E731
        pose, trans = new(B, J * 3), new(B, 3)
        orient, rel = new(B, J, 3, 3), new(B, J, 3, 3)
        scale = new(B) if scale_fit else None
        if B == 0:
            out = dict(trans=trans, orientations=orient)
        else:
            h = bm._native(device, kid=kid is not None)
            ws = bm._workspace(h, B, device)
            with torch.cuda.device(device):
                stream = torch.cuda.current_stream(device).cuda_stream
                _lib.check(_lib.load().smplfit_fit_known_shape_f32(
                    h.ptr, _ptr(betas), betas.shape[1], _ptr(kid), _ptr(init), _ptr(tv), _ptr(tj), _ptr(vw),
                    _ptr(jw), B, int(num_iter), int(bool(final_adjust_rots)), int(bool(scale_fit)),
                    _ptr(pose), _ptr(trans), _ptr(scale), _ptr(orient), _ptr(rel), _ptr(ws), ws.numel(),
                    C.c_void_p(stream)))
            out = dict(trans=trans, orientations=orient)
        if scale_fit:
            out['scale_corr'] = scale
        if 'relative_orientations' in requested_keys or 'pose_rotvecs' in requested_keys:
            out['relative_orientations'] = rel
        if 'pose_rotvecs' in requested_keys:
            out['pose_rotvecs'] = pose
        return out

    # -- stage entry points, used by the parity tests ---------------------------------------------
    def _part_rotations(self, target_vertices, target_joints=None, vertex_weights=None, joint_weights=None):
        """Global part rotations of the first rotation pass (centring included)."""
        bm = self.body_model
        device = bm.v_template.device
        prep = lambda t: None if t is None else t.to(device=device, dtype=torch.float32).contiguous()  # noqa: E731
        tv, tj, vw, jw = prep(target_vertices), prep(target_joints), prep(vertex_weights), prep(joint_weights)
        B = tv.shape[0]
        G = torch.empty((B, bm.num_joints, 3, 3), dtype=torch.float32, device=device)
        h = bm._native(device, kid=self.enable_kid)
        ws = bm._workspace(h, B, device)
        with torch.cuda.device(device):
            stream = torch.cuda.current_stream(device).cuda_stream
            _lib.check(_lib.load().smplfit_part_rotations_f32(
                h.ptr, _ptr(tv), _ptr(tj), _ptr(vw), _ptr(jw), B, _ptr(G), _ptr(ws), ws.numel(),
                C.c_void_p(stream)))
        return G

    def _shape_solve(self, glob_rotmats, target_vertices, target_joints=None, vertex_weights=None,
                     joint_weights=None, beta_regularizer=1.0, beta_regularizer2=0.0,
                     kid_regularizer=None, add_mean=False, want_mesh=True, share_beta=False, scale_mode=0,
                     scale_regularizer=0.0, beta_ref=None, kid_ref=None, share_beta_group=None):
        """One shape solve for given global rotations; targets are centred internally and, unless
        ``add_mean``, the returned trans / vertices / joints live in the centred frame."""
        bm = self.body_model
        device = bm.v_template.device
        prep = lambda t: None if t is None else t.to(device=device, dtype=torch.float32).contiguous()  # noqa: E731
        G, tv, tj = prep(glob_rotmats), prep(target_vertices), prep(target_joints)
        vw, jw = prep(vertex_weights), prep(joint_weights)
        B, J, V, S = tv.shape[0], bm.num_joints, bm.num_vertices, self.n_betas
        bref = None if beta_ref is None else prep(beta_ref)[:, :S].contiguous()
        kref = None
        if kid_ref is not None:
            kref = torch.as_tensor(kid_ref, dtype=torch.float32, device=device).reshape(-1)
            kref = kref.expand(B).contiguous() if kref.numel() == 1 else kref.contiguous()
        betas = torch.empty((B, S), dtype=torch.float32, device=device)
        trans = torch.empty((B, 3), dtype=torch.float32, device=device)
        verts = torch.empty((B, V, 3), dtype=torch.float32, device=device) if want_mesh else None
        joints = torch.empty((B, J, 3), dtype=torch.float32, device=device) if want_mesh else None
        kid = torch.empty((B,), dtype=torch.float32, device=device) if self.enable_kid else None
        scale = torch.empty((B,), dtype=torch.float32, device=device) if scale_mode else None
        kid_reg = float(beta_regularizer if kid_regularizer is None else kid_regularizer)
        h = bm._native(device, kid=self.enable_kid)
        ws = bm._workspace(h, B, device)
        callback, failure = _share_callback(ws, share_beta_group)
        p = lambda t: None if t is None else t.data_ptr()  # noqa: E731
        with torch.cuda.device(device):
            stream = torch.cuda.current_stream(device).cuda_stream
            args = _lib.ShapeSolveArgs(
                glob_rotmats=p(G), target_vertices=p(tv), target_joints=p(tj), vertex_weights=p(vw),
                joint_weights=p(jw), batch=B, beta_regularizer=float(beta_regularizer),
                beta_regularizer2=float(beta_regularizer2), kid_regularizer=kid_reg,
                add_mean=int(bool(add_mean)), beta_regularizer_reference=p(bref),
                num_reference_betas=0 if bref is None else bref.shape[1], kid_regularizer_reference=p(kref),
                share_beta=int(bool(share_beta)), scale_mode=int(scale_mode),
                scale_regularizer=float(scale_regularizer), shape_betas=p(betas), trans=p(trans),
                kid_factor=p(kid), scale_corr=p(scale), vertices_out=p(verts), joints_out=p(joints),
                workspace=ws.data_ptr(), workspace_bytes=ws.numel(), hip_stream=stream,
                share_allreduce=callback)
            rc = _lib.load().smplfit_shape_solve_ex_f32(h.ptr, C.byref(args))
            if failure:
                raise failure[0]
            _lib.check(rc)
        out = dict(shape_betas=betas, trans=trans)
        if want_mesh:
            out.update(vertices=verts, joints=joints)
        if self.enable_kid:
            out['kid_factor'] = kid
        if scale_mode:
            out['scale_corr'] = scale
        return out
