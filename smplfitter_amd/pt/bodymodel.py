"""``BodyModel`` — same surface as ``smplfitter.pt.BodyModel`` (reference
src/smplfitter/pt/bodymodel.py:12-453), with ``forward`` dispatched to the HIP LBS kernels through the
C-ABI (``smplfit_forward_f32``).  PyTorch tensors are containers only: device memory + stream.
"""

from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from .. import _lib, modelio


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


class BodyModel(nn.Module):
    """Statistical body model of the SMPL family (forward = blend shapes + FK + linear blend
    skinning).  Constructor arguments, buffers and attributes follow the reference
    (pt/bodymodel.py:53-119)."""

    def __init__(
        self,
        model_name: str = 'smpl',
        gender: str = 'neutral',
        model_root: Optional[str] = None,
        num_betas: Optional[int] = None,
        vertex_subset_size: Optional[int] = None,
        vertex_subset=None,
        faces=None,
        joint_regressor_post_lbs=None,
        device=None,
    ):
        super().__init__()
        self.gender = gender
        self.model_name = model_name
        if isinstance(vertex_subset, torch.Tensor):
            vertex_subset = vertex_subset.cpu().numpy()
        data = modelio.load_model(
            model_name, gender, model_root, num_betas, vertex_subset_size, vertex_subset, faces,
            joint_regressor_post_lbs,
        )
        f32 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32)  # noqa: E731
        self.v_template = nn.Buffer(f32(data.v_template))
        self.shapedirs = nn.Buffer(f32(data.shapedirs))
        self.posedirs = nn.Buffer(f32(data.posedirs))
        self.J_regressor_post_lbs = nn.Buffer(f32(data.J_regressor_post_lbs))
        self.J_template = nn.Buffer(f32(data.J_template))
        self.J_shapedirs = nn.Buffer(f32(data.J_shapedirs))
        self.kid_shapedir = nn.Buffer(f32(data.kid_shapedir))
        self.kid_J_shapedir = nn.Buffer(f32(data.kid_J_shapedir))
        self.weights = nn.Buffer(f32(data.weights))
        self.kintree_parents_tensor = nn.Buffer(torch.tensor(data.kintree_parents, dtype=torch.int64))
        self.kintree_parents = data.kintree_parents
        self.faces = data.faces
        self.num_joints = data.num_joints
        self.num_vertices = data.num_vertices
        self.num_betas = self.shapedirs.shape[2]
        self.vertex_subset = data.vertex_subset
        self.joint_names = data.joint_names
        if self.vertex_subset is None:
            self.vertex_subset = np.arange(self.num_vertices)
        self._handles = {}  # device index -> _lib.Handle (model constants uploaded to that GPU)
        from . import ops

        ops.register_model(self)  # id for the torch.library operators (read while compiling)
        if device is not None:
            self.to(device)

    # -- native handle ---------------------------------------------------------------------------
    def _native(self, device: torch.device, kid: bool = False) -> _lib.Handle:
        """The C-ABI handle holding this model's constants on ``device`` (created on first use).
        ``kid=True``: the variant whose last shape unknown is the kid blend shape."""
        if device.type != 'cuda':
            raise RuntimeError(
                'smplfitter_amd runs on MI355X through its HIP kernels only: move the model and the '
                "inputs to a 'cuda' (ROCm) device. There is no CPU path."
            )
        idx = device.index if device.index is not None else torch.cuda.current_device()
        h = self._handles.get((idx, kid))
        if h is None:
            reg = self.J_regressor_post_lbs
            desc, keep = _lib.make_desc(
                self.v_template.cpu().numpy(), self.shapedirs.cpu().numpy(),
                self.posedirs.cpu().numpy(), self.weights.cpu().numpy(),
                self.J_template.cpu().numpy(), self.J_shapedirs.cpu().numpy(), self.kintree_parents,
                reg.cpu().numpy() if reg.shape[1] == self.num_vertices else None,
                is_smpl_family=self.model_name.startswith('smpl'),
                kid_shapedir=self.kid_shapedir.cpu().numpy() if kid else None,
                kid_J_shapedir=self.kid_J_shapedir.cpu().numpy() if kid else None,
            )
            with torch.cuda.device(idx):
                h = _lib.Handle(desc)
            self._handles[(idx, kid)] = h
        return h

    def kernel_path(self, device: Optional[torch.device] = None, enable_kid: bool = False) -> str:
        """Which kernel family fits and forward passes of this model run on (``smplfit_info.vertex_path``):
        ``'batch-major'`` (<= 8 skinning weights per vertex, <= 16 betas: the rates of the benchmark configurations),
        ``'wave-per-instance'`` (small vertex subsets, non-normalised weights: about 0.4 x the rate) or
        ``'general'`` (any other ``num_betas`` — e.g. the default ``None``: every column of the file — or more than eight
        weights per vertex: run-time loops, a correctness path whose per-call workspace holds an S x S fp64 system per
        instance — about 1.7 MB per instance at 300 betas).  No counterpart in the reference."""
        device = self.v_template.device if device is None else torch.device(device)
        return ('wave-per-instance', 'batch-major', 'general')[int(self._native(device, kid=enable_kid).info.vertex_path)]

    @staticmethod
    def _workspace(h: _lib.Handle, batch: int, device) -> torch.Tensor:
        return torch.empty(h.workspace_bytes(batch), dtype=torch.uint8, device=device)

    # -- forward ---------------------------------------------------------------------------------
    def forward(
        self,
        pose_rotvecs: Optional[torch.Tensor] = None,
        shape_betas: Optional[torch.Tensor] = None,
        trans: Optional[torch.Tensor] = None,
        kid_factor: Optional[torch.Tensor] = None,
        rel_rotmats: Optional[torch.Tensor] = None,
        glob_rotmats: Optional[torch.Tensor] = None,
        return_vertices: bool = True,
    ) -> dict[str, torch.Tensor]:
        """Vertices, joints and global orientations for a batch (pt/bodymodel.py:121-307)."""
        if torch.compiler.is_compiling():  # one opaque operator for torch.compile / export
            kid = kid_factor
            if kid is not None and not isinstance(kid, torch.Tensor):
                kid = torch.as_tensor(kid, dtype=torch.float32, device=self.v_template.device)
            j, o, v = torch.ops.smplfitter_amd.forward(
                self._model_id, pose_rotvecs, shape_betas, trans, kid, rel_rotmats, glob_rotmats,
                return_vertices)
            res = dict(joints=j, orientations=o)
            if return_vertices:
                res['vertices'] = v
            return res
        return self._forward_direct(pose_rotvecs, shape_betas, trans, kid_factor, rel_rotmats,
                                    glob_rotmats, return_vertices)

    def _forward_direct(self, pose_rotvecs=None, shape_betas=None, trans=None, kid_factor=None,
                        rel_rotmats=None, glob_rotmats=None, return_vertices: bool = True):
        """The C-ABI call behind ``forward`` (and behind the ``smplfitter_amd::forward`` operator)."""
        n_rot = sum(x is not None for x in (pose_rotvecs, rel_rotmats, glob_rotmats))
        if n_rot > 1:
            raise ValueError(
                'Only one rotation input may be provided '
                '(pose_rotvecs, rel_rotmats, or glob_rotmats).'
            )
        for name, arg, min_ndim in [
            ('pose_rotvecs', pose_rotvecs, 2), ('shape_betas', shape_betas, 2), ('trans', trans, 2),
            ('kid_factor', kid_factor, 1), ('rel_rotmats', rel_rotmats, 4),
            ('glob_rotmats', glob_rotmats, 4),
        ]:
            if arg is not None:
                if isinstance(arg, np.ndarray):
                    raise TypeError(
                        f"Expected torch.Tensor for '{name}', got numpy.ndarray. "
                        f'Convert with torch.from_numpy() or torch.as_tensor().'
                    )
                if arg.ndim < min_ndim:
                    raise ValueError(
                        f"Expected batched input for '{name}' with at least "
                        f'{min_ndim} dimensions, but got shape {tuple(arg.shape)}. '
                        f'For single (unbatched) inputs, use model.single() instead.'
                    )
        device = self.v_template.device
        J, V = self.num_joints, self.num_vertices
        batch = 0
        for arg in (pose_rotvecs, shape_betas, trans, rel_rotmats, glob_rotmats):
            if arg is not None:
                batch = arg.shape[0]
                break
        if batch == 0:
            res = dict(
                joints=torch.empty((0, J, 3), device=device),
                orientations=torch.empty((0, J, 3, 3), device=device),
            )
            if return_vertices:
                res['vertices'] = torch.empty((0, V, 3), device=device)
            return res
        prep = lambda t: None if t is None else t.to(device=device, dtype=torch.float32).contiguous()  # noqa: E731
        # rel_rotmats: the kinematic chain of pt/bodymodel.py:230-234 runs inside the joint kernel
        rel = prep(rel_rotmats.reshape(batch, J, 3, 3)) if rel_rotmats is not None else None
        pose = prep(pose_rotvecs.reshape(batch, J * 3)) if pose_rotvecs is not None else None
        glob = prep(glob_rotmats)
        betas = prep(shape_betas)
        nb = 0
        if betas is not None:
            nb = min(betas.shape[1], self.num_betas)
            betas = betas[:, :nb].contiguous()
            if nb == 0:
                betas = None
        tr = prep(trans)
        if tr is not None and tr.shape[0] != batch:
            tr = tr.expand(batch, 3).contiguous()
        kid = None
        if kid_factor is not None:
            kid = torch.as_tensor(kid_factor, dtype=torch.float32, device=device).reshape(-1)
            kid = kid.expand(batch).contiguous() if kid.numel() == 1 else kid.contiguous()
        h = self._native(device, kid=kid is not None)
        ws = self._workspace(h, batch, device)
        joints = torch.empty((batch, J, 3), dtype=torch.float32, device=device)
        orient = torch.empty((batch, J, 3, 3), dtype=torch.float32, device=device)
        verts = torch.empty((batch, V, 3), dtype=torch.float32, device=device) if return_vertices else None
        with torch.cuda.device(device):
            stream = torch.cuda.current_stream(device).cuda_stream
            p = lambda t: None if t is None else t.data_ptr()  # noqa: E731
            args = _lib.ForwardArgs(
                pose_rotvecs=p(pose), glob_rotmats=p(glob), rel_rotmats=p(rel), shape_betas=p(betas),
                num_betas_given=nb, trans=p(tr), kid_factor=p(kid), batch=batch, vertices=p(verts),
                joints=p(joints), orientations=p(orient), workspace=ws.data_ptr(), workspace_bytes=ws.numel(),
                hip_stream=stream)
            _lib.check(_lib.load().smplfit_forward_ex_f32(h.ptr, C.byref(args)))
        res = dict(joints=joints, orientations=orient)
        if return_vertices:
            res['vertices'] = verts
        return res

    def single(self, pose_rotvecs=None, shape_betas=None, trans=None, kid_factor=None,
               rel_rotmats=None, glob_rotmats=None, return_vertices: bool = True):
        """Unbatched ``forward`` (pt/bodymodel.py:310-380)."""
        u = lambda t: t.unsqueeze(0) if t is not None else None  # noqa: E731
        if all(x is None for x in (pose_rotvecs, shape_betas, trans, rel_rotmats, glob_rotmats)):
            shape_betas = torch.zeros((0,), dtype=torch.float32, device=self.v_template.device)
            pose_rotvecs = torch.zeros((self.num_joints * 3,), dtype=torch.float32,
                                       device=self.v_template.device)
        res = self.forward(u(pose_rotvecs), u(shape_betas), u(trans), kid_factor, u(rel_rotmats),
                           u(glob_rotmats), return_vertices)
        return {k: v.squeeze(0) for k, v in res.items()}
