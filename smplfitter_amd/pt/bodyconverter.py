"""``BodyConverter`` — same surface as ``smplfitter.pt.BodyConverter`` (reference
src/smplfitter/pt/bodyconverter.py:15-158): converts parameters between SMPL-family models by evaluating the
input model, transferring the vertices to the output topology with the sparse barycentric matrix, and fitting the
output model with the kid blend shape enabled.

Everything runs in the HIP kernels.  The default ``convert`` branch is ONE C-ABI call (``smplfit_convert_f32``):
the input model's forward, the topology transfer and the fit hand their meshes to each other in the kernels' own
instance-innermost stream layout, so no ``(B, V, 3)`` intermediate is written.  ``convert_vertices`` — the
stand-alone transfer a caller can use on its own — is ``smplfit_transfer_f32`` (``k_transfer_rows``).
"""

from __future__ import annotations

import ctypes as C
import os
import os.path as osp
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from .. import _lib, modelio
from .bodyfitter import BodyFitter
from .bodymodel import BodyModel


def load_vertex_converter_csr(path):
    """The official ``*_deftrafo_setup.pkl`` files hold a (V_out, 2 V_in) scipy matrix whose first
    V_in columns are the barycentric transfer (reference common.py:425-429)."""
    with open(path, 'rb') as f:
        m = modelio.restricted_load(f, encoding='latin1')['mtx'].tocsr().astype(np.float32)
    return m[:, : m.shape[1] // 2]


# official topology-transfer files under $DATA_ROOT/body_models, keyed by (V_in, V_out)
_TRANSFER_FILES = {(6890, 10475): 'smpl2smplx_deftrafo_setup.pkl', (10475, 6890): 'smplx2smpl_deftrafo_setup.pkl'}


class BodyConverter(nn.Module):
    """Converts parameters between SMPL-family models (reference pt/bodyconverter.py:15-47)."""

    def __init__(self, body_model_in: BodyModel, body_model_out: BodyModel):
        super().__init__()
        self.body_model_in, self.body_model_out = body_model_in, body_model_out
        self.fitter = BodyFitter(body_model_out, enable_kid=True)
        fname = _TRANSFER_FILES.get((body_model_in.num_vertices, body_model_out.num_vertices))
        # the reference keeps a torch sparse-CSR buffer; here the matrix lives in the native library (one device copy
        # per GPU, made on first use) and ``vertex_converter_csr`` is its host (scipy) form, None = same topology
        self.vertex_converter_csr = None
        if fname is not None:
            mat = load_vertex_converter_csr(osp.join(os.getenv('DATA_ROOT', '.'), 'body_models', fname))
            mat.sort_indices()
            self.vertex_converter_csr = mat
        self._transfers = {}  # device index -> _lib.Transfer
        self._plans = {}      # device index -> _lib.ConvertPlan, or None where the fused call does not apply

    # the native objects (ctypes handles) are per-process caches: a copy / pickle of the module starts without them
    def __getstate__(self):
        state = dict(self.__dict__)
        state['_transfers'], state['_plans'] = {}, {}
        return state

    def __deepcopy__(self, memo):
        import copy

        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = {} if k in ('_transfers', '_plans') else copy.deepcopy(v, memo)
        return new

    # -- native objects ----------------------------------------------------------------------------
    def _transfer(self, device: torch.device) -> Optional[_lib.Transfer]:
        if self.vertex_converter_csr is None:
            return None
        idx = device.index if device.index is not None else torch.cuda.current_device()
        t = self._transfers.get(idx)
        if t is None:
            m = self.vertex_converter_csr
            with torch.cuda.device(idx):
                t = _lib.Transfer(m.shape[1], m.shape[0], m.indptr, m.indices, m.data)
            self._transfers[idx] = t
        return t

    def _plan(self, device: torch.device) -> Optional[_lib.ConvertPlan]:
        idx = device.index if device.index is not None else torch.cuda.current_device()
        if idx not in self._plans:
            try:
                with torch.cuda.device(idx):
                    self._plans[idx] = _lib.ConvertPlan(
                        self.body_model_in._native(device), self.body_model_out._native(device, kid=True),
                        self._transfer(device))
            except NotImplementedError:  # a model outside the batch-major kernels: forward + transfer + fit
                self._plans[idx] = None
        return self._plans[idx]

    # -- API -----------------------------------------------------------------------------------------
    def convert(
        self,
        pose_rotvecs: torch.Tensor,
        shape_betas: torch.Tensor,
        trans: torch.Tensor,
        kid_factor: Optional[torch.Tensor] = None,
        known_output_pose_rotvecs: Optional[torch.Tensor] = None,
        known_output_shape_betas: Optional[torch.Tensor] = None,
        known_output_kid_factor: Optional[torch.Tensor] = None,
        num_iter: int = 1,
    ) -> dict[str, torch.Tensor]:
        """Same arguments / results as the reference's ``convert`` (pt/bodyconverter.py:49-126)."""
        kid_reg = 1e9 if kid_factor is None else 0.0
        if known_output_shape_betas is None and known_output_pose_rotvecs is None:
            fit = self._convert_fused(pose_rotvecs, shape_betas, trans, num_iter, kid_reg)
            if fit is None:
                fit = self.fitter.fit(
                    target_vertices=self._input_mesh(pose_rotvecs, shape_betas, trans), num_iter=num_iter,
                    beta_regularizer=0.0, final_adjust_rots=False, kid_regularizer=kid_reg,
                    requested_keys=['pose_rotvecs', 'shape_betas'])
            out = dict(pose_rotvecs=fit['pose_rotvecs'], shape_betas=fit['shape_betas'], trans=fit['trans'])
        elif known_output_shape_betas is not None:  # (:89-98)
            fit = self.fitter.fit_with_known_shape(
                shape_betas=known_output_shape_betas, kid_factor=known_output_kid_factor,
                target_vertices=self._input_mesh(pose_rotvecs, shape_betas, trans), num_iter=num_iter,
                final_adjust_rots=False, requested_keys=['pose_rotvecs'])
            return dict(pose_rotvecs=fit['pose_rotvecs'], trans=fit['trans'])
        else:
            fit = self.fitter.fit_with_known_pose(
                pose_rotvecs=known_output_pose_rotvecs,
                target_vertices=self._input_mesh(pose_rotvecs, shape_betas, trans), beta_regularizer=0.0,
                kid_regularizer=kid_reg)
            out = dict(shape_betas=fit['shape_betas'], trans=fit['trans'])
        if kid_factor is not None:
            out['kid_factor'] = fit['kid_factor']
        return out

    def _input_mesh(self, pose_rotvecs, shape_betas, trans):
        # as in the reference (:86), the input mesh is evaluated WITHOUT kid_factor; it only selects the kid
        # ridge weight and whether kid_factor is returned
        return self.convert_vertices(self.body_model_in(pose_rotvecs, shape_betas, trans)['vertices'])

    def _convert_fused(self, pose_rotvecs, shape_betas, trans, num_iter, kid_reg):
        """``smplfit_convert_f32``: forward (input model) + transfer + fit (output model) in one call; None when the
        fused path does not apply (models outside the batch-major kernels, tracing, empty batch)."""
        mi, mo = self.body_model_in, self.body_model_out
        device = mo.v_template.device
        if torch.compiler.is_compiling() or device.type != 'cuda' or pose_rotvecs.shape[0] == 0:
            return None
        if any(t is not None and t.requires_grad for t in (pose_rotvecs, shape_betas, trans)):
            return None  # the fit raises the NotImplementedError of the unfused path
        plan = self._plan(device)
        if plan is None:
            return None
        B, Ji, Jo, S = pose_rotvecs.shape[0], mi.num_joints, mo.num_joints, self.fitter.n_betas
        prep = lambda t: None if t is None else t.to(device=device, dtype=torch.float32).contiguous()  # noqa: E731
        pose = prep(pose_rotvecs.reshape(B, Ji * 3))
        betas = prep(shape_betas)
        nb = 0
        if betas is not None:
            nb = min(betas.shape[1], mi.num_betas)
            betas = betas[:, :nb].contiguous() if nb > 0 else None
        tr = prep(trans)
        if tr is not None and tr.shape[0] != B:
            tr = tr.expand(B, 3).contiguous()
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=device)  # noqa: E731
        out = dict(pose_rotvecs=new(B, 3 * Jo), shape_betas=new(B, S), trans=new(B, 3), kid_factor=new(B))
        ws = torch.empty(plan.workspace_bytes(B), dtype=torch.uint8, device=device)
        p = lambda t: None if t is None else t.data_ptr()  # noqa: E731
        with torch.cuda.device(device):
            args = _lib.ConvertArgs(
                pose_rotvecs=p(pose), shape_betas=p(betas), num_betas_given=nb, trans=p(tr), batch=B,
                num_iter=int(num_iter), beta_regularizer=0.0, beta_regularizer2=0.0, kid_regularizer=float(kid_reg),
                final_adjust_rots=0, out_pose_rotvecs=p(out['pose_rotvecs']), out_shape_betas=p(out['shape_betas']),
                out_trans=p(out['trans']), out_kid_factor=p(out['kid_factor']), workspace=ws.data_ptr(),
                workspace_bytes=ws.numel(), hip_stream=torch.cuda.current_stream(device).cuda_stream)
            try:
                _lib.check(_lib.load().smplfit_convert_f32(plan.ptr, C.byref(args)))
            except NotImplementedError:
                # the plan was made while the batch-major kernels applied; the tuning options have been reloaded since
                # (SMPLFIT_BM=0, SMPLFIT_GEMM=f32): THIS call takes the forward + transfer + fit calls.  The plan is
                # kept — it is valid again as soon as the options are restored
                return None
        return out

    def convert_vertices(self, inp_vertices: torch.Tensor) -> torch.Tensor:
        """Barycentric topology transfer (pt/bodyconverter.py:128-149); identity when the two models share a
        topology.  ``smplfit_transfer_f32``: (B, V_in, 3) -> (B, V_out, 3)."""
        if self.vertex_converter_csr is None:
            return inp_vertices
        vout, vin = self.vertex_converter_csr.shape
        if inp_vertices.ndim != 3 or tuple(inp_vertices.shape[1:]) != (vin, 3):
            raise ValueError(f'inp_vertices must have shape (batch, {vin}, 3), got {tuple(inp_vertices.shape)}')
        if inp_vertices.requires_grad:
            raise NotImplementedError('the HIP transfer kernel is not differentiable; detach the input')
        device = self.body_model_out.v_template.device
        v = inp_vertices.to(device=device, dtype=torch.float32).contiguous()
        out = torch.empty((v.shape[0], vout, 3), dtype=torch.float32, device=device)
        if v.shape[0] > 0:
            t = self._transfer(device)
            with torch.cuda.device(device):
                _lib.check(_lib.load().smplfit_transfer_f32(
                    t.ptr, C.c_void_p(v.data_ptr()), v.shape[0], C.c_void_p(out.data_ptr()),
                    C.c_void_p(torch.cuda.current_stream(device).cuda_stream)))
        return out
