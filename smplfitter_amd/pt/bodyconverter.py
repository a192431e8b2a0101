"""``BodyConverter`` — same surface as ``smplfitter.pt.BodyConverter`` (reference
src/smplfitter/pt/bodyconverter.py:15-158): converts parameters between SMPL-family models by
evaluating the input model (HIP forward kernels), transferring the vertices to the output topology
with a sparse barycentric matrix, and fitting the output model with the kid blend shape enabled
(HIP fit kernels)."""

from __future__ import annotations

import os
import os.path as osp
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from .. import modelio
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
        self.vertex_converter_csr: Optional[torch.Tensor]
        if fname is None:
            self.vertex_converter_csr = None  # same topology: identity
        else:
            mat = load_vertex_converter_csr(osp.join(os.getenv('DATA_ROOT', '.'), 'body_models', fname))
            csr = torch.sparse_csr_tensor(*(torch.from_numpy(a) for a in (mat.indptr, mat.indices, mat.data)),
                                          mat.shape)
            self.vertex_converter_csr = nn.Buffer(csr.to(body_model_out.v_template.device))

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
        # as in the reference (:86), the input mesh is evaluated WITHOUT kid_factor; it only selects the
        # kid ridge weight and whether kid_factor is returned
        inp_vertices = self.body_model_in(pose_rotvecs, shape_betas, trans)['vertices']
        verts = self.convert_vertices(inp_vertices)
        kid_reg = 1e9 if kid_factor is None else 0.0
        if known_output_shape_betas is not None:  # (:89-98)
            fit = self.fitter.fit_with_known_shape(
                shape_betas=known_output_shape_betas, kid_factor=known_output_kid_factor,
                target_vertices=verts, num_iter=num_iter, final_adjust_rots=False,
                requested_keys=['pose_rotvecs'])
            return dict(pose_rotvecs=fit['pose_rotvecs'], trans=fit['trans'])
        if known_output_pose_rotvecs is not None:
            fit = self.fitter.fit_with_known_pose(
                pose_rotvecs=known_output_pose_rotvecs, target_vertices=verts, beta_regularizer=0.0,
                kid_regularizer=kid_reg)
            out = dict(shape_betas=fit['shape_betas'], trans=fit['trans'])
        else:
            fit = self.fitter.fit(
                target_vertices=verts, num_iter=num_iter, beta_regularizer=0.0,
                final_adjust_rots=False, kid_regularizer=kid_reg,
                requested_keys=['pose_rotvecs', 'shape_betas'])
            out = dict(pose_rotvecs=fit['pose_rotvecs'], shape_betas=fit['shape_betas'], trans=fit['trans'])
        if kid_factor is not None:
            out['kid_factor'] = fit['kid_factor']
        return out

    def convert_vertices(self, inp_vertices: torch.Tensor) -> torch.Tensor:
        """Barycentric topology transfer (pt/bodyconverter.py:128-149); identity when the two models
        share a topology.  One sparse (V_out x V_in) @ (V_in x 3B) product (host-side glue)."""
        if self.vertex_converter_csr is None:
            return inp_vertices
        vin, vout = self.body_model_in.num_vertices, self.body_model_out.num_vertices
        v = inp_vertices.permute(1, 0, 2).reshape(vin, -1)
        r = torch.sparse.mm(self.vertex_converter_csr, v)
        return r.reshape(vout, -1, 3).permute(1, 0, 2).contiguous()
