"""``BodyFlipper`` — same surface as ``smplfitter.pt.BodyFlipper`` (reference
src/smplfitter/pt/bodyflipper.py:18-169): mirrors body-model parameters along the x axis by
evaluating the model (HIP forward kernels), mirroring + re-indexing the vertices with a sparse
matrix, and fitting the mirrored mesh with a warm start from the naively mirrored parameters
(HIP fit kernels, ``smplfit_fit_warm_f32``)."""

from __future__ import annotations

import os
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from .bodyconverter import load_vertex_converter_csr
from .bodyfitter import BodyFitter
from .bodymodel import BodyModel


def get_mirror_mapping(points: torch.Tensor) -> torch.Tensor:
    """Permutation pairing every point with its mirror image: the optimal assignment between the
    points and their x-flipped copies (reference :129-133, same scipy calls)."""
    import scipy.optimize
    import scipy.spatial.distance

    p = points.detach().cpu().numpy()
    dist = scipy.spatial.distance.cdist(p, p * [-1, 1, 1])
    v_inds, mirror_inds = scipy.optimize.linear_sum_assignment(dist)
    return torch.tensor(mirror_inds[np.argsort(v_inds)], dtype=torch.int, device=points.device)


def load_mirror_csr(path):
    """``smplx_flip_correspondences.npz``: per vertex 3 vertex ids (``closest_faces``) and barycentric
    weights (``bc``) of its mirror image (reference :157-169)."""
    import scipy.sparse

    m = np.load(path)
    faces, bc = m['closest_faces'], m['bc']
    n_verts, n_rows = bc.shape[0], faces.shape[0]
    row = np.repeat(np.arange(n_rows), 3)
    coo = scipy.sparse.coo_matrix((bc.flatten(), (row, faces.flatten())), shape=(n_rows, n_verts))
    return coo.tocsr().astype(np.float32)


def get_mirror_csr(num_verts: int) -> torch.Tensor:
    """Mirror matrix of the model's topology (reference :136-154); files under
    ``$DATA_ROOT/body_models`` as in the reference."""
    data_root = os.getenv('DATA_ROOT', '.')
    smplx2mirror = load_mirror_csr(f'{data_root}/body_models/smplx/smplx_flip_correspondences.npz')
    if num_verts == 6890:
        smpl2smplx = load_vertex_converter_csr(f'{data_root}/body_models/smpl2smplx_deftrafo_setup.pkl')
        smplx2smpl = load_vertex_converter_csr(f'{data_root}/body_models/smplx2smpl_deftrafo_setup.pkl')
        m = (smplx2smpl @ smplx2mirror @ smpl2smplx).tocsr().astype(np.float32)
    elif num_verts == 10475:
        m = smplx2mirror
    else:
        raise ValueError(f'Unsupported number of vertices: {num_verts}')
    m.sort_indices()
    return torch.sparse_csr_tensor(
        torch.from_numpy(m.indptr), torch.from_numpy(m.indices), torch.from_numpy(m.data), m.shape)


class BodyFlipper(nn.Module):
    """Horizontally (x axis) flips SMPL-like body model parameters, to mirror the body."""

    def __init__(self, body_model: BodyModel):
        super().__init__()
        self.body_model = body_model
        self.fitter = BodyFitter(self.body_model, enable_kid=True)
        device = body_model.v_template.device
        res = self.body_model.single()
        self.mirror_csr = nn.Buffer(get_mirror_csr(body_model.num_vertices).to(device))
        self.mirror_inds_joints = nn.Buffer(get_mirror_mapping(res['joints']))
        self._rest_vertices = res['vertices']
        self._mirror_inds: Optional[torch.Tensor] = None

    @property
    def mirror_inds(self) -> torch.Tensor:
        """Vertex mirror permutation.  The reference computes it in ``__init__`` (:32) but ``flip``
        never reads it; the V x V assignment takes tens of seconds, so it is computed on first use."""
        if self._mirror_inds is None:
            self._mirror_inds = get_mirror_mapping(self._rest_vertices)
        return self._mirror_inds

    def flip(
        self,
        pose_rotvecs: torch.Tensor,
        shape_betas: torch.Tensor,
        trans: torch.Tensor,
        kid_factor: Optional[torch.Tensor] = None,
        num_iter: int = 1,
    ) -> dict[str, torch.Tensor]:
        """Parameters of the horizontally flipped body (reference :34-87): the mirrored mesh is fitted
        with ``beta_regularizer = beta_regularizer2 = 1e-2``, the kid unknown pinned unless a
        ``kid_factor`` came in, warm-started from the naively mirrored pose and the input shape."""
        inp = self.body_model(pose_rotvecs, shape_betas, trans, kid_factor=kid_factor)
        flipped_vertices = self.flip_vertices(inp['vertices'])
        fit = self.fitter.fit(
            target_vertices=flipped_vertices,
            num_iter=num_iter,
            beta_regularizer=1e-2,
            beta_regularizer2=1e-2,
            final_adjust_rots=True,
            kid_regularizer=1e9 if kid_factor is None else 0.0,
            initial_pose_rotvecs=self.naive_flip_rotvecs(pose_rotvecs),
            initial_shape_betas=shape_betas,
            requested_keys=['pose_rotvecs', 'shape_betas'],
        )
        return dict(
            pose_rotvecs=fit['pose_rotvecs'],
            shape_betas=fit['shape_betas'],
            trans=fit['trans'],
            kid_factor=fit.get('kid_factor'),
        )

    def flip_vertices(self, inp_vertices: torch.Tensor) -> torch.Tensor:
        """Mirrored, re-indexed vertices (reference :89-107): one sparse (V x V) @ (V x 3B) product
        (host-side glue) and a sign flip of x."""
        V = self.body_model.num_vertices
        hflip = torch.tensor([-1, 1, 1], dtype=inp_vertices.dtype, device=inp_vertices.device)
        v = inp_vertices.permute(1, 0, 2).reshape(V, -1)
        r = torch.sparse.mm(self.mirror_csr, v)
        return (r.reshape(V, -1, 3).permute(1, 0, 2) * hflip).contiguous()

    def naive_flip_rotvecs(self, pose_rotvecs: torch.Tensor) -> torch.Tensor:
        """Rotation vectors mirrored along x with left / right parts exchanged (reference :109-126)."""
        J = self.body_model.num_joints
        hflip = torch.tensor([1, -1, -1], dtype=pose_rotvecs.dtype, device=pose_rotvecs.device)
        reshaped = pose_rotvecs.reshape(-1, J, 3)
        flipped = reshaped[:, self.mirror_inds_joints.long()] * hflip
        return flipped.reshape(-1, J * 3)
