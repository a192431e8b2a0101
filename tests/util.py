"""Shared helpers for the parity tests (oracle construction, metrics)."""

import os.path as osp
import sys

import numpy as np

ROOT = osp.dirname(osp.dirname(osp.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import smplfit_oracle as O  # noqa: E402
from smplfitter_amd import modelio  # noqa: E402


def load_md(root, name, g=None):
    """ModelData for golden set ``name`` (smpl, smplx, smpl1024)."""
    kind = 'smplx' if name.startswith('smplx') else 'smpl'
    kw = {}
    if g is not None and 'vertex_subset' in g:
        kw['vertex_subset'] = g['vertex_subset']
    return kind, modelio.load_model(kind, 'neutral', model_root=f'{root}/{kind}', num_betas=10, **kw)


def make_oracle(md, kind, dtype=np.float32):
    om = O.OracleModel(md, dtype, kind)
    return om, O.OracleFitter(om)


def cfg_from_name(c):
    it, reg, j, w, fa = c.split('_')
    return dict(num_iter=int(it[2:]), beta_regularizer=float(reg[3:]), joints=(j == 'j'),
                weights=(w == 'w'), final_adjust_rots=(fa == 'fa'))


def fit_configs(g):
    return sorted({k.split('.')[1] for k in g if k.startswith('fit.')})


def vertex_l2(om64, a, b):
    """max over batch and vertices of ||forward(a) - forward(b)||_2, evaluated in fp64."""
    va = om64.forward(a['pose_rotvecs'], a['shape_betas'], a['trans'])['vertices']
    vb = om64.forward(b['pose_rotvecs'], b['shape_betas'], b['trans'])['vertices']
    return float(np.linalg.norm(va - vb, axis=-1).max())
