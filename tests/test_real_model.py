"""Opt-in acceptance test on a REAL (licensed) SMPL model file: the reference's own known-answer vector
for BodyModel.forward (reference tests/test_forward.py:7-127; data in
tests/golden/reference_known_answer_smpl.npz).  Skipped unless a real model is supplied through
SMPLFITTER_BODY_MODELS or DATA_ROOT (the synthetic fixture models cannot reproduce these numbers)."""

import os
import os.path as osp

import numpy as np
import pytest

import util

HERE = osp.dirname(osp.abspath(__file__))


def _real_model_root():
    for root in (os.getenv('SMPLFITTER_BODY_MODELS'),
                 osp.join(os.getenv('DATA_ROOT'), 'body_models') if os.getenv('DATA_ROOT') else None):
        if root and osp.exists(osp.join(root, 'smpl', 'basicmodel_neutral_lbs_10_207_0_v1.1.0.pkl')):
            return root
    return None


needs_real = pytest.mark.skipif(_real_model_root() is None, reason='no real SMPL model file supplied')


@pytest.fixture(scope='module')
def known():
    return dict(np.load(osp.join(HERE, 'golden', 'reference_known_answer_smpl.npz')))


def test_fixture_shapes(known):
    assert known['pose_rotvecs'].shape == (1, 72) and known['shape_betas'].shape == (1, 10)
    assert known['vertices_every_300th'].shape[-2:] == (23, 3) and known['joints'].shape == (1, 24, 3)


@needs_real
def test_oracle_forward_real_smpl(known):
    from smplfitter_amd import modelio

    md = modelio.load_model('smpl', 'neutral', model_root=f'{_real_model_root()}/smpl', num_betas=10)
    om, _ = util.make_oracle(md, 'smpl')
    fw = om.forward(known['pose_rotvecs'], known['shape_betas'])
    assert np.allclose(fw['vertices'][:, ::300], known['vertices_every_300th'], atol=2e-6)
    assert np.allclose(fw['joints'], known['joints'], atol=2e-6)


@needs_real
@pytest.mark.gpu
def test_hip_forward_and_fit_real_smpl(known):
    import torch

    from smplfitter_amd.pt import BodyFitter, BodyModel

    dev = torch.device('cuda:0')
    m = BodyModel('smpl', 'neutral', model_root=f'{_real_model_root()}/smpl', num_betas=10, device=dev)
    t = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
    fw = m(t(known['pose_rotvecs']), t(known['shape_betas']))
    assert np.allclose(fw['vertices'].cpu().numpy()[:, ::300], known['vertices_every_300th'], atol=2e-6)
    assert np.allclose(fw['joints'].cpu().numpy(), known['joints'], atol=2e-6)
    # round trip on the real mesh: the reference's own acceptance bar (tests/test_fitter_common.py:31-72)
    r = BodyFitter(m).fit(fw['vertices'], fw['joints'], num_iter=3, beta_regularizer=0.0)
    back = m(r['pose_rotvecs'], r['shape_betas'], r['trans'])
    assert (back['vertices'] - fw['vertices']).norm(dim=-1).mean().item() < 5e-3
