"""The differentiable restatement of the fit (smplfitter_amd/pt/_autograd.py: PyTorch operators, used by
``BodyFitter.fit`` only for inputs that require gradients) — pinned against the reference's fixtures like the oracle, and
its gradients checked the way the reference checks its own (tests/pt/test_fitter_grad.py:31-99: finite, non-zero,
directional finite differences within 5 %).  No GPU needed: the module is plain PyTorch and is called directly here; the
product entry (``BodyFitter.fit`` with ``requires_grad`` inputs on a cuda model) is covered by the ``-m gpu`` tests below."""

import numpy as np
import pytest
import torch

import util


def _cpu_model(model_root, name, g=None):
    from smplfitter_amd.pt import BodyModel

    kind = 'smplx' if name.startswith('smplx') else 'smpl'
    return BodyModel(kind, 'neutral', model_root=f'{model_root}/{util.model_dir(name)}', num_betas=10)


@pytest.mark.parametrize('name', ['smpl', 'smplxfat'])
def test_torch_restatement_matches_reference_fixtures(name, model_root, golden):
    from smplfitter_amd.pt._autograd import TorchFit

    torch.set_num_threads(8)
    g = golden(name)
    kind, md = util.load_md(model_root, name, g)
    om64, _ = util.make_oracle(md, kind, np.float64)
    tf = TorchFit(_cpu_model(model_root, name))
    t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
    n = 0
    for c in util.fit_configs(g):
        cfg = util.cfg_from_name(c)
        with torch.no_grad():
            o = tf.fit(t(g['target_vertices']), t(g['target_joints']) if cfg['joints'] else None,
                       t(g['vertex_weights']) if cfg['weights'] else None,
                       t(g['joint_weights']) if (cfg['weights'] and cfg['joints']) else None,
                       num_iter=cfg['num_iter'], beta_regularizer=cfg['beta_regularizer'],
                       final_adjust_rots=cfg['final_adjust_rots'])
        o = {k: v.numpy() for k, v in o.items()}
        ref = {k: g[f'fit.{c}.{k}'] for k in ('pose_rotvecs', 'shape_betas', 'trans')}
        assert util.vertex_l2(om64, o, ref) < 1e-4, c
        assert np.abs(o['trans'] - ref['trans']).max() < 1e-5, c
        n += 1
        if name != 'smpl' and n >= 3:
            break
    assert n >= 3


def _loss(fit):
    return sum(fit[k].pow(2).sum() for k in ('pose_rotvecs', 'shape_betas', 'trans'))


@pytest.mark.parametrize('num_iter', [1, 3])
def test_gradients_finite_and_match_finite_differences(num_iter, model_root, golden):
    from smplfitter_amd.pt._autograd import TorchFit

    torch.set_num_threads(8)
    g = golden('smpl')
    tf = TorchFit(_cpu_model(model_root, 'smpl'))
    tv0 = torch.from_numpy(g['target_vertices'][:1].copy())
    tj0 = torch.from_numpy(g['target_joints'][:1].copy())
    fit = lambda a, b: _loss(tf.fit(a, b, num_iter=num_iter, beta_regularizer=1.0))  # noqa: E731
    tv, tj = tv0.clone().requires_grad_(True), tj0.clone().requires_grad_(True)
    fit(tv, tj).backward()
    for grad in (tv.grad, tj.grad):
        assert grad is not None and torch.isfinite(grad).all() and grad.abs().max().item() > 0
    if num_iter > 1:
        return
    gen = torch.Generator().manual_seed(100)
    dv, dj = torch.randn(tv0.shape, generator=gen), torch.randn(tj0.shape, generator=gen)
    dv, dj = dv / dv.norm(), dj / dj.norm()
    ag = (tv.grad * dv).sum().item() + (tj.grad * dj).sum().item()
    eps = 1e-2
    with torch.no_grad():
        fd = (fit(tv0 + eps * dv, tj0 + eps * dj).item() - fit(tv0 - eps * dv, tj0 - eps * dj).item()) / (2 * eps)
    assert abs(ag - fd) / max(abs(ag), abs(fd), 1e-3) < 5e-2, (ag, fd)


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['smpl', 'smplx'])
def test_fit_with_requires_grad_on_the_gpu(name, model_root, golden):
    """``BodyFitter.fit`` with inputs that require gradients: the differentiable restatement on the model's device — the
    same results as the HIP path (which the no-gradient calls of the finite-difference legs run), finite non-zero
    gradients, and a directional derivative that matches finite differences OF THE HIP PATH."""
    from smplfitter_amd.pt import BodyFitter, BodyModel

    dev = torch.device('cuda:0')
    g = golden(name)
    kind = 'smplx' if name == 'smplx' else 'smpl'
    m = BodyModel(kind, 'neutral', model_root=f'{model_root}/{util.model_dir(name)}', num_betas=10, device=dev)
    f = BodyFitter(m)
    tv0 = torch.from_numpy(g['target_vertices'][:2].copy()).to(dev)
    tj0 = torch.from_numpy(g['target_joints'][:2].copy()).to(dev)
    kw = dict(num_iter=1, beta_regularizer=1.0, requested_keys=['pose_rotvecs', 'shape_betas', 'trans'])
    hip = f.fit(tv0, tj0, **kw)
    tv, tj = tv0.clone().requires_grad_(True), tj0.clone().requires_grad_(True)
    diff = f.fit(tv, tj, **kw)
    assert diff['pose_rotvecs'].requires_grad
    fwa = m(diff['pose_rotvecs'].detach(), diff['shape_betas'].detach(), diff['trans'].detach())['vertices']
    fwb = m(hip['pose_rotvecs'], hip['shape_betas'], hip['trans'])['vertices']
    assert (fwa - fwb).norm(dim=-1).max().item() < 1e-4
    _loss(diff).backward()
    for grad in (tv.grad, tj.grad):
        assert grad is not None and torch.isfinite(grad).all() and grad.abs().max().item() > 0
    gen = torch.Generator().manual_seed(7)
    dv, dj = torch.randn(tv0.shape, generator=gen).to(dev), torch.randn(tj0.shape, generator=gen).to(dev)
    dv, dj = dv / dv.norm(), dj / dj.norm()
    ag = (tv.grad * dv).sum().item() + (tj.grad * dj).sum().item()
    eps = 1e-2
    lp = _loss(f.fit(tv0 + eps * dv, tj0 + eps * dj, **kw)).item()  # (no gradients: the HIP kernels)
    lm = _loss(f.fit(tv0 - eps * dv, tj0 - eps * dj, **kw)).item()
    fd = (lp - lm) / (2 * eps)
    assert abs(ag - fd) / max(abs(ag), abs(fd), 1e-3) < 5e-2, (ag, fd)
    with pytest.raises(NotImplementedError):
        f.fit(tv, tj, share_beta=True)
    # three iterations + the refinement: finite gradients
    tv2 = tv0.clone().requires_grad_(True)
    _loss(f.fit(tv2, tj0, num_iter=3, beta_regularizer=1.0)).backward()
    assert torch.isfinite(tv2.grad).all() and tv2.grad.abs().max().item() > 0
