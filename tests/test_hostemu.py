"""The kernels' arithmetic, compiled for the host (tests/hostemu): table builder, stage logic and the
kernel orchestration checked against the golden vectors and the oracle — no GPU involved.
This is a test harness of shared headers, not a product path."""

import numpy as np
import pytest

import hostemu_util as H
import util


def test_primitives(golden):
    lib = H.load()
    g = golden('primitives')
    n = int(g['proj_n_random'])
    A = np.ascontiguousarray(g['proj_in'])
    R = np.zeros_like(A)
    lib.hostemu_proj_so3(H._p(A), H._p(R), len(A))
    assert np.abs(R[:n] - g['proj_out'][:n]).max() < 2e-5
    det = np.linalg.det(R.astype(np.float64))
    assert np.abs(det - 1).max() < 1e-5
    assert np.abs(R @ np.swapaxes(R, -1, -2) - np.eye(3)).max() < 1e-5
    for i in (n + 2, n + 3, n + 5, n + 6, n + 7, n + 8):
        assert np.abs(R[i] - g['proj_out'][i]).max() < 1e-4, i
    assert np.abs(R[n + 4] - np.eye(3)).max() == 0  # zero matrix -> identity
    rv = np.ascontiguousarray(g['rotvec_in'])
    M = np.zeros((len(rv), 3, 3), np.float32)
    lib.hostemu_rotvec2mat(H._p(rv), H._p(M), len(rv))
    assert np.abs(M - g['rotvec2mat_out']).max() < 1e-6
    Rin = np.ascontiguousarray(g['rotvec2mat_out'])
    out = np.zeros((len(Rin), 3), np.float32)
    lib.hostemu_mat2rotvec(H._p(Rin), H._p(out), len(Rin))
    assert np.abs(out - g['mat2rotvec_out']).max() < 1e-5
    a, b = np.ascontiguousarray(g['align_a']), np.ascontiguousarray(g['align_b'])
    Ra = np.zeros((len(a), 3, 3), np.float32)
    lib.hostemu_align(H._p(a), H._p(b), H._p(Ra), len(a))
    assert np.abs(Ra[:-4] - g['align_out'][:-4]).max() < 1e-6
    assert np.abs(Ra[-4:] - np.eye(3)).max() == 0  # exactly antiparallel: zero rotvec -> identity


def test_proj_so3_against_fp64_svd():
    """sf::proj_so3 (round 6: Horn's closed form, Jacobi sweeps behind it) against the nearest rotation from numpy's fp64
    SVD with the reference's reflection fix (pt/rotation.py:100-110) on sets where that rotation is unique: random,
    badly scaled, nearly singular, reflected (det < 0 with separated singular values) and rank 2."""
    lib = H.load()
    rs = np.random.RandomState(5)

    def with_sv(sv, reflect):
        n = len(sv)
        U, _ = np.linalg.qr(rs.randn(n, 3, 3))
        V, _ = np.linalg.qr(rs.randn(n, 3, 3))
        U[:, :, 2] *= np.sign(np.linalg.det(U))[:, None]
        V[:, :, 2] *= np.sign(np.linalg.det(V))[:, None]
        sv = np.array(sv, np.float64)
        if reflect:
            sv[:, 2] *= -1
        return U @ (sv[:, :, None] * np.swapaxes(V, -1, -2))

    n = 2000
    sets = {
        'random': rs.randn(n, 3, 3),
        'scaled': rs.randn(n, 3, 3) * 10.0 ** rs.uniform(-4, 4, size=(n, 1, 1)),
        'near_singular': with_sv(np.stack([rs.uniform(0.5, 2, n), rs.uniform(0.1, 0.4, n), rs.uniform(1e-5, 1e-3, n)], 1), False),
        'reflected': with_sv(np.stack([rs.uniform(1, 2, n), rs.uniform(0.5, 0.9, n), rs.uniform(0.05, 0.4, n)], 1), True),
        'rank2': with_sv(np.stack([rs.uniform(1, 2, n), rs.uniform(0.3, 0.9, n), np.zeros(n)], 1), False),
    }
    for name, A64 in sets.items():
        A = np.ascontiguousarray(A64.astype(np.float32))
        R = np.zeros_like(A)
        lib.hostemu_proj_so3(H._p(A), H._p(R), len(A))
        U, s, Vt = np.linalg.svd(A.astype(np.float64))
        d = np.sign(np.linalg.det(U @ Vt))
        U[:, :, 2] *= d[:, None]
        ref = U @ Vt
        # where the two smallest singular values (with the reflection's sign) nearly cancel, the answer is ill-defined
        ok = (s[:, 1] + d * s[:, 2]) > 1e-3 * s[:, 0]
        assert ok.mean() > 0.9, name
        err = np.abs(R.astype(np.float64) - ref)[ok].max()
        assert err < 2e-6, (name, err)  # (3e-8 observed on every set: the rounding of the fp32 outputs)
        assert np.abs(np.linalg.det(R.astype(np.float64)) - 1).max() < 1e-5, name


@pytest.mark.parametrize('name', ['smpl', 'smplx', 'smpl1024'])
def test_forward(name, model_root, golden):
    g = golden(name)
    kind, md = util.load_md(model_root, name, g)
    fw = H.forward(md, kind, g['pose'], g['betas'], g['trans'])
    assert np.abs(fw['vertices'] - g['target_vertices']).max() < 2e-6
    assert np.abs(fw['joints'] - g['fwd_joints']).max() < 2e-6
    assert np.abs(fw['orientations'] - g['fwd_orientations']).max() < 1e-6
    # global-rotation input reproduces the same mesh
    fw2 = H.forward(md, kind, None, g['betas'], g['trans'], glob=g['fwd_orientations'])
    assert np.abs(fw2['vertices'] - g['target_vertices']).max() < 5e-6


@pytest.mark.parametrize('name', ['smpl', 'smplx', 'smpl1024', 'smplxfat', *util.SKIN_KINDS])
def test_fit_goldens(name, model_root, golden):
    g = golden(name)
    kind, md = util.load_md(model_root, name, g)
    om64, of64 = util.make_oracle(md, kind, np.float64)
    # 3e-4 on the well-conditioned fixtures (SMPL, fat-part SMPL-X); the thin-finger SMPL-X one is
    # ill-conditioned in the reference itself and is judged on vertices
    pose_tol = util.pose_tol(name)
    G0 = None
    for c in util.fit_configs(g):
        cfg = util.cfg_from_name(c)
        if not cfg['joints'] and name == 'smpl1024':
            continue
        o = H.fit(
            md, kind, g['target_vertices'], g['target_joints'] if cfg['joints'] else None,
            g['vertex_weights'] if cfg['weights'] else None,
            g['joint_weights'] if (cfg['weights'] and cfg['joints']) else None,
            num_iter=cfg['num_iter'], beta_regularizer=cfg['beta_regularizer'],
            final_adjust_rots=cfg['final_adjust_rots'],
        )
        ref = {k: g[f'fit.{c}.{k}'] for k in ('pose_rotvecs', 'shape_betas', 'trans', 'orientations')}
        assert util.vertex_l2(om64, o, ref) < 1e-4, c
        assert np.abs(o['shape_betas'] - ref['shape_betas']).max() < 3e-4, c
        assert np.abs(o['trans'] - ref['trans']).max() < 1e-5, c
        assert np.abs(o['pose_rotvecs'] - ref['pose_rotvecs']).max() < pose_tol, c
        assert np.abs(o['orientations'] - ref['orientations']).max() < pose_tol, c
        if cfg['joints'] and not cfg['weights']:
            G0 = o['glob_rotmats_iter0']
    if 'stage.glob_rotmats_iter0' in g:
        assert np.abs(G0 - g['stage.glob_rotmats_iter0']).max() < (2e-3 if name == 'smplx' else 5e-4)


@pytest.mark.parametrize('name', ['smpl'])
def test_kid_goldens(name, model_root, golden):
    g = golden(name)
    kind, md = util.load_md(model_root, name, g)
    om, _ = util.make_oracle(md, kind, np.float64)
    fw = H.forward(md, kind, g['pose'], g['betas'], g['trans'], kid=g['kid'])
    assert np.abs(fw['vertices'] - g['kid.target_vertices']).max() < 2e-6
    assert np.abs(fw['joints'] - g['kid.fwd_joints']).max() < 2e-6
    cfgs = dict(
        a=dict(num_iter=3, beta_regularizer=1.0, use_joints=True),
        b=dict(num_iter=1, beta_regularizer=0.0, final_adjust_rots=False, kid_regularizer=1e9, use_joints=False),
    )
    for tag, kw in cfgs.items():
        kw = dict(kw)
        uj = kw.pop('use_joints')
        o = H.fit(md, kind, g['kid.target_vertices'], g['kid.target_joints'] if uj else None,
                  enable_kid=True, **kw)
        ref = {k: g[f'kidfit.{tag}.{k}'] for k in ('pose_rotvecs', 'shape_betas', 'trans', 'kid_factor')}
        va = om.forward(o['pose_rotvecs'], o['shape_betas'], o['trans'], kid_factor=o['kid_factor'])['vertices']
        vb = om.forward(ref['pose_rotvecs'], ref['shape_betas'], ref['trans'], kid_factor=ref['kid_factor'])['vertices']
        assert np.linalg.norm(va - vb, axis=-1).max() < 1e-4, tag
        assert np.abs(o['shape_betas'] - ref['shape_betas']).max() < 1e-3, tag
        assert np.abs(o['kid_factor'] - ref['kid_factor']).max() < 1e-3, tag
        assert np.abs(o['trans'] - ref['trans']).max() < 2e-5, tag


@pytest.mark.parametrize('name', ['smpl', 'smplx'])
def test_known_shape_goldens(name, model_root, golden):
    """fit_with_known_shape through the shared stage code (forward joint stage, part rotations,
    scale/translation alignment, refinement) against the reference's fixture."""
    g, ge = golden(name), golden(f'ext_{name}')
    kind, md = util.load_md(model_root, name, g)
    om, _ = util.make_oracle(md, kind, np.float64)
    for case in util.KNOWN_SHAPE_CASES:
        if f'knownshape.{case}.trans' not in ge:
            continue
        betas, tv, kw = util.known_shape_inputs(g, case)
        o = H.fit_known_shape(md, kind, betas, tv, **kw)
        util.check_known_shape(om, name, case, o, ge, betas, kw)


@pytest.mark.parametrize('name', ['smpl', 'smplx'])
def test_warm_start_goldens(name, model_root, golden):
    """Warm-started fit (initial pose / betas / kid + ridge references) through the shared stage code."""
    g, ge = golden(name), golden(f'ext_{name}')
    kind, md = util.load_md(model_root, name, g)
    om, _ = util.make_oracle(md, kind, np.float64)
    for case in util.WARM_CASES:
        if f'warm.{case}.trans' not in ge:
            continue
        kid_fit, tv, kw = util.warm_inputs(g, case)
        o = H.fit_warm(md, kind, tv, enable_kid=kid_fit, **kw)
        util.check_warm(om, name, case, o, ge, kid_fit)


@pytest.mark.parametrize('name', ['smpl', 'smplx'])
def test_share_beta_goldens(name, model_root, golden):
    """share_beta through the shared solve stage (assemble / sum / solve the sum) against the reference."""
    g, ge = golden(name), golden(f'ext_{name}')
    kind, md = util.load_md(model_root, name, g)
    om, _ = util.make_oracle(md, kind)
    for case in util.SHARE_CASES:
        if f'share.{case}.trans' not in ge:
            continue
        kid_fit, tv, kw = util.share_inputs(g, om, case)
        o = H.fit_warm(md, kind, tv, enable_kid=kid_fit, share_beta=True, **kw)
        util.check_share(om, name, case, o, ge, kid_fit)
    gk = golden(f'kp_{name}')
    if 'sharewarm.a.trans' in gk:  # share_beta + warm start: the ridge reference is dropped (pt/lstsq.py:45-47)
        _, tv, kw = util.warm_inputs(g, 'a')
        o = H.fit_warm(md, kind, tv, share_beta=True, **kw)
        util.check_share(om, name, 'a', o, gk, False, prefix='sharewarm')


@pytest.mark.parametrize('name', ['smpl', 'smplx'])
def test_scale_goldens(name, model_root, golden):
    """scale_target / scale_fit through the shared stage code (extra vertex sums, scaled solve stage,
    scaled refinement inputs) against the reference."""
    g, ge = golden(name), golden(f'ext_{name}')
    kind, md = util.load_md(model_root, name, g)
    om, _ = util.make_oracle(md, kind)
    for case in util.SCALE_CASES:
        if f'scale.{case}.trans' not in ge:
            continue
        kid_fit, tv, kw = util.scale_inputs(g, case)
        o = H.fit_warm(md, kind, tv, enable_kid=kid_fit, **kw)
        util.check_scale(om, name, case, o, ge, kid_fit)


@pytest.mark.parametrize('name', ['smpl', 'smplx'])
def test_share_scale_goldens(name, model_root, golden):
    """share_beta with a scale unknown through the shared stage code: the Schur-reduced systems of the
    scaled solve stage, their sum, the shared solve with every instance's own scale."""
    g, gk = golden(name), golden(f'kp_{name}')
    kind, md = util.load_md(model_root, name, g)
    om, _ = util.make_oracle(md, kind)
    n = 0
    for case in util.SHARE_SCALE_CASES:
        if f'sharescale.{case}.trans' not in gk:
            continue
        kid_fit, tv, kw = util.share_scale_inputs(g, om, case)
        o = H.fit_warm(md, kind, tv, enable_kid=kid_fit, share_beta=True, **kw)
        util.check_share_scale(om, name, case, o, gk, kid_fit)
        n += 1
    assert n >= 1


@pytest.mark.parametrize('nb', [6, 13])
def test_num_betas_padding(nb, model_root, golden):
    """num_betas = 6 / 13: the tables pad the shape unknowns to 10 / 16 and pin the padding with a unit
    ridge; the caller sees its own count (reference fixture golden_nb_smpl.npz)."""
    from smplfitter_amd import modelio

    gnb = golden('nb_smpl')
    md = modelio.load_model('smpl', 'neutral', model_root=f'{model_root}/{util.NB_DIR[nb]}', num_betas=nb)
    om64, _ = util.make_oracle(md, 'smpl', np.float64)
    fw = H.forward(md, 'smpl', gnb[f'nb{nb}.pose'], gnb[f'nb{nb}.betas'], gnb[f'nb{nb}.trans'])
    assert np.abs(fw['vertices'][:, ::50] - gnb[f'nb{nb}.fwd_vertices_every_50th']).max() < 2e-6
    assert np.abs(fw['joints'] - gnb[f'nb{nb}.fwd_joints']).max() < 2e-6
    for nb2, kid, cfg in util.NB_CASES:
        if nb2 != nb:
            continue
        o = H.fit(md, 'smpl', gnb[f'nb{nb}.target_vertices'], gnb[f'nb{nb}.target_joints'], enable_kid=kid, **util.NB_CFG[cfg])
        util.check_nb(om64, gnb, nb, kid, cfg, o)
