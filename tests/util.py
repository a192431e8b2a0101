"""Shared helpers for the parity tests (oracle construction, metrics)."""

import os
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
    return kind, modelio.load_model(kind, 'neutral', model_root=f'{root}/{model_dir(name)}', num_betas=10, **kw)


# num_betas outside the counts the kernels are instantiated for (tests/golden/make_golden_nb.py)
NB_DIR = {6: 'smpl', 13: 'smpl_b16'}  # model directory under the synthetic root
NB_CASES = [(nb, kid, cfg) for nb in (6, 13) for kid in (False, True) for cfg in ('it3_reg1', 'it2_reg0')]
NB_CFG = dict(it3_reg1=dict(num_iter=3, beta_regularizer=1.0),
              it2_reg0=dict(num_iter=2, beta_regularizer=0.0, final_adjust_rots=False))


def check_nb(om64, gnb, nb, kid, cfg, o):
    """Assertions of a fit with ``nb`` betas against the reference's fixture: mesh gate 1e-4 m."""
    pre = f'nb{nb}.kid{int(kid)}.{cfg}.'
    ref = {k: gnb[pre + k] for k in ('pose_rotvecs', 'shape_betas', 'trans') + (('kid_factor',) if kid else ())}
    assert o['shape_betas'].shape == ref['shape_betas'].shape == (ref['trans'].shape[0], nb)
    kw_o = dict(kid_factor=o['kid_factor']) if kid else {}
    kw_r = dict(kid_factor=ref['kid_factor']) if kid else {}
    va = om64.forward(o['pose_rotvecs'], o['shape_betas'], o['trans'], **kw_o)['vertices']
    vb = om64.forward(ref['pose_rotvecs'], ref['shape_betas'], ref['trans'], **kw_r)['vertices']
    assert np.linalg.norm(va - vb, axis=-1).max() < 1e-4, (nb, kid, cfg)
    assert np.abs(o['trans'] - ref['trans']).max() < 2e-5, (nb, kid, cfg)
    assert np.abs(o['shape_betas'] - ref['shape_betas']).max() < 2e-3, (nb, kid, cfg)


# models of the library's GENERAL path (tests/golden/make_golden_general.py): kind -> num_betas (None: every column)
GENERAL_KINDS = {'smpl_b32': 32, 'smpl_b300': None, 'smpl_w12': 10}
GENERAL_CASES = {
    'it3_reg1_j_nw_fa': dict(joints=True, weights=False, kw=dict(num_iter=3, beta_regularizer=1.0)),
    'it2_reg0_nj_nw_nfa': dict(joints=False, weights=False, kw=dict(num_iter=2, beta_regularizer=0.0, final_adjust_rots=False)),
    'it2_reg1_j_w_fa': dict(joints=True, weights=True, kw=dict(num_iter=2, beta_regularizer=1.0)),
    # what BodyConverter.convert runs (pt/bodyconverter.py:74-88): enable_kid fitter
    'conv': dict(joints=False, weights=False, kid=True,
                 kw=dict(num_iter=1, beta_regularizer=0.0, final_adjust_rots=False, kid_regularizer=1e9)),
}


# further models of the general path without a fixture of their own (checked against the fp64 oracle): 100 betas — the
# accumulate kernel's sixteen-wave workgroup shape — and 400 — its 91 blocks split over two workgroups per instance
GENERAL_SHAPE_KINDS = {'smpl_b100': 100, 'smpl_b400': 400}


def general_num_betas(kind):
    return GENERAL_KINDS[kind] if kind in GENERAL_KINDS else GENERAL_SHAPE_KINDS[kind]


def load_general_md(root, kind):
    return modelio.load_model('smpl', 'neutral', model_root=f'{root}/{kind}', num_betas=general_num_betas(kind))


# options of fit on the general path (tests/golden/make_golden_general_opts.py): the cases of SCALE_CASES / SHARE_CASES /
# SHARE_SCALE_CASES run per model (case d of SCALE_CASES — kid + scale, nearly collinear on the synthetic model — is left
# to the fixtures of the other paths)
GENERAL_OPT_KINDS = {
    'smpl_b32': dict(scale=('a', 'b', 'c', 'e'), share=('a', 'b', 'c'), sharescale=('a', 'c')),
    'smpl_w12': dict(scale=('a', 'c'), share=('b',), sharescale=()),
}


def general_view(gg, kind):
    """The arrays of one model of golden_general.npz under the names the fixtures of the other paths use."""
    pre = kind + '.'
    return {k[len(pre):]: v for k, v in gg.items() if k.startswith(pre) and '.fit.' not in k}


def general_oracle(root, kind, dtype=np.float32):
    return O.OracleModel(load_general_md(root, kind), dtype, 'smpl')


def general_fit_args(gg, kind, case):
    """(target_vertices, keyword arguments) of a fixture case, as numpy arrays."""
    c = GENERAL_CASES[case]
    pre = kind + '.'
    kw = dict(c['kw'])
    kw['target_joints'] = gg[pre + 'target_joints'] if c['joints'] else None
    if c['weights']:
        kw['vertex_weights'] = gg[pre + 'vertex_weights']
        kw['joint_weights'] = gg[pre + 'joint_weights']
    return gg[pre + 'target_vertices'], kw


def check_general(om64, gg, kind, case, o, mesh_tol=1e-4):
    """A fit of a general-path model against the reference's fixture: the mesh gate of every other fixture."""
    pre = f'{kind}.fit.{case}.'
    kid = GENERAL_CASES[case].get('kid', False)
    ref = {k: gg[pre + k] for k in ('pose_rotvecs', 'shape_betas', 'trans') + (('kid_factor',) if kid else ())}
    assert o['shape_betas'].shape == ref['shape_betas'].shape
    kw_o = dict(kid_factor=np.asarray(o['kid_factor'])) if kid else {}
    kw_r = dict(kid_factor=ref['kid_factor']) if kid else {}
    va = om64.forward(np.asarray(o['pose_rotvecs']), np.asarray(o['shape_betas']), np.asarray(o['trans']), **kw_o)['vertices']
    vb = om64.forward(ref['pose_rotvecs'], ref['shape_betas'], ref['trans'], **kw_r)['vertices']
    err = np.linalg.norm(va - vb, axis=-1).max()
    db, dt = np.abs(o['shape_betas'] - ref['shape_betas']).max(), np.abs(o['trans'] - ref['trans']).max()
    print(f'[general] {kind:10s} {case:20s} vtx {err:.2e} betas {db:.2e} trans {dt:.2e}')
    assert err < mesh_tol, (kind, case, err)
    assert dt < 5e-5, (kind, case, dt)
    # the shape itself (round 6, advisor): up to 4.6e-4 observed at S = 300 — low-energy directions of an fp32 300 x 300
    # Gramian; the same gate as check_nb.  A wrong beta on such a direction moves no vertex, so the mesh gate cannot see it.
    assert db < 2e-3, (kind, case, db)
    return err


def general_arbiter(om64, gg, kind, case, o):
    """Whose shape is closer to the fp64 evaluation of the algorithm, ours or the reference's fp32 fixture?  Prints the
    line profiles/r06_general_path_parity.txt keeps; gates ours at 2 x the reference's own distance (+ 1e-4)."""
    tv, kw = general_fit_args(gg, kind, case)
    r64 = O.OracleFitter(om64).fit(tv.astype(np.float64), kw.pop('target_joints'), **{k: v for k, v in kw.items()})
    ref = gg[f'{kind}.fit.{case}.shape_betas']
    d_ours = np.abs(np.asarray(o['shape_betas']) - r64['shape_betas']).max()
    d_ref = np.abs(ref - r64['shape_betas']).max()
    v64 = om64.forward(r64['pose_rotvecs'], r64['shape_betas'], r64['trans'])['vertices']
    vo = om64.forward(np.asarray(o['pose_rotvecs']), np.asarray(o['shape_betas']), np.asarray(o['trans']))['vertices']
    vr = om64.forward(gg[f'{kind}.fit.{case}.pose_rotvecs'], ref, gg[f'{kind}.fit.{case}.trans'])['vertices']
    e_ours, e_ref = np.linalg.norm(vo - v64, axis=-1).max(), np.linalg.norm(vr - v64, axis=-1).max()
    print(f'[general-arbiter] {kind:10s} {case:20s} betas ours-vs-f64 {d_ours:.2e} reference-vs-f64 {d_ref:.2e} | '
          f'vtx ours-vs-f64 {e_ours:.2e} reference-vs-f64 {e_ref:.2e}')
    assert d_ours < 2 * d_ref + 1e-4, (kind, case, d_ours, d_ref)
    assert e_ours < 2 * e_ref + 2e-5, (kind, case, e_ours, e_ref)


def model_dir(name):
    """Directory of golden set ``name`` under the synthetic model root (smplxfat: the fat-part SMPL-X
    variant of synth.make_model_arrays('smplx_fat'), its own directory, the official SMPL-X file name; the
    skinning variants smpl_w6 / smplx_w6 / smpl_rnd of tests/golden/make_golden_skin.py likewise)."""
    if name in SKIN_KINDS:
        return name
    return 'smplx_fat' if name == 'smplxfat' else ('smplx' if name.startswith('smplx') else 'smpl')


def pose_tol(name):
    """Gate on pose_rotvecs / orientations against the reference's fixture: 3e-4 on the well-conditioned fixtures
    (SMPL-shaped, the fat-part SMPL-X); the thin-finger SMPL-X-shaped ones are ill-conditioned in the reference itself
    (the bone parts' twist comes from the vertices' off-axis spread) and are judged on the vertices; 1.5e-3 on the
    1024-vertex subset."""
    if name in ('smplx', 'smplx_w6'):
        # (round 5, B = 8 fixture: the reference's own fp32 result sits 7.9e-4 from the fp64 evaluation of its algorithm
        # on this ill-conditioned fixture; two independent fp32 evaluations — the reference's and ours — are up to 2.0e-3
        # apart (host emulation, it1_reg0_j_nw_nfa).  Rounds 1-4 gated this at 5e-3.)
        return 3e-3
    return 3e-4 if name in ('smpl', 'smplxfat', 'smpl_w6', 'smpl_rnd') else 1.5e-3


# skinning variants (synth.make_model_arrays): six weights per vertex (KW = 8 kernels), random joint sets
SKIN_KINDS = ('smpl_w6', 'smplx_w6', 'smpl_rnd')


def stats(a, b):
    """max / p99 / median of |a - b| (SURVEY.md §8d asks for all three on pose_rotvecs)."""
    d = np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).reshape(-1)
    return dict(max=float(d.max()), p99=float(np.percentile(d, 99)), median=float(np.median(d)))


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


# ---- inputs of the extended fixtures (tests/golden/make_golden_ext.py), shared with the tests ------
# case -> (num_iter, joints, weights, scale_fit, kid, initial_pose, final_adjust)
KNOWN_SHAPE_CASES = {
    'a': (1, True, False, False, False, False, True),
    'b': (3, False, True, False, False, False, True),
    'c': (2, True, True, True, False, False, True),
    'd': (2, True, False, False, True, True, False),
    'e': (3, True, False, True, False, True, True),
    'f': (1, False, False, False, False, False, True),
}


def known_shape_inputs(g, case):
    """Inputs of a case, derived deterministically from the base fixture (shared with the tests)."""
    num_iter, joints, weights, scale_fit, kid, init, final = KNOWN_SHAPE_CASES[case]
    tv = g['kid.target_vertices'] if kid else g['target_vertices']
    tj = g['kid.target_joints'] if kid else g['target_joints']
    if scale_fit:  # a target that really is a scaled body
        tv, tj = tv * np.float32(1.1), tj * np.float32(1.1)
    kw = dict(num_iter=num_iter, final_adjust_rots=final, scale_fit=scale_fit)
    kw['target_joints'] = tj if joints else None
    kw['vertex_weights'] = g['vertex_weights'] if weights else None
    kw['joint_weights'] = g['joint_weights'] if (weights and joints) else None
    kw['kid_factor'] = g['kid'] if kid else None
    if init:
        rs = np.random.RandomState(99)
        kw['initial_pose_rotvecs'] = (g['pose'] + rs.randn(*g['pose'].shape) * 0.05).astype(np.float32)
    else:
        kw['initial_pose_rotvecs'] = None
    return g['betas'], tv, kw


# case -> (joints, weights, scale)
SCALE_TRANS_CASES = dict(a=(True, False, False), b=(True, True, True), c=(False, True, True),
                         d=(False, False, True))


def scale_trans_inputs(g):
    tv, tj = g['target_vertices'], g['target_joints']
    rv = np.ascontiguousarray(tv[::-1]) * np.float32(0.9) + np.float32(0.05)  # another body, scaled and shifted
    rj = np.ascontiguousarray(tj[::-1]) * np.float32(0.9) + np.float32(0.05)
    return tv, tj, rv, rj, g['vertex_weights'], g['joint_weights']


def check_known_shape(om, name, case, o, ge, betas, kw):
    """Shared assertions of the known-shape fit against the reference's fixture (golden_ext_*.npz):
    the mesh posed with the result is the gate (1e-4 m)."""
    ref = {k: ge[f'knownshape.{case}.{k}'] for k in ('pose_rotvecs', 'trans', 'orientations')}
    assert np.abs(o['trans'] - ref['trans']).max() < 2e-5, case
    assert np.abs(o['orientations'] - ref['orientations']).max() < (1.5e-3 if name == 'smpl' else 5e-3), case
    if kw['scale_fit']:
        assert np.abs(o['scale_corr'] - ge[f'knownshape.{case}.scale_corr'].reshape(-1)).max() < 1e-5, case
    kid = kw['kid_factor']
    va = om.forward(o['pose_rotvecs'], betas, o['trans'], kid_factor=kid)['vertices']
    vb = om.forward(ref['pose_rotvecs'], betas, ref['trans'], kid_factor=kid)['vertices']
    assert np.linalg.norm(va - vb, axis=-1).max() < 1e-4, case


# warm-started fits (initial_pose_rotvecs / initial_shape_betas / initial_kid_factor, reference
# pt/bodyfitter.py:363-382 and the regulariser references :1072-1081, :1224-1255)
# case -> (enable_kid, joints, fit kwargs, which initial values are given)
WARM_CASES = {
    'a': (False, True, dict(num_iter=2, beta_regularizer=1.0), ('pose', 'betas')),
    'b': (False, False, dict(num_iter=1, beta_regularizer=1.0), ('pose',)),
    'c': (True, False, dict(num_iter=2, beta_regularizer=1e-2, beta_regularizer2=1e-2,
                            final_adjust_rots=True, kid_regularizer=1e9), ('pose', 'betas')),
    'd': (True, True, dict(num_iter=3, beta_regularizer=2.0, kid_regularizer=0.5), ('pose', 'betas', 'kid')),
    'e': (False, True, dict(num_iter=1, beta_regularizer=5.0, final_adjust_rots=False), ('betas',)),
    # initial_kid_factor ALONE: no warm first pass, but the kid ridge still pulls towards it (:413-414)
    'f': (True, True, dict(num_iter=2, beta_regularizer=1.0, kid_regularizer=0.7), ('kid',)),
}


def warm_inputs(g, case):
    kid_fit, joints, kw, given = WARM_CASES[case]
    rs = np.random.RandomState(321)
    pose0 = (g['pose'] + rs.randn(*g['pose'].shape) * 0.05).astype(np.float32)
    betas0 = (g['betas'] + rs.randn(*g['betas'].shape) * 0.2).astype(np.float32)
    kid0 = (g['kid'] + rs.randn(*g['kid'].shape) * 0.1).astype(np.float32)
    use_kid_target = kid_fit and 'kid' in given
    tv = g['kid.target_vertices'] if use_kid_target else g['target_vertices']
    tj = g['kid.target_joints'] if use_kid_target else g['target_joints']
    kw = dict(kw)
    kw['target_joints'] = tj if joints else None
    kw['initial_pose_rotvecs'] = pose0 if 'pose' in given else None
    kw['initial_shape_betas'] = betas0 if 'betas' in given else None
    kw['initial_kid_factor'] = kid0 if 'kid' in given else None
    return kid_fit, tv, kw


def check_warm(om, name, case, o, ge, kid_fit):
    """Assertions of a warm-started fit against the reference's fixture (mesh gate 1e-4 m)."""
    keys = ('pose_rotvecs', 'shape_betas', 'trans') + (('kid_factor',) if kid_fit else ())
    ref = {k: ge[f'warm.{case}.{k}'] for k in keys}
    kw_o = dict(kid_factor=o['kid_factor']) if kid_fit else {}
    kw_r = dict(kid_factor=ref['kid_factor']) if kid_fit else {}
    va = om.forward(o['pose_rotvecs'], o['shape_betas'], o['trans'], **kw_o)['vertices']
    vb = om.forward(ref['pose_rotvecs'], ref['shape_betas'], ref['trans'], **kw_r)['vertices']
    assert np.linalg.norm(va - vb, axis=-1).max() < 1e-4, case
    assert np.abs(o['trans'] - ref['trans']).max() < 2e-5, case
    assert np.abs(o['shape_betas'] - ref['shape_betas']).max() < (3e-4 if name == 'smpl' else 1e-3), case
    if kid_fit:
        assert np.abs(o['kid_factor'] - ref['kid_factor']).max() < 1e-3, case


# share_beta fits (one shape for the whole batch; reference pt/lstsq.py:24-26, :32-90)
# case -> (enable_kid, joints, weights, fit kwargs)
SHARE_CASES = {
    'a': (False, True, False, dict(num_iter=3, beta_regularizer=1.0)),
    'b': (False, False, True, dict(num_iter=2, beta_regularizer=0.5)),
    'c': (True, True, False, dict(num_iter=2, beta_regularizer=1.0, kid_regularizer=2.0)),
}


def share_inputs(g, om, case):
    """A batch of ONE body shape in different poses (+3 mm noise), built from the fixture's pose rows."""
    kid_fit, joints, weights, kw = SHARE_CASES[case]
    rs = np.random.RandomState(77)
    B = g['pose'].shape[0]
    betas = np.repeat(g['betas'][:1], B, 0)
    fw = om.forward(g['pose'], betas, g['trans'])
    tv = (fw['vertices'] + rs.randn(*fw['vertices'].shape) * 0.003).astype(np.float32)
    kw = dict(kw)
    kw['target_joints'] = fw['joints'].astype(np.float32) if joints else None
    kw['vertex_weights'] = g['vertex_weights'] if weights else None
    kw['joint_weights'] = g['joint_weights'] if (weights and joints) else None
    return kid_fit, tv, kw


def check_share(om, name, case, o, ge, kid_fit, prefix='share'):
    """share_beta fit against the reference's fixture.  The shared-shape pipeline amplifies fp32
    reduction noise (the reference documents ~2e-3 in pose_rotvecs at the ankles, pt/bodyfitter.py
    :250-255), so the mesh gate is 5e-4 m here and the shared shape itself is pinned tightly."""
    keys = ('pose_rotvecs', 'shape_betas', 'trans') + (('kid_factor',) if kid_fit else ())
    ref = {k: ge[f'{prefix}.{case}.{k}'] for k in keys}
    assert np.abs(o['shape_betas'] - o['shape_betas'][:1]).max() == 0, case  # one shape for the batch
    assert np.abs(o['shape_betas'] - ref['shape_betas']).max() < (1e-3 if name == 'smpl' else 3e-3), case
    assert np.abs(o['trans'] - ref['trans']).max() < 1e-4, case
    kw_o = dict(kid_factor=o['kid_factor']) if kid_fit else {}
    kw_r = dict(kid_factor=ref['kid_factor']) if kid_fit else {}
    va = om.forward(o['pose_rotvecs'], o['shape_betas'], o['trans'], **kw_o)['vertices']
    vb = om.forward(ref['pose_rotvecs'], ref['shape_betas'], ref['trans'], **kw_r)['vertices']
    assert np.linalg.norm(va - vb, axis=-1).max() < 5e-4, case


# scale_target / scale_fit of fit (the extra unknown of the LAST shape solve, reference
# pt/bodyfitter.py:1170-1175, :1284-1296, driver :434-519; the reference's TestFitterWithScale)
# case -> (enable_kid, scale option, joints, weights, fit kwargs)
SCALE_CASES = {
    'a': (False, 'scale_target', True, False, dict(num_iter=3, beta_regularizer=0.0)),
    'b': (False, 'scale_fit', True, False, dict(num_iter=3, beta_regularizer=0.0)),
    'c': (False, 'scale_fit', False, True, dict(num_iter=2, beta_regularizer=1.0, scale_regularizer=0.5)),
    'd': (True, 'scale_target', True, True, dict(num_iter=2, beta_regularizer=1.0)),
    'e': (False, 'scale_target', False, False, dict(num_iter=1, beta_regularizer=1.0, final_adjust_rots=False)),
}


def scale_inputs(g, case):
    kid_fit, opt, joints, weights, kw = SCALE_CASES[case]
    f = np.float32(1.1)
    kw = dict(kw)
    kw[opt] = True
    kw['target_joints'] = g['target_joints'] * f if joints else None
    kw['vertex_weights'] = g['vertex_weights'] if weights else None
    kw['joint_weights'] = g['joint_weights'] if (weights and joints) else None
    return kid_fit, g['target_vertices'] * f, kw


def check_scale(om, name, case, o, ge, kid_fit, loose=1.0):
    """``loose``: gate multiplier (3 for the fp32 numpy oracle, whose (S + 1)-unknown solve without a ridge is noisier than
    the reference's own fp32: 1.4e-4 m on case b of the B = 8 fixture where its fp64 form and the HIP arithmetic sit at
    1e-5)."""
    keys = ('pose_rotvecs', 'shape_betas', 'trans', 'scale_corr') + (('kid_factor',) if kid_fit else ())
    ref = {k: ge[f'scale.{case}.{k}'] for k in keys}
    if kid_fit:
        # kid AND scale unknowns (case d): on the synthetic model the kid direction (0.6 x the template) is almost a pure
        # scaling, so the two unknowns are nearly collinear and the reference's fp32 solve amplifies its own rounding — on
        # the round-5 (B = 8) fixture its result sits 1.1e-4 m (mesh), 6.6e-5 (trans), 1.0e-4 (kid) from the fp64
        # evaluation of the same algorithm, and an independent fp32 evaluation 3.5e-4 m from it.  Gates: 4x that distance.
        assert np.abs(o['scale_corr'] - ref['scale_corr']).max() < 4e-4, case
        assert np.abs(o['trans'] - ref['trans']).max() < 3e-4, case
        assert np.abs(o['shape_betas'] - ref['shape_betas']).max() < 1e-3, case
        va = om.forward(o['pose_rotvecs'], o['shape_betas'], o['trans'], kid_factor=o['kid_factor'])['vertices']
        vb = om.forward(ref['pose_rotvecs'], ref['shape_betas'], ref['trans'], kid_factor=ref['kid_factor'])['vertices']
        assert np.linalg.norm(va - vb, axis=-1).max() < 4.5e-4, case
        return
    assert np.abs(o['scale_corr'] - ref['scale_corr']).max() < 1e-4 * loose, case  # the reference solves this system in fp32
    assert np.abs(o['trans'] - ref['trans']).max() < 5e-5 * loose, case
    assert np.abs(o['shape_betas'] - ref['shape_betas']).max() < (1e-3 if name == 'smpl' else 3e-3), case
    kw_o = dict(kid_factor=o['kid_factor']) if kid_fit else {}
    kw_r = dict(kid_factor=ref['kid_factor']) if kid_fit else {}
    va = om.forward(o['pose_rotvecs'], o['shape_betas'], o['trans'], **kw_o)['vertices']
    vb = om.forward(ref['pose_rotvecs'], ref['shape_betas'], ref['trans'], **kw_r)['vertices']
    assert np.linalg.norm(va - vb, axis=-1).max() < 1e-4 * loose, case


# fit_with_known_pose with the options of the general shape solve (reference pt/bodyfitter.py:552-653 ->
# _fit_shape_general :1104-1319): share_beta, scale_target / scale_fit, ridge references.
# case -> (enable_kid, joints, weights, options)
KNOWN_POSE_CASES = {
    'a': (False, True, False, dict(share_beta=True, beta_regularizer=1.0)),
    'b': (False, True, False, dict(scale_target=True, beta_regularizer=0.0)),
    'c': (False, False, True, dict(scale_fit=True, beta_regularizer=1.0, scale_regularizer=0.5)),
    'd': (True, True, True, dict(beta_regularizer=1.0, kid_regularizer=2.0, refs='bk')),
    'e': (True, True, False, dict(share_beta=True, beta_regularizer=0.5, refs='bk')),
    'f': (True, False, False, dict(scale_target=True, beta_regularizer=1.0, scale_regularizer=0.1, refs='b')),
    'g': (False, True, False, dict(beta_regularizer=2.0, beta_regularizer2=0.5, refs='b4')),
    # shared shape + one scale per instance: the partially shared solve (pt/lstsq.py:50-90)
    'h': (False, True, False, dict(share_beta=True, scale_target=True, beta_regularizer=1.0)),
    'i': (True, True, True, dict(share_beta=True, scale_fit=True, beta_regularizer=0.5, scale_regularizer=0.3,
                                 refs='bk')),
}


def known_pose_inputs(g, case):
    """pose = the fixture's true pose; targets = the fixture's noisy targets (scaled by 1.1 for the scale
    options); ridge references = seeded offsets of the true betas ('b4': only 4 columns given)."""
    kid_fit, joints, weights, kw = KNOWN_POSE_CASES[case]
    kw = dict(kw)
    refs = kw.pop('refs', '')
    f = np.float32(1.1 if (kw.get('scale_target') or kw.get('scale_fit')) else 1.0)
    rs = np.random.RandomState(123)
    B = g['pose'].shape[0]
    if 'b' in refs:
        br = (g['betas'] + rs.randn(*g['betas'].shape) * 0.2).astype(np.float32)
        kw['beta_regularizer_reference'] = br[:, :4].copy() if '4' in refs else br
    if 'k' in refs:
        kw['kid_regularizer_reference'] = (rs.rand(B) * 0.3).astype(np.float32)
    kw['target_joints'] = g['target_joints'] * f if joints else None
    kw['vertex_weights'] = g['vertex_weights'] if weights else None
    kw['joint_weights'] = g['joint_weights'] if (weights and joints) else None
    return kid_fit, g['pose'], g['target_vertices'] * f, kw


def check_known_pose(name, case, o, gk, kid_fit, tol_scale=1.0):
    """Shape / translation / scale of one solve against the reference's fixture.  The reference solves
    these systems in fp32 (lstsq of the general path), so the gates are 1e-3-class on shape_betas."""
    share = KNOWN_POSE_CASES[case][3].get('share_beta', False)
    if share:
        assert np.abs(o['shape_betas'] - o['shape_betas'][:1]).max() == 0, case
    tb = (1e-3 if name == 'smpl' else 3e-3) * tol_scale
    assert np.abs(o['shape_betas'] - gk[f'knownpose.{case}.shape_betas']).max() < tb, case
    assert np.abs(o['trans'] - gk[f'knownpose.{case}.trans']).max() < 1e-4 * tol_scale, case
    if kid_fit:
        assert np.abs(o['kid_factor'] - gk[f'knownpose.{case}.kid_factor']).max() < 2e-3 * tol_scale, case
    if f'knownpose.{case}.scale_corr' in gk:
        assert np.abs(o['scale_corr'] - gk[f'knownpose.{case}.scale_corr']).max() < 1e-4 * tol_scale, case
    else:
        assert 'scale_corr' not in o or o['scale_corr'] is None


# fit(share_beta=True) with a scale unknown: all-shared solves, then the partially shared LAST solve
# (pt/lstsq.py:50-90).  case -> (enable_kid, scale option, weights, warm start, fit kwargs)
SHARE_SCALE_CASES = {
    'a': (False, 'scale_target', False, False, dict(num_iter=2, beta_regularizer=1.0)),
    'b': (True, 'scale_fit', True, True, dict(num_iter=2, beta_regularizer=0.5, scale_regularizer=0.2,
                                              kid_regularizer=2.0)),
    'c': (False, 'scale_fit', False, False, dict(num_iter=1, beta_regularizer=0.0, final_adjust_rots=False)),
}


def share_scale_inputs(g, om, case):
    """One body shape in the fixture's poses (+3 mm noise), scaled by 1.1; joints always given."""
    kid_fit, opt, weights, warm, kw = SHARE_SCALE_CASES[case]
    rs = np.random.RandomState(78)
    B = g['pose'].shape[0]
    betas = np.repeat(g['betas'][:1], B, 0)
    fw = om.forward(g['pose'], betas, g['trans'])
    f = np.float32(1.1)
    tv = ((fw['vertices'] + rs.randn(*fw['vertices'].shape) * 0.003) * f).astype(np.float32)
    kw = dict(kw)
    kw[opt] = True
    kw['target_joints'] = (fw['joints'] * f).astype(np.float32)
    kw['vertex_weights'] = g['vertex_weights'] if weights else None
    kw['joint_weights'] = g['joint_weights'] if weights else None
    if warm:
        kw['initial_pose_rotvecs'] = (g['pose'] + rs.randn(*g['pose'].shape) * 0.05).astype(np.float32)
        kw['initial_shape_betas'] = (betas + rs.randn(*betas.shape) * 0.2).astype(np.float32)
    return kid_fit, tv, kw


def check_share_scale(om, name, case, o, gk, kid_fit):
    keys = ('pose_rotvecs', 'shape_betas', 'trans', 'scale_corr') + (('kid_factor',) if kid_fit else ())
    ref = {k: gk[f'sharescale.{case}.{k}'] for k in keys}
    assert np.abs(o['shape_betas'] - o['shape_betas'][:1]).max() == 0, case  # one shape for the batch
    assert np.abs(o['scale_corr'] - ref['scale_corr']).max() < 2e-4, case
    assert np.abs(o['shape_betas'] - ref['shape_betas']).max() < (1e-3 if name == 'smpl' else 3e-3), case
    assert np.abs(o['trans'] - ref['trans']).max() < 2e-4, case
    kw_o = dict(kid_factor=o['kid_factor']) if kid_fit else {}
    kw_r = dict(kid_factor=ref['kid_factor']) if kid_fit else {}
    va = om.forward(o['pose_rotvecs'], o['shape_betas'], o['trans'], **kw_o)['vertices']
    vb = om.forward(ref['pose_rotvecs'], ref['shape_betas'], ref['trans'], **kw_r)['vertices']
    assert np.linalg.norm(va - vb, axis=-1).max() < 5e-4, case


# ---- cross-topology BodyConverter fixture (tests/golden/make_golden_convert.py) -----------------------
CONVERT_DIRS = dict(s2x=('smpl', 'smplxfat'), x2s=('smplxfat', 'smpl'))  # golden-set names (load_md)


def csr_digest(m):
    """sha256 of a scipy CSR matrix as the fixture generator computes it."""
    import hashlib

    h = hashlib.sha256()
    for a in (m.indptr.astype(np.int64), m.indices.astype(np.int64), m.data.astype(np.float32)):
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def load_transfer_csr(data_root, tag):
    """The synthetic transfer matrix of direction ``tag`` in canonical CSR form, read from the official file layout
    (first half of the columns, reference common.py:425-429)."""
    import pickle

    import scipy.sparse as sp

    name = 'smpl2smplx_deftrafo_setup.pkl' if tag == 's2x' else 'smplx2smpl_deftrafo_setup.pkl'
    with open(osp.join(data_root, 'body_models', name), 'rb') as f:
        m = pickle.load(f)['mtx'].tocsr().astype(np.float32)
    m = sp.csr_matrix(m[:, : m.shape[1] // 2].toarray())
    return m


def check_convert(om_out, tag, case, o, gc):
    """Assertions of one ``convert`` result against the reference's cross-topology fixture.  The converted mesh does
    not lie in the output model's space (synthetic models of unrelated shape), so the parameters are only pinned
    through the mesh they produce: max vertex L2 <= 1e-4 m, translation 2e-5."""
    pre = f'{tag}.{case}.'
    keys = sorted(k[len(pre):] for k in gc if k.startswith(pre) and not k.endswith('_in'))
    assert set(o) == set(keys), (tag, case, sorted(o), keys)
    ref = {k: gc[pre + k] for k in keys}
    # kid ridge 0 with beta ridge 0: the kid direction is nearly a combination of the betas' (and of a translation of
    # the mesh), so there only the mesh is pinned
    if case != 'kid.it1':
        assert np.abs(o['trans'] - ref['trans']).max() < 2e-5, (tag, case, float(np.abs(o['trans'] - ref['trans']).max()))
    if case == 'kshape':
        betas = gc[f'{tag}.kshape.betas_in']
        a = dict(pose_rotvecs=o['pose_rotvecs'], shape_betas=betas, trans=o['trans'])
        b = dict(pose_rotvecs=ref['pose_rotvecs'], shape_betas=betas, trans=ref['trans'])
    elif case == 'kpose':
        pose = gc[f'{tag}.kpose.pose_in']
        a = dict(pose_rotvecs=pose, shape_betas=o['shape_betas'], trans=o['trans'])
        b = dict(pose_rotvecs=pose, shape_betas=ref['shape_betas'], trans=ref['trans'])
    else:
        a, b = o, ref
    kw_a = dict(kid_factor=o['kid_factor']) if 'kid_factor' in o else {}
    kw_b = dict(kid_factor=ref['kid_factor']) if 'kid_factor' in ref else {}
    va = om_out.forward(a['pose_rotvecs'], a['shape_betas'], a['trans'], **kw_a)['vertices']
    vb = om_out.forward(b['pose_rotvecs'], b['shape_betas'], b['trans'], **kw_b)['vertices']
    err = float(np.linalg.norm(va - vb, axis=-1).max())
    if os.getenv('SMPLFIT_TEST_VERBOSE'):
        print(f'[convert] {tag} {case}: vertex L2 {err:.2e}, trans {np.abs(o["trans"] - ref["trans"]).max():.1e}')
    assert err < 1e-4, (tag, case, err)
    return err
