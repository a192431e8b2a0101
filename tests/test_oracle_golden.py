"""Pin the CPU oracle against the golden vectors captured from the reference
(tests/golden/make_golden.py).  Tolerances follow BASELINE.md §5: vertices are the gate (1e-4 m);
betas/trans 1e-4-class; pose_rotvecs sits at the reference's own fp32 noise floor."""

import numpy as np
import pytest

import util
from oracle import smplfit_oracle as O
from smplfitter_amd import synth


@pytest.mark.parametrize('name', ['smpl', 'smplx', 'smpl1024'])
def test_model_digest_and_forward(name, model_root, golden):
    g = golden(name)
    kind, md = util.load_md(model_root, name, g)
    assert synth.model_sha256(synth.make_model_arrays(kind, 0)) == str(g['model_sha256'])
    om, _ = util.make_oracle(md, kind)
    fw = om.forward(g['pose'], g['betas'], g['trans'])
    assert np.abs(fw['vertices'][:, ::300] - g['fwd_vertices_sub']).max() < 2e-6
    assert np.abs(fw['joints'] - g['fwd_joints']).max() < 2e-6
    assert np.abs(fw['vertices'] - g['target_vertices']).max() < 2e-6
    assert np.abs(fw['orientations'] - g['fwd_orientations']).max() < 1e-6


@pytest.mark.parametrize('name', ['smpl', 'smplx', 'smpl1024'])
def test_stage_goldens(name, model_root, golden):
    g = golden(name)
    kind, md = util.load_md(model_root, name, g)
    om, of = util.make_oracle(md, kind)
    tv, tj = g['target_vertices'], g['target_joints']
    mean = np.concatenate([tv, tj], 1).mean(1)
    tvc, tjc = tv - mean[:, None], tj - mean[:, None]
    raw, st, sa, sw = of.part_sums(tvc, of.default_mesh[None], None)
    scale = np.abs(g['stage.part_sums.raw']).max()
    assert np.abs(raw - g['stage.part_sums.raw']).max() < 2e-5 * max(scale, 1.0)
    # fp32 sums of ~300 terms in a different order: 3e-5 relative to the largest sum
    assert np.abs(st - g['stage.part_sums.s_t']).max() < 3e-5 * np.abs(g['stage.part_sums.s_t']).max()
    assert np.abs(sa - g['stage.part_sums.s_a']).max() < 3e-5 * np.abs(g['stage.part_sums.s_a']).max()
    assert np.abs(sw - g['stage.part_sums.s_w']).max() == 0
    G0 = of.fit_global_rotations(tvc, tjc, of.default_mesh[None], om.J_template[None], None, None)
    assert np.abs(G0 - g['stage.glob_rotmats_iter0']).max() < (2e-3 if name == 'smplx' else 5e-4)
    # shape solve on the reference's own rotations isolates the solve from rotation noise
    r = of.fit_shape(g['stage.glob_rotmats_iter0'], tvc, tjc, None, None, 1.0, 0.0)
    # fp32 Gramian + centring cancellation: 1e-4-class on the thin-part SMPL-X fixture
    assert np.abs(r['shape_betas'] - g['stage.shape_betas0']).max() < (3e-4 if name == 'smplx' else 5e-5)
    assert np.abs(r['trans'] - g['stage.trans0']).max() < 1e-5
    assert np.abs(r['joints'] - g['stage.joints0']).max() < 2e-5
    assert np.abs(r['vertices'][:, ::300] - g['stage.vertices0_sub']).max() < 2e-5


@pytest.mark.parametrize('name', ['smpl', 'smplx', 'smpl1024', 'smplxfat', *util.SKIN_KINDS])
@pytest.mark.parametrize('dtype', [np.float32, np.float64])
def test_fit_goldens(name, dtype, model_root, golden):
    g = golden(name)
    kind, md = util.load_md(model_root, name, g)
    if name == 'smplxfat':  # the box regenerated the same fat-part model the reference was run on
        assert synth.model_sha256(synth.make_model_arrays('smplx_fat', 0)) == str(g['model_sha256'])
    if name in util.SKIN_KINDS:  # the skinning variants (tests/golden/make_golden_skin.py)
        assert synth.model_sha256(synth.make_model_arrays(name, 0)) == str(g['model_sha256'])
        assert (md.weights != 0).sum(1).max() == int(g['skin_nnz']) == (6 if name.endswith('_w6') else 4)
    om, of = util.make_oracle(md, kind, dtype)
    om64, _ = util.make_oracle(md, kind, np.float64)
    # pose_rotvecs: the thin-finger SMPL-X fixture is ill-conditioned in the reference itself; on the
    # fat-part twin (tests/golden/make_golden_fat.py) and on SMPL the restatement sits within 3e-4
    pose_tol = util.pose_tol(name)
    for c in util.fit_configs(g):
        cfg = util.cfg_from_name(c)
        o = of.fit(
            g['target_vertices'],
            g['target_joints'] if cfg['joints'] else None,
            vertex_weights=g['vertex_weights'] if cfg['weights'] else None,
            joint_weights=g['joint_weights'] if (cfg['weights'] and cfg['joints']) else None,
            num_iter=cfg['num_iter'], beta_regularizer=cfg['beta_regularizer'],
            final_adjust_rots=cfg['final_adjust_rots'],
        )
        ref = {k: g[f'fit.{c}.{k}'] for k in ('pose_rotvecs', 'shape_betas', 'trans', 'orientations')}
        assert util.vertex_l2(om64, o, ref) < 1e-4, c
        assert np.abs(o['shape_betas'] - ref['shape_betas']).max() < 3e-4, c
        assert np.abs(o['trans'] - ref['trans']).max() < 1e-5, c
        assert np.abs(o['pose_rotvecs'] - ref['pose_rotvecs']).max() < pose_tol, c
        assert np.abs(o['orientations'] - ref['orientations']).max() < pose_tol, c


def test_primitive_goldens(golden):
    g = golden('primitives')
    n = int(g['proj_n_random'])
    A = g['proj_in']
    out = O.proj_so3(A)
    assert np.abs(out[:n] - g['proj_out'][:n]).max() < 2e-5
    # degenerate inputs: the projection must still be a proper rotation
    det = np.linalg.det(out.astype(np.float64))
    orth = np.abs(out @ np.swapaxes(out, -1, -2) - np.eye(3)).max()
    assert np.abs(det - 1).max() < 1e-4 and orth < 1e-4
    for i in (n + 2, n + 3, n + 5, n + 6, n + 7, n + 8):  # well-defined degenerate cases
        assert np.abs(out[i] - g['proj_out'][i]).max() < 1e-4, i
    R = O.rotvec2mat(g['rotvec_in'])
    assert np.abs(R - g['rotvec2mat_out']).max() < 1e-6
    rv = O.mat2rotvec(g['rotvec2mat_out'])
    assert np.abs(rv - g['mat2rotvec_out']).max() < 1e-5
    Ra = O.align_unit_vectors(g['align_a'], g['align_b'])
    # last 4 pairs are antiparallel: the reference's answer there is a 180-degree turn about a
    # rounding-noise axis ("arbitrary", pt/rotation.py:217) — only the defined cases are pinned
    assert np.abs(Ra[:-4] - g['align_out'][:-4]).max() < 1e-6


@pytest.mark.parametrize('name', ['smpl', 'smplx'])
def test_kid_knownpose_converter_goldens(name, model_root, golden):
    """enable_kid fits (the extra blend-shape unknown), forward with kid_factor, fit_with_known_pose
    and BodyConverter's default call (same topology: fit(enable_kid, beta_reg=0, kid_reg=1e9,
    final_adjust_rots=False, joints omitted)) against the reference's outputs."""
    g = golden(name)
    kind, md = util.load_md(model_root, name, g)
    om, of = util.make_oracle(md, kind, np.float64)
    fw = om.forward(g['pose'], g['betas'], g['trans'], kid_factor=g['kid'])
    assert np.abs(fw['vertices'] - g['kid.target_vertices']).max() < 2e-6
    assert np.abs(fw['joints'] - g['kid.fwd_joints']).max() < 2e-6
    kf = O.OracleFitter(om, enable_kid=True)
    cfgs = dict(
        a=dict(num_iter=3, beta_regularizer=1.0, use_joints=True),
        b=dict(num_iter=1, beta_regularizer=0.0, final_adjust_rots=False, kid_regularizer=1e9, use_joints=False),
        c=dict(num_iter=3, beta_regularizer=0.0, kid_regularizer=0.0, use_joints=True),
    )
    for tag, kw in cfgs.items():
        kw = dict(kw)
        uj = kw.pop('use_joints')
        o = kf.fit(g['kid.target_vertices'], g['kid.target_joints'] if uj else None, **kw)
        ref = {k: g[f'kidfit.{tag}.{k}'] for k in ('pose_rotvecs', 'shape_betas', 'trans', 'kid_factor')}
        va = om.forward(o['pose_rotvecs'], o['shape_betas'], o['trans'], kid_factor=o['kid_factor'])['vertices']
        vb = om.forward(ref['pose_rotvecs'], ref['shape_betas'], ref['trans'], kid_factor=ref['kid_factor'])['vertices']
        assert np.linalg.norm(va - vb, axis=-1).max() < 1e-4, tag
        assert np.abs(o['trans'] - ref['trans']).max() < 2e-5, tag
        if tag != 'c':  # c: betas and kid direction are nearly collinear -> only the mesh is pinned
            assert np.abs(o['shape_betas'] - ref['shape_betas']).max() < 1e-3, tag
            assert np.abs(o['kid_factor'] - ref['kid_factor']).max() < 1e-3, tag
    r = of.fit_with_known_pose(g['pose'], g['target_vertices'], g['target_joints'], beta_regularizer=1.0)
    assert np.abs(r['shape_betas'] - g['knownpose.shape_betas']).max() < 1e-4
    assert np.abs(r['trans'] - g['knownpose.trans']).max() < 1e-5
    for ni in (1, 3):  # BodyConverter.convert with identical in/out topology
        o = kf.fit(g['target_vertices'], None, num_iter=ni, beta_regularizer=0.0, final_adjust_rots=False,
                   kid_regularizer=1e9)
        ref = {k: g[f'convert.it{ni}.{k}'] for k in ('pose_rotvecs', 'shape_betas', 'trans')}
        assert util.vertex_l2(om, o, ref) < 1e-4, ni
        assert np.abs(o['trans'] - ref['trans']).max() < 2e-5, ni


@pytest.mark.parametrize('name', ['smpl', 'smplx'])
def test_known_shape_and_scale_goldens(name, model_root, golden):
    """fit_with_known_shape (pose + translation for given betas; with scale_fit, kid_factor, a warm
    start, weights, joints omitted) and fit_scale_and_translation against the reference's outputs
    (tests/golden/make_golden_ext.py).  The reference's scale branch only runs for B == 1
    (a (B,) * (B,3) broadcast, pt/bodyfitter.py:1675-1676); its per-instance results are the fixture."""
    g = golden(name)
    ge = golden(f'ext_{name}')
    kind, md = util.load_md(model_root, name, g)
    om, of = util.make_oracle(md, kind)
    for case in util.KNOWN_SHAPE_CASES:
        if f'knownshape.{case}.trans' not in ge:
            continue
        betas, tv, kw = util.known_shape_inputs(g, case)
        o = of.fit_with_known_shape(betas, tv, **kw)
        ref = {k: ge[f'knownshape.{case}.{k}'] for k in ('pose_rotvecs', 'trans', 'orientations')}
        assert np.abs(o['trans'] - ref['trans']).max() < 2e-5, case
        # single rotation entries sit at the reference's own fp32 noise floor (~3e-4;
        # SMPL-X finger / eye parts with a few dozen vertices are noisier still)
        assert np.abs(o['orientations'] - ref['orientations']).max() < (6e-4 if name == 'smpl' else 3e-3), case
        sc_o = o.get('scale_corr')
        if kw['scale_fit']:
            sc_r = ge[f'knownshape.{case}.scale_corr'].reshape(-1)
            assert np.abs(sc_o - sc_r).max() < 1e-5, case
        # the gate: meshes posed with the two results agree to 1e-4 m
        kid = kw['kid_factor']
        va = om.forward(o['pose_rotvecs'], betas, o['trans'], kid_factor=kid)['vertices']
        vb = om.forward(ref['pose_rotvecs'], betas, ref['trans'], kid_factor=kid)['vertices']
        assert np.linalg.norm(va - vb, axis=-1).max() < 1e-4, case
    if name == 'smpl':
        tv, tj, rv, rj, vw, jw = util.scale_trans_inputs(g)
        for case, (uj, uw, sc) in util.SCALE_TRANS_CASES.items():
            s, t = O.fit_scale_and_translation(tv, rv, tj if uj else None, rj if uj else None,
                                               vw if uw else None, jw if (uw and uj) else None, sc)
            assert np.abs(t - ge[f'scaletrans.{case}.trans']).max() < 1e-5, case
            if sc:
                assert np.abs(s - ge[f'scaletrans.{case}.scale']).max() < 1e-5, case


@pytest.mark.parametrize('name', ['smpl', 'smplx'])
def test_warm_start_goldens(name, model_root, golden):
    """fit with initial_pose_rotvecs / initial_shape_betas / initial_kid_factor: the first rotation pass
    runs against the model posed with the initial values, and the ridge pulls towards the initial shape
    (pt/bodyfitter.py:363-382, :1072-1081).  Case c is BodyFlipper's call (pt/bodyflipper.py:71-81)."""
    g, ge = golden(name), golden(f'ext_{name}')
    kind, md = util.load_md(model_root, name, g)
    om, of = util.make_oracle(md, kind)
    kf = O.OracleFitter(om, enable_kid=True)
    for case in util.WARM_CASES:
        if f'warm.{case}.trans' not in ge:
            continue
        kid_fit, tv, kw = util.warm_inputs(g, case)
        kw = dict(kw)
        tj = kw.pop('target_joints')
        o = (kf if kid_fit else of).fit(tv, tj, **kw)
        util.check_warm(om, name, case, o, ge, kid_fit)


@pytest.mark.parametrize('name', ['smpl', 'smplx'])
def test_share_beta_goldens(name, model_root, golden):
    """fit(share_beta=True): the regularised normal equations of all instances are summed before the
    solve (pt/lstsq.py:24-26), every instance keeps its own translation."""
    g, ge = golden(name), golden(f'ext_{name}')
    kind, md = util.load_md(model_root, name, g)
    om, of = util.make_oracle(md, kind)
    kf = O.OracleFitter(om, enable_kid=True)
    for case in util.SHARE_CASES:
        if f'share.{case}.trans' not in ge:
            continue
        kid_fit, tv, kw = util.share_inputs(g, om, case)
        assert np.abs(tv[:, ::300] - ge[f'share.{case}.target_vertices_sub']).max() < 1e-6
        kw = dict(kw)
        tj = kw.pop('target_joints')
        o = (kf if kid_fit else of).fit(tv, tj, share_beta=True, **kw)
        util.check_share(om, name, case, o, ge, kid_fit)


@pytest.mark.parametrize('name', ['smpl', 'smplx'])
def test_scale_goldens(name, model_root, golden):
    """fit(scale_target=True) / fit(scale_fit=True): one more unknown in the LAST shape solve, scaled
    targets resp. scaled reference in the refinement, scaled mean added back (pt/bodyfitter.py:434-519,
    :1170-1175, :1284-1296; the reference's TestFitterWithScale)."""
    g, ge = golden(name), golden(f'ext_{name}')
    kind, md = util.load_md(model_root, name, g)
    for dtype, loose in ((np.float64, 1.0), (np.float32, 3.0)):  # the arbiter at the gates, the fp32 form at 3 x
        om, of = util.make_oracle(md, kind, dtype)
        kf = O.OracleFitter(om, enable_kid=True)
        for case in util.SCALE_CASES:
            if f'scale.{case}.trans' not in ge:
                continue
            kid_fit, tv, kw = util.scale_inputs(g, case)
            kw = dict(kw)
            tj = kw.pop('target_joints')
            o = (kf if kid_fit else of).fit(tv, tj, **kw)
            util.check_scale(om, name, case, o, ge, kid_fit, loose=loose)
            if case in ('a', 'b'):  # the target really is a 1.1x body
                want = 1 / 1.1 if case == 'a' else 1.1
                assert np.abs(o['scale_corr'] - want).max() < 0.02


@pytest.mark.parametrize('name', ['smpl', 'smplx'])
def test_known_pose_option_goldens(name, model_root, golden):
    """fit_with_known_pose with share_beta / scale_target / scale_fit / ridge references against the
    reference's fixture (tests/golden/make_golden_knownpose.py)."""
    g, gk = golden(name), golden(f'kp_{name}')
    kind, md = util.load_md(model_root, name, g)
    om, of = util.make_oracle(md, kind, np.float64)
    fitters = {False: of, True: O.OracleFitter(om, enable_kid=True)}
    n = 0
    for case in util.KNOWN_POSE_CASES:
        if f'knownpose.{case}.trans' not in gk:
            continue
        kid_fit, pose, tv, kw = util.known_pose_inputs(g, case)
        o = fitters[kid_fit].fit_with_known_pose(pose, tv, **kw)
        util.check_known_pose(name, case, o, gk, kid_fit)
        n += 1
    assert n >= 3
    if 'sharewarm.a.trans' in gk:  # share_beta + warm start: the ridge reference is dropped (pt/lstsq.py:45-47)
        _, tv, kw = util.warm_inputs(g, 'a')
        o = util.make_oracle(md, kind)[1].fit(tv, share_beta=True, **kw)
        util.check_share(om, name, 'a', o, gk, False, prefix='sharewarm')


@pytest.mark.parametrize('name', ['smpl', 'smplx'])
def test_share_scale_goldens(name, model_root, golden):
    """fit(share_beta=True) with a scale unknown: the partially shared last solve (pt/lstsq.py:50-90)."""
    g, gk = golden(name), golden(f'kp_{name}')
    kind, md = util.load_md(model_root, name, g)
    om, of = util.make_oracle(md, kind)
    fitters = {False: of, True: O.OracleFitter(om, enable_kid=True)}
    n = 0
    for case in util.SHARE_SCALE_CASES:
        if f'sharescale.{case}.trans' not in gk:
            continue
        kid_fit, tv, kw = util.share_scale_inputs(g, om, case)
        o = fitters[kid_fit].fit(tv, share_beta=True, **kw)
        util.check_share_scale(om, name, case, o, gk, kid_fit)
        n += 1
    assert n >= 1


@pytest.mark.parametrize('kind', list(util.GENERAL_OPT_KINDS))
def test_general_option_goldens(kind, model_root, golden):
    """scale_target / scale_fit / share_beta (and both) on models of the general path against the reference
    (golden_general_opts.npz; the reference's _fit_shape_general carries them as one more column / a partially shared
    solve, pt/bodyfitter.py:1170-1175, pt/lstsq.py:50-90): the fp64 oracle at the gates of the other fixtures."""
    gg, go = golden('general'), golden('general_opts')
    g, ge = util.general_view(gg, kind), util.general_view(go, kind)
    cases = util.GENERAL_OPT_KINDS[kind]
    om = util.general_oracle(model_root, kind, np.float64)
    om32 = util.general_oracle(model_root, kind)
    fitters = {False: O.OracleFitter(om), True: O.OracleFitter(om, enable_kid=True)}
    for case in cases['scale']:
        kid_fit, tv, kw = util.scale_inputs(g, case)
        util.check_scale(om, 'smpl', case, fitters[kid_fit].fit(tv, **kw), ge, kid_fit)
    for case in cases['share']:
        kid_fit, tv, kw = util.share_inputs(g, om32, case)
        assert np.array_equal(tv[:, ::300], ge[f'share.{case}.target_vertices_sub'])
        util.check_share(om, 'smpl', case, fitters[kid_fit].fit(tv, share_beta=True, **kw), ge, kid_fit)
    for case in cases['sharescale']:
        kid_fit, tv, kw = util.share_scale_inputs(g, om32, case)
        assert np.array_equal(tv[:, ::300], ge[f'sharescale.{case}.target_vertices_sub'])
        util.check_share_scale(om, 'smpl', case, fitters[kid_fit].fit(tv, share_beta=True, **kw), ge, kid_fit)


@pytest.mark.parametrize('nb', [6, 13])
def test_num_betas_goldens(nb, model_root, golden):
    """The oracle against the reference's fixture for num_betas = 6 / 13 (golden_nb_smpl.npz)."""
    from smplfitter_amd import modelio

    gnb = golden('nb_smpl')
    md = modelio.load_model('smpl', 'neutral', model_root=f'{model_root}/{util.NB_DIR[nb]}', num_betas=nb)
    om64, _ = util.make_oracle(md, 'smpl', np.float64)
    om, _ = util.make_oracle(md, 'smpl', np.float32)
    fw = om.forward(gnb[f'nb{nb}.pose'], gnb[f'nb{nb}.betas'], gnb[f'nb{nb}.trans'])
    assert np.abs(fw['vertices'][:, ::50] - gnb[f'nb{nb}.fwd_vertices_every_50th']).max() < 2e-6
    for nb2, kid, cfg in util.NB_CASES:
        if nb2 != nb:
            continue
        of = util.O.OracleFitter(om, enable_kid=kid)
        o = of.fit(gnb[f'nb{nb}.target_vertices'], gnb[f'nb{nb}.target_joints'], **util.NB_CFG[cfg])
        util.check_nb(om64, gnb, nb, kid, cfg, o)


@pytest.mark.parametrize('kind', list(util.GENERAL_KINDS))
def test_general_goldens(kind, model_root, golden):
    """The oracle against the reference's fixture for the models of the library's general path: 32 betas, every column
    of a 300-column file (num_betas=None; the reference's _fit_shape_general), twelve skinning weights per vertex
    (golden_general.npz, tests/golden/make_golden_general.py)."""
    from smplfitter_amd import synth

    gg = golden('general')
    md = util.load_general_md(model_root, kind)
    assert synth.model_sha256(synth.make_model_arrays(kind, seed=0)) == str(gg[f'{kind}.model_sha256'])
    assert md.shapedirs.shape[2] == int(gg[f'{kind}.num_betas'])
    assert (md.weights != 0).sum(1).max() == int(gg[f'{kind}.skin_nnz'])
    om64, om = O.OracleModel(md, np.float64, 'smpl'), O.OracleModel(md, np.float32, 'smpl')
    fw = om.forward(gg[f'{kind}.pose'], gg[f'{kind}.betas'], gg[f'{kind}.trans'])
    assert np.abs(fw['vertices'] - gg[f'{kind}.target_vertices']).max() < 3e-6
    assert np.abs(fw['joints'] - gg[f'{kind}.target_joints']).max() < 3e-6
    if f'{kind}.fwd10_joints' in gg:  # fewer betas given than the model has
        fw10 = om.forward(gg[f'{kind}.pose'], gg[f'{kind}.betas'][:, :10], gg[f'{kind}.trans'])
        assert np.abs(fw10['vertices'][:, ::50] - gg[f'{kind}.fwd10_vertices_every_50th']).max() < 3e-6
    for case, c in util.GENERAL_CASES.items():
        tv, kw = util.general_fit_args(gg, kind, case)
        o = O.OracleFitter(om, enable_kid=c.get('kid', False)).fit(tv, **kw)
        util.check_general(om64, gg, kind, case, o)


@pytest.mark.parametrize('tag', ['s2x', 'x2s'])
def test_convert_cross_topology_goldens(tag, model_root, golden, data_root_fat):
    """BodyConverter between the two topologies (reference pt/bodyconverter.py:22-149 run on the synthetic
    transfer files): oracle forward of the input model -> the CSR product -> oracle fit of the output model with the
    kid unknown, against the reference's outputs (every branch of convert)."""
    gc = golden('convert')
    a, b = util.CONVERT_DIRS[tag]
    csr = util.load_transfer_csr(data_root_fat, tag)
    assert util.csr_digest(csr) == str(gc[f'{tag}.csr_sha256']), 'the synthetic transfer matrix differs from the fixture\'s'
    _, md_in = util.load_md(model_root, a)
    _, md_out = util.load_md(model_root, b)
    ka, kb = a[:5] if a.startswith('smplx') else a, b[:5] if b.startswith('smplx') else b  # model family names
    om_in = O.OracleModel(md_in, np.float32, ka)
    om_out32, om_out = O.OracleModel(md_out, np.float32, kb), O.OracleModel(md_out, np.float64, kb)
    pose, betas, trans, kid = (gc[f'{tag}.{k}'] for k in ('pose', 'betas', 'trans', 'kid'))
    vin = om_in.forward(pose, betas, trans)['vertices'].astype(np.float32)
    verts = np.stack([csr @ v for v in vin]).astype(np.float32)
    assert np.abs(verts[:, ::97] - gc[f'{tag}.vertices_sub']).max() < 2e-6
    kf = O.OracleFitter(om_out32, enable_kid=True)
    for ni in (1, 3):
        o = kf.fit(verts, None, num_iter=ni, beta_regularizer=0.0, final_adjust_rots=False, kid_regularizer=1e9)
        util.check_convert(om_out, tag, f'it{ni}', {k: o[k] for k in ('pose_rotvecs', 'shape_betas', 'trans')}, gc)
    o = kf.fit(verts, None, num_iter=1, beta_regularizer=0.0, final_adjust_rots=False, kid_regularizer=0.0)
    util.check_convert(om_out, tag, 'kid.it1', {k: o[k] for k in ('pose_rotvecs', 'shape_betas', 'trans', 'kid_factor')}, gc)
    o = kf.fit_with_known_shape(gc[f'{tag}.kshape.betas_in'], verts, None, num_iter=2, final_adjust_rots=False)
    util.check_convert(om_out, tag, 'kshape', {k: o[k] for k in ('pose_rotvecs', 'trans')}, gc)
    o = kf.fit_with_known_pose(gc[f'{tag}.kpose.pose_in'], verts, None, beta_regularizer=0.0, kid_regularizer=1e9)
    util.check_convert(om_out, tag, 'kpose', {k: o[k] for k in ('shape_betas', 'trans')}, gc)
