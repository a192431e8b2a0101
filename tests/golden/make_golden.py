"""Generate the golden vectors in this directory by running the REFERENCE itself.

Runs only in the build container (imports ``/root/reference/src``; the reference cannot travel to
the GPU box).  It feeds the reference's PyTorch backend (CPU, fp32) the seeded synthetic model files
of ``smplfitter_amd.synth`` and records inputs + outputs as small ``.npz`` fixtures:

* ``golden_<kind>.npz`` (kind = smpl, smplx, smpl1024):
  - inputs: ``pose, betas, trans`` (B=4), the exact ``target_vertices / target_joints`` the reference's
    forward produced from them (fit outputs are sensitive to 1-ulp input changes, so targets are
    stored, not regenerated), ``vertex_weights, joint_weights``;
  - forward pins: ``fwd_vertices_sub`` = vertices[:, ::300] and ``fwd_joints`` (same sampling as the
    reference's only known-answer test, tests/test_forward.py:126-127);
  - ``fit.<config>.<key>`` for a grid of ``fit`` options (num_iter, beta_regularizer, joints
    given/None, weights given/None, final_adjust_rots);
  - stage pins for the default config: ``stage.glob_rotmats_iter0``, ``stage.part_sums.*``,
    ``stage.gram_cen0 / rhs_cen0 / shape_betas0 / trans0``;
  - ``model_sha256`` of the generated model arrays.
* ``golden_primitives.npz``: ``proj_SO3`` on random + degenerate 3x3s, ``mat2rotvec`` over all four
  branches and angles near 0 / pi, ``align_unit_vectors`` at (anti)parallel, ``rotvec2mat`` at 0.

Usage:  python tests/golden/make_golden.py
"""

import itertools
import os
import os.path as osp
import sys

import numpy as np
import torch

HERE = osp.dirname(osp.abspath(__file__))
sys.path.insert(0, osp.join(HERE, '..', '..'))
sys.path.insert(0, '/root/reference/src')

import smplfitter.pt as ref  # noqa: E402
from smplfitter.pt import rotation as ref_rot  # noqa: E402
from smplfitter_amd import synth  # noqa: E402

B = 8


def cfg_name(num_iter, beta_reg, joints, weights, final):
    return f'it{num_iter}_reg{int(beta_reg)}_{"j" if joints else "nj"}_{"w" if weights else "nw"}_{"fa" if final else "nfa"}'


def make_kind(kind, root, subset=None):
    name = kind if subset is None else f'{kind}{len(subset)}'
    arrs = synth.make_model_arrays(kind, seed=0)
    kw = {}
    if subset is not None:
        kw = dict(vertex_subset=subset)
    model = ref.BodyModel(kind, 'neutral', model_root=f'{root}/{kind}', num_betas=10, **kw)
    fitter = ref.BodyFitter(model)
    J = model.num_joints
    rs = np.random.RandomState(1234)
    pose = (rs.randn(B, 3 * J) * 0.1).astype(np.float32)
    betas = (rs.randn(B, 10) * 0.5).astype(np.float32)
    trans = rs.randn(B, 3).astype(np.float32)
    vw = rs.uniform(0.5, 1.5, size=(B, model.num_vertices)).astype(np.float32)
    jw = rs.uniform(0.5, 1.5, size=(B, J)).astype(np.float32)
    out = dict(pose=pose, betas=betas, trans=trans, vertex_weights=vw, joint_weights=jw)
    out['model_sha256'] = np.array(synth.model_sha256(arrs))
    if subset is not None:
        out['vertex_subset'] = np.asarray(subset)
    with torch.no_grad():
        fw = model(torch.from_numpy(pose), torch.from_numpy(betas), torch.from_numpy(trans))
        tv, tj = fw['vertices'], fw['joints']
        out['target_vertices'] = tv.numpy()
        out['target_joints'] = tj.numpy()
        out['fwd_vertices_sub'] = tv.numpy()[:, ::300]
        out['fwd_joints'] = tj.numpy()
        out['fwd_orientations'] = fw['orientations'].numpy()

        if subset is None and kind == 'smpl':
            grid = list(itertools.product([1, 3], [0.0, 1.0], [True, False], [False, True], [True, False]))
        else:
            grid = [(3, 1.0, True, False, True), (1, 0.0, True, False, False), (3, 1.0, True, True, True)]
            if subset is None:
                grid += [(2, 1.0, False, False, True), (3, 0.0, False, True, False)]
        for num_iter, reg, joints, weights, final in grid:
            r = fitter.fit(
                tv,
                tj if joints else None,
                vertex_weights=torch.from_numpy(vw) if weights else None,
                joint_weights=torch.from_numpy(jw) if (weights and joints) else None,
                num_iter=num_iter,
                beta_regularizer=reg,
                final_adjust_rots=final,
                requested_keys=['pose_rotvecs', 'shape_betas', 'trans'],
            )
            c = cfg_name(num_iter, reg, joints, weights, final)
            for k in ('pose_rotvecs', 'shape_betas', 'trans', 'orientations'):
                out[f'fit.{c}.{k}'] = r[k].numpy()

        # stage pins for the default configuration (joints given, no weights)
        mean = torch.cat([tv, tj], 1).mean(1)
        tvc, tjc = tv - mean[:, None], tj - mean[:, None]
        raw, s_t, s_a, s_w = fitter._part_sums(tvc, fitter.default_mesh_tf[None], None)
        out['stage.part_sums.raw'] = raw.numpy()
        out['stage.part_sums.s_t'] = s_t.numpy()
        out['stage.part_sums.s_a'] = s_a.numpy()
        out['stage.part_sums.s_w'] = s_w.numpy()
        G0 = fitter._fit_global_rotations(
            tvc, tjc, fitter.default_mesh_tf[None], model.J_template[None], None, None
        )
        out['stage.glob_rotmats_iter0'] = G0.numpy()
        sh = fitter._fit_shape(G0, tvc, tjc, None, None, 1.0, 0.0, requested_keys=['vertices', 'joints'])
        out['stage.shape_betas0'] = sh['shape_betas'].numpy()
        out['stage.trans0'] = sh['trans'].numpy()
        out['stage.vertices0_sub'] = sh['vertices'].numpy()[:, ::300]
        out['stage.joints0'] = sh['joints'].numpy()
        if subset is None:
            # kid blend shape (enable_kid): forward with kid_factor, fits with the extra unknown, the
            # known-pose solve, and BodyConverter's default call (same-topology models -> no CSR):
            # fit(enable_kid, beta_regularizer=0, final_adjust_rots=False, kid_regularizer=1e9)
            kid = (rs.randn(B) * 0.3).astype(np.float32)
            out['kid'] = kid
            fwk = model(torch.from_numpy(pose), torch.from_numpy(betas), torch.from_numpy(trans),
                        kid_factor=torch.from_numpy(kid))
            out['kid.fwd_vertices_sub'] = fwk['vertices'].numpy()[:, ::300]
            out['kid.fwd_joints'] = fwk['joints'].numpy()
            out['kid.target_vertices'] = fwk['vertices'].numpy()
            out['kid.target_joints'] = fwk['joints'].numpy()
            kfitter = ref.BodyFitter(model, enable_kid=True)
            for tag, kw in (
                ('a', dict(num_iter=3, beta_regularizer=1.0, use_joints=True)),
                ('b', dict(num_iter=1, beta_regularizer=0.0, final_adjust_rots=False, kid_regularizer=1e9, use_joints=False)),
                ('c', dict(num_iter=3, beta_regularizer=0.0, kid_regularizer=0.0, use_joints=True)),
            ):
                kw = dict(kw)
                uj = kw.pop('use_joints')
                r = kfitter.fit(fwk['vertices'], fwk['joints'] if uj else None,
                                requested_keys=['pose_rotvecs', 'shape_betas', 'trans'], **kw)
                for k in ('pose_rotvecs', 'shape_betas', 'trans', 'kid_factor', 'orientations'):
                    out[f'kidfit.{tag}.{k}'] = r[k].numpy()
            r = fitter.fit_with_known_pose(torch.from_numpy(pose), tv, tj, beta_regularizer=1.0)
            out['knownpose.shape_betas'] = r['shape_betas'].numpy()
            out['knownpose.trans'] = r['trans'].numpy()
            conv = ref.BodyConverter(model, model)
            for ni in (1, 3):
                r = conv.convert(torch.from_numpy(pose), torch.from_numpy(betas), torch.from_numpy(trans), num_iter=ni)
                for k in ('pose_rotvecs', 'shape_betas', 'trans'):
                    out[f'convert.it{ni}.{k}'] = r[k].numpy()
    path = osp.join(HERE, f'golden_{name}.npz')
    np.savez_compressed(path, **out)
    print(path, f'{os.path.getsize(path) / 1e6:.2f} MB', len(out), 'arrays')
    return model


def make_primitives():
    rs = np.random.RandomState(7)
    out = {}
    A = rs.randn(256, 3, 3).astype(np.float32)
    deg = []
    u = rs.randn(3).astype(np.float32)
    v = rs.randn(3).astype(np.float32)
    deg.append(np.outer(u, v))  # rank 1
    deg.append(np.outer(u, v) + np.outer(rs.randn(3), rs.randn(3)).astype(np.float32))  # rank 2
    deg.append(np.diag([1.0, 1.0, -1.0]).astype(np.float32))  # reflection
    deg.append(-np.eye(3, dtype=np.float32))  # reflection, all negative
    deg.append(np.zeros((3, 3), np.float32))  # zero
    deg.append(np.eye(3, dtype=np.float32) * 1e-6)  # tiny scale
    deg.append((np.eye(3) * 1e6).astype(np.float32))  # huge scale
    q, _ = np.linalg.qr(rs.randn(3, 3))
    deg.append((q @ np.diag([3.0, 2.0, 1e-4]) @ q.T).astype(np.float32))  # nearly rank 2
    deg.append((q @ np.diag([2.0, 1.0, -0.5])).astype(np.float32))  # negative determinant
    A = np.concatenate([A, np.stack(deg).astype(np.float32)], 0)
    out['proj_in'] = A
    out['proj_out'] = ref_rot.proj_SO3(torch.from_numpy(A)).numpy()
    out['proj_n_random'] = np.array(256)

    # rotations covering all mat2rotvec branches, incl. angles near 0 and pi
    axes = rs.randn(64, 3)
    axes /= np.linalg.norm(axes, axis=1, keepdims=True)
    axes = np.concatenate([axes, np.eye(3), -np.eye(3)], 0)
    angles = np.array([0.0, 1e-7, 1e-4, 0.3, 1.5, 2.5, 3.0, np.pi - 1e-3, np.pi - 1e-6, np.pi])
    rv = (axes[:, None, :] * angles[None, :, None]).reshape(-1, 3).astype(np.float32)
    out['rotvec_in'] = rv
    R = ref_rot.rotvec2mat(torch.from_numpy(rv))
    out['rotvec2mat_out'] = R.numpy()
    out['mat2rotvec_out'] = ref_rot.mat2rotvec(R).numpy()

    a = rs.randn(32, 3).astype(np.float32)
    a /= np.linalg.norm(a, axis=1, keepdims=True)
    b = rs.randn(32, 3).astype(np.float32)
    b /= np.linalg.norm(b, axis=1, keepdims=True)
    a = np.concatenate([a, a[:4], a[:4]], 0)
    b = np.concatenate([b, a[:4], -a[:4]], 0)  # parallel and antiparallel pairs
    out['align_a'], out['align_b'] = a, b
    out['align_out'] = ref_rot.align_unit_vectors(torch.from_numpy(a), torch.from_numpy(b)).numpy()
    path = osp.join(HERE, 'golden_primitives.npz')
    np.savez_compressed(path, **out)
    print(path, f'{os.path.getsize(path) / 1e6:.2f} MB')


def main():
    torch.set_num_threads(8)
    root = synth.ensure_model_root(kinds=('smpl', 'smplx'), seed=0)
    m = make_kind('smpl', root)
    make_kind('smplx', root)
    fitter = ref.BodyFitter(m)
    subset = synth.subset_indices(m.num_vertices, fitter.part_assignment.numpy(), 1024, 8, seed=1)
    make_kind('smpl', root, subset=subset)
    make_primitives()


if __name__ == '__main__':
    main()
