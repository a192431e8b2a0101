"""Deterministic SMPL-*shaped* synthetic body-model files.

The licensed SMPL / SMPL-X model files cannot ship with this repository, so parity tests, the
benchmark and the golden-vector generator all work on a seeded synthetic model that has exactly
the on-disk layout the reference loader reads (reference: src/smplfitter/common.py:302-331 — keys
``v_template, shapedirs, posedirs, J_regressor, weights, f, kintree_table`` plus
``kid_template.npy``).  Real model files drop into the same loader unchanged.

Design of the fixture: noisy *fat* capsules around the public SMPL /
SMPL-X kinematic trees (radial spread ~0.8x radius so the bone-twist solve is well conditioned),
top-4 skinning weights, local Gaussian joint regressor, small random shape / pose blend shapes,
and a *shuffled* vertex order (real SMPL vertices are not sorted by body part, so the part-sorting
gather in the first kernel is exercised realistically).

Everything is driven by ``numpy.random.RandomState(seed)`` (the legacy, version-stable stream) so the
GPU box regenerates bit-identical arrays; ``model_sha256`` lets tests verify that.
"""

from __future__ import annotations

import hashlib
import os
import re
import os.path as osp
import pickle

import numpy as np

SMPL_PARENTS = [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21]

# SMPL-X: 22 body joints, jaw + 2 eyes on the head, 5 three-joint finger chains per wrist.
SMPLX_PARENTS = (
    SMPL_PARENTS[:22]
    + [15, 15, 15]
    + [20, 25, 26, 20, 28, 29, 20, 31, 32, 20, 34, 35, 20, 37, 38]
    + [21, 40, 41, 21, 43, 44, 21, 46, 47, 21, 49, 50, 21, 52, 53]
)

# A plausible T-pose skeleton (metres, y up, +x = body's left).  Not SMPL's actual numbers.
_BODY_JOINTS = np.array(
    [
        [0.00, 0.00, 0.00],  # 0 pelvis
        [0.07, -0.09, 0.00],  # 1 l_hip
        [-0.07, -0.09, 0.00],  # 2 r_hip
        [0.00, 0.11, -0.02],  # 3 spine1
        [0.10, -0.47, 0.00],  # 4 l_knee
        [-0.10, -0.47, 0.00],  # 5 r_knee
        [0.00, 0.25, 0.00],  # 6 spine2
        [0.09, -0.87, -0.03],  # 7 l_ankle
        [-0.09, -0.87, -0.03],  # 8 r_ankle
        [0.00, 0.30, 0.02],  # 9 spine3
        [0.11, -0.93, 0.09],  # 10 l_foot
        [-0.11, -0.93, 0.09],  # 11 r_foot
        [0.00, 0.52, -0.03],  # 12 neck
        [0.08, 0.42, -0.02],  # 13 l_collar
        [-0.08, 0.42, -0.02],  # 14 r_collar
        [0.00, 0.60, 0.02],  # 15 head
        [0.18, 0.46, -0.03],  # 16 l_shoulder
        [-0.18, 0.46, -0.03],  # 17 r_shoulder
        [0.44, 0.45, -0.05],  # 18 l_elbow
        [-0.44, 0.45, -0.05],  # 19 r_elbow
        [0.69, 0.46, -0.05],  # 20 l_wrist
        [-0.69, 0.46, -0.05],  # 21 r_wrist
        [0.78, 0.45, -0.06],  # 22 l_hand
        [-0.78, 0.45, -0.06],  # 23 r_hand
    ],
    dtype=np.float64,
)


def _smplx_joints():
    j = np.zeros((55, 3), np.float64)
    j[:22] = _BODY_JOINTS[:22]
    j[22] = j[15] + [0.0, -0.03, 0.05]  # jaw
    j[23] = j[15] + [0.03, 0.06, 0.08]  # left eye
    j[24] = j[15] + [-0.03, 0.06, 0.08]  # right eye
    # finger chains fan out from each wrist; order index, middle, pinky, ring, thumb
    fan = [(-0.02, 0.03), (-0.025, 0.01), (-0.03, -0.03), (-0.028, -0.01), (0.0, 0.045)]
    for side, (wrist, base) in enumerate([(20, 25), (21, 40)]):
        sx = 1.0 if side == 0 else -1.0
        for f, (dy, dz) in enumerate(fan):
            first_len = 0.09 if f < 4 else 0.04
            p = j[wrist] + np.array([sx * first_len, dy, dz])
            step = np.array([sx * 0.03, -0.004, 0.002 * (f - 2)])
            if f == 4:  # thumb points forward-outward
                step = np.array([sx * 0.02, -0.005, 0.022])
            for k in range(3):
                j[base + 3 * f + k] = p + k * step
    return j


def _segments(joints, parents):
    """Per joint: segment from the joint to the mean of its children (leaf: extended from parent)."""
    J = len(parents)
    children = [[] for _ in range(J)]
    for i in range(1, J):
        children[parents[i]].append(i)
    seg = np.zeros((J, 2, 3))
    for i in range(J):
        a = joints[i]
        if children[i]:
            b = joints[children[i]].mean(axis=0)
        else:
            d = joints[i] - joints[parents[i]]
            b = a + 0.6 * d
        if np.linalg.norm(b - a) < 0.02:  # keep every segment non-degenerate
            d = b - a
            n = np.linalg.norm(d)
            d = d / n if n > 1e-9 else np.array([0.0, 1.0, 0.0])
            b = a + 0.02 * d
        seg[i] = (a, b)
    return seg, children


def _dist_to_segments(pts, seg):
    """(V,3) x (J,2,3) -> (V,J) distance from every point to every segment."""
    a = seg[:, 0][None]  # (1,J,3)
    d = (seg[:, 1] - seg[:, 0])[None]
    len2 = (d * d).sum(-1)
    t = ((pts[:, None] - a) * d).sum(-1) / len2
    t = np.clip(t, 0.0, 1.0)
    closest = a + t[..., None] * d
    return np.linalg.norm(pts[:, None] - closest, axis=-1)


def make_model_arrays(kind='smpl', seed=0, num_vertices=None, num_betas=10, shuffle=True):
    """Build the raw arrays of a synthetic SMPL-shaped ('smpl', 24 joints) or SMPL-X-shaped
    ('smplx', 55 joints) model.  Returns a dict with the reference's on-disk keys.

    'smplx_fat' is the SMPL-X-shaped model with FAT fingers / face parts (3.5 cm instead of 1.2 cm): the
    bone parts' twist about the bone axis is recovered from the vertices' off-axis spread
    (pt/bodyfitter.py:1398-1410), so thin parts make pose_rotvecs ill-conditioned in the reference itself;
    on the fat variant a tight pose_rotvecs comparison is meaningful."""
    m = re.search(r'_b(\d+)$', kind)
    if m:  # the same construction with that many shape directions ('smpl_b16'; 'smpl_b32', 'smpl_b300': the general path)
        kind, num_betas = kind[:m.start()], int(m.group(1))
    # skinning variants: '_w6' keeps the SIX largest weights of a vertex (real models are not capped at four: the
    # reference blends with the dense (V, J) matrix, pt/bodyfitter.py:1000-1003); '_rnd' gives every vertex its own
    # part (weight 0.75) and THREE RANDOM other joints — joint sets the distance-based construction never produces
    # (the vertex-piece / cell tables of the batch-major kernels must cope with any of them)
    skin_nnz, skin_random = 4, False
    m = re.search(r'_w(\d+)$', kind)
    if m:  # ('_w6': eight pairs per vertex; '_w12': more than eight — the general path)
        kind, skin_nnz = kind[:m.start()], int(m.group(1))
    if kind.endswith('_rnd'):
        kind, skin_random = kind[:-4], True
    fat = kind.endswith('_fat')
    kind = kind[:-4] if fat else kind
    rs = np.random.RandomState(seed)
    if kind == 'smpl':
        parents = list(SMPL_PARENTS)
        joints = _BODY_JOINTS.copy()
        V = 6890 if num_vertices is None else num_vertices
    elif kind == 'smplx':
        parents = list(SMPLX_PARENTS)
        joints = _smplx_joints()
        V = 10475 if num_vertices is None else num_vertices
    else:
        raise ValueError(f'unknown synthetic model kind {kind!r}')
    J = len(parents)
    seg, children = _segments(joints, parents)

    # part radii: torso fat, limbs medium, fingers / face thin
    radius = np.full(J, 0.055)
    for i in (0, 3, 6, 9):
        radius[i] = 0.10
    for i in (12, 15):
        radius[i] = 0.07
    for i in (1, 2):
        radius[i] = 0.075
    for i in (18, 19, 20, 21, 7, 8, 10, 11):
        radius[i] = 0.04
    if kind == 'smpl':
        radius[22:] = 0.035
    else:
        radius[22:] = 0.035 if fat else 0.012

    # vertices: ~V/J per part, a few more on big parts
    share = radius ** 0.5
    counts = np.floor(share / share.sum() * V).astype(int)
    counts[: V - counts.sum()] += 1
    assert counts.sum() == V and counts.min() >= 64
    verts = []
    gen_part = []
    for i in range(J):
        n = counts[i]
        t = rs.uniform(-0.15, 1.15, size=(n, 1))
        base = seg[i, 0] + t * (seg[i, 1] - seg[i, 0])
        # fat capsule: isotropic gaussian radial noise with std 0.8 x radius
        noise = rs.randn(n, 3) * (0.8 * radius[i])
        verts.append(base + noise)
        gen_part.append(np.full(n, i))
    v_template = np.concatenate(verts, axis=0)
    gen_part = np.concatenate(gen_part)

    if shuffle:
        order = rs.permutation(V)
        v_template = v_template[order]
        gen_part = gen_part[order]

    # skinning weights: exp(-dist/sigma) to each part's segment, boost own part, top-4
    sigma = 0.03 if kind == 'smpl' else 0.015
    dist = _dist_to_segments(v_template, seg)
    dist[np.arange(V), gen_part] *= 0.25  # the generating part dominates
    logits = -dist / sigma
    logits -= logits.max(axis=1, keepdims=True)
    w = np.exp(logits)
    top = np.argsort(-w, axis=1, kind='stable')[:, :skin_nnz]
    keep = np.zeros_like(w, dtype=bool)
    keep[np.arange(V)[:, None], top] = True
    w = np.where(keep, w, 0.0)
    w /= w.sum(axis=1, keepdims=True)
    if skin_random:
        rs2 = np.random.RandomState(seed + 77)
        w = np.zeros_like(w)
        w[np.arange(V), gen_part] = 0.75
        for v in range(V):
            others = rs2.choice([j for j in range(J) if j != gen_part[v]], size=3, replace=False)
            w[v, others] = 0.25 * rs2.dirichlet(np.ones(3))
    weights = w.astype(np.float64)

    # joint regressor: row-normalised gaussian over the 60 nearest vertices of each joint
    J_regressor = np.zeros((J, V))
    for i in range(J):
        d = np.linalg.norm(v_template - joints[i], axis=1)
        nn = np.argsort(d)[:60]
        g = np.exp(-0.5 * (d[nn] / 0.05) ** 2) + 1e-3
        J_regressor[i, nn] = g / g.sum()

    S_file = max(num_betas, 10)
    shapedirs = rs.randn(V, 3, S_file) * 0.004 + v_template[:, :, None] * (
        rs.randn(1, 1, S_file) * 0.03
    )
    posedirs = rs.randn(V, 3, (J - 1) * 9) * 0.002
    faces = np.stack([np.arange(V - 2), np.arange(1, V - 1), np.arange(2, V)], axis=1).astype(
        np.uint32
    )
    kintree_table = np.stack(
        [np.array([-1] + parents[1:], dtype=np.int64), np.arange(J, dtype=np.int64)]
    )
    kid_template = 0.6 * v_template
    return dict(
        v_template=v_template,
        shapedirs=shapedirs,
        posedirs=posedirs,
        J_regressor=J_regressor,
        weights=weights,
        f=faces,
        kintree_table=kintree_table,
        kid_template=kid_template,
    )


def write_model_files(root, kind='smpl', seed=0, num_vertices=None):
    """Write ``<root>/<kind>/<official file name>`` (+ ``kid_template.npy``); returns the dir.

    File names follow the reference loader (src/smplfitter/common.py:266-283)."""
    arrs = make_model_arrays(kind, seed, num_vertices)
    kid = arrs.pop('kid_template')
    d = osp.join(root, kind)  # 'smplx_fat' / 'smpl_b16' live in their own directory, same official file name
    os.makedirs(d, exist_ok=True)
    if not kind.startswith('smplx'):
        path = osp.join(d, 'basicmodel_neutral_lbs_10_207_0_v1.1.0.pkl')
        tmp = path + f'.tmp{os.getpid()}'
        with open(tmp, 'wb') as f:
            pickle.dump(arrs, f, protocol=2)
        os.replace(tmp, path)
    else:
        path = osp.join(d, 'SMPLX_NEUTRAL.npz')
        tmp = path + f'.tmp{os.getpid()}.npz'
        np.savez(tmp, **arrs)
        os.replace(tmp, path)
    kid_path = osp.join(d, 'kid_template.npy')
    tmpk = kid_path + f'.tmp{os.getpid()}.npy'
    np.save(tmpk, kid)
    os.replace(tmpk, kid_path)
    return d


def ensure_model_root(root=None, kinds=('smpl',), seed=0):
    """Create the synthetic model files under ``root`` if missing; returns ``root``.

    Default root: ``$SMPLFIT_SYNTH_ROOT`` or ``/tmp/smplfit_synth_models_seed<seed>``."""
    if root is None:
        root = os.getenv('SMPLFIT_SYNTH_ROOT', f'/tmp/smplfit_synth_models_seed{seed}')
    for kind in kinds:
        fn = (
            'SMPLX_NEUTRAL.npz' if kind.startswith('smplx') else 'basicmodel_neutral_lbs_10_207_0_v1.1.0.pkl'
        )
        if not (
            osp.exists(osp.join(root, kind, fn))
            and osp.exists(osp.join(root, kind, 'kid_template.npy'))
        ):
            write_model_files(root, kind, seed)
    return root


def model_sha256(arrs) -> str:
    """Order-independent digest of the float64 arrays (to check the box regenerated the same model)."""
    h = hashlib.sha256()
    for k in sorted(arrs):
        a = np.ascontiguousarray(arrs[k])
        h.update(k.encode())
        h.update(str(a.dtype).encode())
        h.update(str(a.shape).encode())
        h.update(a.tobytes())
    return h.hexdigest()


def subset_indices(num_vertices, part_assignment, size=1024, min_per_part=8, seed=1):
    """A seeded vertex subset with at least ``min_per_part`` vertices of every body part
    (BASELINE.json config 4: 'SMPL-neutral 1024-vertex subset')."""
    rs = np.random.RandomState(seed)
    part_assignment = np.asarray(part_assignment)
    chosen = []
    for p in np.unique(part_assignment):
        idx = np.where(part_assignment == p)[0]
        take = min(min_per_part, len(idx))
        chosen.extend(rs.choice(idx, size=take, replace=False).tolist())
    chosen = set(chosen)
    rest = np.array([i for i in range(num_vertices) if i not in chosen])
    extra = rs.choice(rest, size=size - len(chosen), replace=False)
    return np.sort(np.concatenate([np.array(sorted(chosen)), extra])).astype(np.int64)


def _nearest3(src_pts, query_pts):
    """For each query point: the 3 nearest source points and inverse-distance weights (sum 1)."""
    from scipy.spatial import cKDTree

    d, idx = cKDTree(src_pts).query(query_pts, k=3)
    w = 1.0 / (d + 1e-4)
    w /= w.sum(1, keepdims=True)
    return idx.astype(np.int64), w.astype(np.float32)


def write_transfer_files(data_root, seed=0, smplx_kind='smplx'):
    """Synthetic stand-ins, in the official file layouts, for the topology-transfer and mirror files
    ``BodyConverter`` / ``BodyFlipper`` read under ``$DATA_ROOT/body_models`` (reference
    common.py:425-429, pt/bodyflipper.py:136-169):

    * ``smpl2smplx_deftrafo_setup.pkl`` / ``smplx2smpl_deftrafo_setup.pkl``: ``{'mtx': (V_out, 2 V_in)}``
      scipy CSR whose first V_in columns are a barycentric transfer (3 nearest template vertices);
    * ``smplx/smplx_flip_correspondences.npz``: ``closest_faces (V,3)``, ``bc (V,3)`` — for every SMPL-X
      vertex the 3 template vertices nearest to its mirror image and their weights.
    Returns ``data_root``."""
    import scipy.sparse as sp

    smpl = make_model_arrays('smpl', seed)['v_template'].astype(np.float64)
    smplx = make_model_arrays(smplx_kind, seed)['v_template'].astype(np.float64)  # 'smplx' or its fat-part variant
    d = osp.join(data_root, 'body_models')
    os.makedirs(osp.join(d, 'smplx'), exist_ok=True)

    def dump(name, src, dst):
        idx, w = _nearest3(src, dst)
        rows = np.repeat(np.arange(len(dst)), 3)
        m = sp.csr_matrix((w.reshape(-1), (rows, idx.reshape(-1))), shape=(len(dst), len(src)))
        tmp = osp.join(d, name + f'.tmp{os.getpid()}')
        with open(tmp, 'wb') as f:
            pickle.dump(dict(mtx=sp.hstack([m, m]).tocsr()), f, protocol=2)
        os.replace(tmp, osp.join(d, name))

    dump('smpl2smplx_deftrafo_setup.pkl', smpl, smplx)
    dump('smplx2smpl_deftrafo_setup.pkl', smplx, smpl)
    idx, w = _nearest3(smplx, smplx * np.array([-1.0, 1.0, 1.0]))
    tmp = osp.join(d, 'smplx', f'flip.tmp{os.getpid()}.npz')
    np.savez(tmp, closest_faces=idx, bc=w)
    os.replace(tmp, osp.join(d, 'smplx', 'smplx_flip_correspondences.npz'))
    return data_root
