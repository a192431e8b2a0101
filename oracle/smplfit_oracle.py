"""CPU ORACLE — test infrastructure only.  NOT part of the product path.

A plain-numpy restatement of the default-configuration algorithm of the reference's
``smplfitter.pt.BodyFitter.fit`` and ``BodyModel.forward`` (reference files cited per function as
``file:line`` relative to ``/root/reference/src/smplfitter``).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module, and
only as the checker / the timed CPU baseline; the shipped package never does.

Parity pinning: the reference has no golden vectors for ``fit`` (its tests are round-trip
property tests, tests/test_fitter_common.py:31-72).  This oracle is therefore pinned against
outputs of the reference itself, captured in the build container by ``tests/golden/make_golden.py``
(imports ``/root/reference/src``) and committed as ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks every one of them.

Precision model: everything in ``dtype`` (float32 = the reference's arithmetic;
float64 = the arbiter when two fp32 evaluations disagree) except the 10x10 normal-equation
combination / Cholesky solve, which is float64 in both (pt/bodyfitter.py:1034-1089).
Written for clarity (per-part Python loops over <= 55 parts, batched over B), not speed.
"""

from __future__ import annotations

import numpy as np


# ----------------------------------------------------------------------------------------------
# rotation primitives (pt/rotation.py)
# ----------------------------------------------------------------------------------------------
def divide_no_nan(a, b):
    """a / b, 0 where b == 0 (pt/rotation.py:8-11)."""
    safe = np.where(b == 0, np.ones_like(b), b)
    return np.where(b == 0, np.zeros_like(a / safe), a / safe)


def proj_so3(A):
    """Nearest rotation by SVD with the reflection fix on the last singular pair
    (pt/rotation.py:100-110)."""
    U, _, Vh = np.linalg.svd(A)
    T = U @ Vh
    neg = np.linalg.det(T) < 0
    mirror = T - 2 * U[..., :, -1:] @ Vh[..., -1:, :]
    return np.where(neg[..., None, None], mirror, T).astype(A.dtype)


def rotvec2mat(r):
    """Rodrigues, element order as pt/rotation.py:236-258."""
    angle = np.linalg.norm(r, axis=-1, keepdims=True)
    axis = divide_no_nan(r, angle)
    s = np.sin(angle) * axis
    c = np.cos(angle)
    c1 = (1.0 - c).astype(r.dtype) * axis
    ax, ay, az = axis[..., 0], axis[..., 1], axis[..., 2]
    c1x, c1y = c1[..., 0], c1[..., 1]
    sx, sy, sz = s[..., 0], s[..., 1], s[..., 2]
    t = c1x * ay
    m01, m10 = t - sz, t + sz
    t = c1x * az
    m02, m20 = t + sy, t - sy
    t = c1y * az
    m12, m21 = t - sx, t + sx
    diag = c1 * axis + c
    m = np.stack(
        [diag[..., 0], m01, m02, m10, diag[..., 1], m12, m20, m21, diag[..., 2]], axis=-1
    )
    return m.reshape(r.shape[:-1] + (3, 3)).astype(r.dtype)


def mat2rotvec(R):
    """Quaternion by the 4-way branch, then 2*atan2(|xyz|, w)/|xyz| * xyz; w may be negative
    (rotvec norm > pi) — kept as is (pt/rotation.py:261-289)."""
    dt = R.dtype
    one = dt.type(1.0)
    r00, r01, r02 = R[..., 0, 0], R[..., 0, 1], R[..., 0, 2]
    r10, r11, r12 = R[..., 1, 0], R[..., 1, 1], R[..., 1, 2]
    r20, r21, r22 = R[..., 2, 0], R[..., 2, 1], R[..., 2, 2]
    trace = r00 + r11 + r22
    c0 = np.stack([r21 - r12, r02 - r20, r10 - r01, one + trace], -1)
    c1 = np.stack([(one - r22) + (r00 - r11), r10 + r01, r02 + r20, r21 - r12], -1)
    c2 = np.stack([r10 + r01, (one - r22) - (r00 - r11), r21 + r12, r02 - r20], -1)
    c3 = np.stack([r02 + r20, r21 + r12, (one + r22) - (r00 + r11), r10 - r01], -1)
    tp = (trace > 0)[..., None]
    d0 = np.logical_and(r00 > r11, r00 > r22)[..., None]
    d1 = (r11 > r22)[..., None]
    q = np.where(tp, c0, np.where(d0, c1, np.where(d1, c2, c3)))
    xyz, w = q[..., :3], q[..., 3:]
    n = np.linalg.norm(xyz, axis=-1, keepdims=True)
    return ((divide_no_nan(np.full_like(n, 2.0), n) * np.arctan2(n, w)) * xyz).astype(dt)


def align_unit_vectors(a, b):
    """Rotation taking unit a to unit b; identity when (anti)parallel (pt/rotation.py:210-224)."""
    cr = np.cross(a, b)
    dot = (a * b).sum(-1, keepdims=True)
    sn = np.linalg.norm(cr, axis=-1, keepdims=True)
    ang = np.arctan2(sn, dot)
    return rotvec2mat(divide_no_nan(cr * ang, sn).astype(a.dtype))


# ----------------------------------------------------------------------------------------------
# model + forward (pt/bodymodel.py)
# ----------------------------------------------------------------------------------------------
class OracleModel:
    """Constants of one body model in ``dtype`` (pt/bodymodel.py:80-93)."""

    def __init__(self, md, dtype=np.float32, model_name='smpl'):
        dt = np.dtype(dtype)
        self.dtype = dt
        self.model_name = model_name
        # the reference stores float32 buffers; the fp64 oracle upcasts those same fp32 values
        f32 = lambda x: np.asarray(x, np.float32).astype(dt)  # noqa: E731
        self.v_template = f32(md.v_template)
        self.shapedirs = f32(md.shapedirs)
        self.posedirs = f32(md.posedirs)
        self.J_regressor_post_lbs = f32(md.J_regressor_post_lbs)
        self.J_template = f32(md.J_template)
        self.J_shapedirs = f32(md.J_shapedirs)
        self.weights = f32(md.weights)
        self.kid_shapedir = f32(md.kid_shapedir)
        self.kid_J_shapedir = f32(md.kid_J_shapedir)
        self.parents = list(md.kintree_parents)
        self.J = md.num_joints
        self.V = md.num_vertices
        self.S = self.shapedirs.shape[2]

    def forward(self, pose_rotvecs=None, shape_betas=None, trans=None, glob_rotmats=None,
                kid_factor=None):
        """LBS forward (pt/bodymodel.py:121-307): returns vertices, joints, orientations."""
        dt = self.dtype
        J, par = self.J, self.parents
        if glob_rotmats is None:
            B = pose_rotvecs.shape[0]
            rel = rotvec2mat(np.asarray(pose_rotvecs, dt).reshape(B, J, 3))
            glob = [rel[:, 0]]
            for i in range(1, J):
                glob.append(glob[par[i]] @ rel[:, i])
            glob = np.stack(glob, 1)
            rel1 = rel[:, 1:]
        else:
            glob = np.asarray(glob_rotmats, dt)
            B = glob.shape[0]
            rel1 = np.swapaxes(glob[:, par[1:]], -1, -2) @ glob[:, 1:]
        betas = np.zeros((B, 0), dt) if shape_betas is None else np.asarray(shape_betas, dt)
        nb = min(betas.shape[1], self.S)
        kid = np.zeros((1,), dt) if kid_factor is None else np.asarray(kid_factor, dt).reshape(-1)
        j = (self.J_template + np.einsum('jcs,bs->bjc', self.J_shapedirs[:, :, :nb], betas[:, :nb])
             + self.kid_J_shapedir[None] * kid[:, None, None])
        bones = j[:, 1:] - j[:, par[1:]]
        rot_bones = np.einsum('bjCc,bjc->bjC', glob[:, par[1:]], bones)
        pos = [j[:, 0]]
        for i in range(1, J):
            pos.append(pos[par[i]] + rot_bones[:, i - 1])
        pos = np.stack(pos, 1)
        t = np.zeros((1, 3), dt) if trans is None else np.asarray(trans, dt)
        feat = rel1.reshape(B, (J - 1) * 9)
        v_posed = (
            self.v_template
            + np.einsum('vcs,bs->bvc', self.shapedirs[:, :, :nb], betas[:, :nb])
            + (feat @ self.posedirs.reshape(self.V * 3, -1).T).reshape(B, self.V, 3)
            + self.kid_shapedir[None] * kid[:, None, None]
        )
        translations = pos - np.einsum('bjCc,bjc->bjC', glob, j)
        rot_blend = np.einsum('vj,bjk->bvk', self.weights, glob.reshape(B, J, 9)).reshape(
            B, self.V, 3, 3
        )
        verts = (
            np.einsum('bvCc,bvc->bvC', rot_blend, v_posed)
            + np.einsum('vj,bjc->bvc', self.weights, translations)
        )
        return dict(
            vertices=(verts + t[:, None]).astype(dt),
            joints=(pos + t[:, None]).astype(dt),
            orientations=glob,
        )


# ----------------------------------------------------------------------------------------------
# fitter (pt/bodyfitter.py)
# ----------------------------------------------------------------------------------------------
class OracleFitter:
    """Default-configuration ``BodyFitter`` (no share_beta / scale / kid / warm start)."""

    def __init__(self, model: OracleModel, enable_kid: bool = False):
        m = self.m = model
        self.enable_kid = enable_kid
        J, par = m.J, m.parents
        self.smpl_family = m.model_name.startswith('smpl')
        # part assignment: dominant skinning weight, SMPL toes -> feet (pt/bodyfitter.py:36-44)
        part = np.argmax(m.weights, axis=1)
        if self.smpl_family:
            part = np.where(part == 10, 7, part)
            part = np.where(part == 11, 8, part)
        self.part = part
        # children-and-self (pt/bodyfitter.py:61-64)
        cas = [[i] for i in range(J)]
        for i in range(1, J):
            cas[par[i]].append(i)
        self.cas = cas
        # buckets (pt/bodyfitter.py:81-97)
        self.multi, self.bone, self.leaf = [], [], []
        for i in range(J):
            if self.smpl_family and i in (10, 11):
                continue
            n = len(cas[i])
            (self.multi if n >= 3 else self.bone if n == 2 else self.leaf).append(i)
        # adjustable parts (pt/bodyfitter.py:101-104)
        self.adjustable = [1, 2, 4, 5, 7, 8, 16, 17, 18, 19] if self.smpl_family else list(range(J))
        stat = sorted(set(self.bone + self.leaf + self.adjustable))
        self.used_parts = stat  # vertices of these parts enter the part sums (:109-114)
        self.sel = {i: np.where(part == i)[0] for i in stat}
        # tree levels (pt/bodyfitter.py:181-192)
        depth = [0] * J
        for i in range(1, J):
            depth[i] = depth[par[i]] + 1
        self.levels = [[i for i in range(J) if depth[i] == d] for d in range(1, max(depth) + 1)]
        # shape directions with the kid blend shape as one more unknown (pt/bodyfitter.py:52-58,
        # :1139-1149); S_all = S (+1)
        self.shapedirs = m.shapedirs
        jsd = m.J_shapedirs
        if enable_kid:
            self.shapedirs = np.concatenate([m.shapedirs, m.kid_shapedir[:, :, None]], axis=2)
            jsd = np.concatenate([m.J_shapedirs, m.kid_J_shapedir[:, :, None]], axis=2)
        self.S_all = self.shapedirs.shape[2]
        self.J_ext = np.concatenate([m.J_template[:, :, None], jsd], axis=2)  # (J,3,S_all+1)
        pw = [0] + par[1:]
        self.bone_ext = self.J_ext - self.J_ext[pw]
        self.default_mesh = m.forward(
            pose_rotvecs=np.zeros((1, 3 * J), m.dtype), shape_betas=np.zeros((1, 0), m.dtype)
        )['vertices'][0]
        adj = set(self.adjustable)
        self.adj_levels = [[i for i in lv if i in adj] for lv in self.levels]
        self.adj_last = max([k for k, a in enumerate(self.adj_levels) if a], default=-1)

    # -- per-part sufficient statistics (pt/bodyfitter.py:235-280) ---------------------------------
    def part_sums(self, t, a, vw):
        dt = self.m.dtype
        B = max(t.shape[0], a.shape[0])
        J = self.m.J
        raw = np.zeros((B, J, 3, 3), dt)
        st = np.zeros((t.shape[0], J, 3), dt)
        sa = np.zeros((B if vw is not None else a.shape[0], J, 3), dt)
        sw = np.zeros((B if vw is not None else 1, J, 1), dt)
        for i in self.used_parts:
            idx = self.sel[i]
            ti, ai = t[:, idx], a[:, idx]
            if vw is not None:
                w = vw[:, idx, None]
                ai = ai * w
                ts = ti * w
                sw[:, i, 0] = vw[:, idx].sum(1)
            else:
                ts = ti
                sw[:, i, 0] = len(idx)
            raw[:, i] = np.einsum('bnr,bnc->brc', ti, np.broadcast_to(ai, (B,) + ai.shape[1:]))
            st[:, i] = ts.sum(1)
            sa[:, i] = ai.sum(1)
        return raw, st, sa, sw

    @staticmethod
    def _centered(raw, st, sa, sw, ct, ca):
        """raw - st ca^T - ct sa^T + sw ct ca^T (pt/bodyfitter.py:1354-1359)."""
        return (
            raw
            - st[..., :, None] * ca[..., None, :]
            - ct[..., :, None] * sa[..., None, :]
            + sw[..., None] * (ct[..., :, None] * ca[..., None, :])
        )

    # -- global rotations (pt/bodyfitter.py:1321-1416) --------------------------------------------
    def fit_global_rotations(self, tv, tj, rv, rj, vw, jw):
        m, dt, J = self.m, self.m.dtype, self.m.J
        if tj is None or rj is None:
            tj = np.einsum('jv,bvc->bjc', m.J_regressor_post_lbs, tv)
            rj = np.einsum('jv,bvc->bjc', m.J_regressor_post_lbs, rv)
        B = tv.shape[0]
        raw, st, sa, sw = self.part_sums(tv, rv, vw)
        R = np.zeros((B, J, 3, 3), dt)

        def center(x, i):  # mean of children-and-self, as (1/n)-weighted sum (:124-129)
            n = len(self.cas[i])
            return (x[:, self.cas[i]] * dt.type(1.0 / n)).sum(1)

        for i in self.leaf:
            A = self._centered(raw[:, i], st[:, i], sa[:, i], sw[:, i], center(tj, i), center(rj, i))
            R[:, i] = proj_so3(np.broadcast_to(A, (B, 3, 3)).astype(dt))
        for i in self.multi:  # Kabsch on the part's joints only (:1361-1383)
            js = self.cas[i]
            tjs, rjs = tj[:, js], rj[:, js]
            if jw is not None:
                w = jw[:, js, None]
                rjs_w = rjs * w
                ts = tjs * w
                swj = jw[:, js].sum(1)[:, None]
            else:
                rjs_w, ts = rjs, tjs
                swj = np.full((1, 1), float(len(js)), dt)
            rawj = np.einsum('bnr,bnc->brc', tjs, np.broadcast_to(rjs_w, (B,) + rjs_w.shape[1:]))
            A = self._centered(rawj, ts.sum(1), rjs_w.sum(1), swj, center(tj, i), center(rj, i))
            R[:, i] = proj_so3(np.broadcast_to(A, (B, 3, 3)).astype(dt))
        for i in self.bone:  # swing from the bone, twist from the vertices (:1389-1412)
            k, c = self.cas[i]
            b_ref = rj[:, c] - rj[:, k]
            b_tgt = tj[:, c] - tj[:, k]
            b_ref = divide_no_nan(b_ref, np.linalg.norm(b_ref, axis=-1, keepdims=True))
            b_tgt = divide_no_nan(b_tgt, np.linalg.norm(b_tgt, axis=-1, keepdims=True))
            b_ref = np.broadcast_to(b_ref, (B, 3)).astype(dt)
            Rsw = align_unit_vectors(b_ref, b_tgt)
            A = self._centered(raw[:, i], st[:, i], sa[:, i], sw[:, i], center(tj, i), center(rj, i))
            H = Rsw @ np.swapaxes(np.broadcast_to(A, (B, 3, 3)), -1, -2)
            trH = H[:, 0, 0] + H[:, 1, 1] + H[:, 2, 2]
            bHb = np.einsum('br,brc,bc->b', b_tgt, H, b_tgt)
            vee = np.stack(
                [H[:, 1, 2] - H[:, 2, 1], H[:, 2, 0] - H[:, 0, 2], H[:, 0, 1] - H[:, 1, 0]], -1
            )
            ang = np.arctan2((b_tgt * vee).sum(-1), trH - bHb)
            R[:, i] = rotvec2mat((b_tgt * ang[:, None]).astype(dt)) @ Rsw
        if self.smpl_family:  # toes take the feet (:147-156)
            R[:, 10] = R[:, 7]
            R[:, 11] = R[:, 8]
        return R

    # -- shape solve (pt/bodyfitter.py:840-1102) --------------------------------------------------
    def fit_shape(self, G, tv, tj, vw, jw, beta_reg, beta_reg2, kid_reg=None, reg_ref=None,
                  share_beta=False, scale_mode=0, scale_reg=0.0):
        """``reg_ref`` (B, S_all): values the ridge pulls towards (beta/kid_regularizer_reference,
        :1072-1081, :1224-1255); zeros when None."""
        m, dt, J, S, par = self.m, self.m.dtype, self.m.J, self.S_all, self.m.parents
        B = tv.shape[0]
        Gpar = np.concatenate([np.broadcast_to(np.eye(3, dtype=dt), (B, 1, 3, 3)), G[:, par[1:]]], 1)
        rel = np.swapaxes(Gpar, -1, -2) @ G
        # FK of joint positions with their beta-Jacobian, level by level (:880-907)
        P = np.zeros((B, J, 3, S + 1), dt)
        P[:, 0] = self.J_ext[0]
        for lv in self.levels:
            for i in lv:
                P[:, i] = P[:, par[i]] + G[:, par[i]] @ self.bone_ext[i]
        T = P - G @ self.J_ext[None]  # (:909-911)
        feat = rel[:, 1:].reshape(B, (J - 1) * 9)
        v_posed = m.v_template + (feat @ m.posedirs.reshape(m.V * 3, -1).T).reshape(B, m.V, 3)
        # per-vertex blended rotation / position / Jacobian (:1000-1016)
        Rb = np.einsum('vj,bjk->bvk', m.weights, G.reshape(B, J, 9)).reshape(B, m.V, 3, 3)
        Tb = np.einsum('vj,bjcs->bvcs', m.weights, T)  # (B,V,3,S+1)
        pos = np.einsum('bvCc,bvc->bvC', Rb, v_posed) + Tb[..., 0]
        jac = np.einsum('bvCc,vcs->bvCs', Rb, self.shapedirs) + Tb[..., 1:]
        b = tv - pos
        # effective weights (:1018-1028)
        if tj is not None and vw is not None and jw is not None:
            evw, ejw = vw, jw
        elif tj is None and vw is not None:
            evw, ejw = vw, None
        else:
            evw, ejw = None, None

        def block(A, bb, w):  # raw normal equations of one point block, fp32 -> fp64 (:1598-1625)
            n = A.shape[1]
            Af = A.reshape(B, n * 3, S)
            bf = bb.reshape(B, n * 3, 1)
            if w is None:
                WA = A
                wb = bb
                W = np.full((B, 1, 1), float(n), np.float64)
            else:
                WA = A * w[:, :, None, None]
                wb = bb * w[:, :, None]
                W = w.sum(1).reshape(B, 1, 1).astype(np.float64)
            WAf = WA.reshape(B, n * 3, S)
            gram = np.swapaxes(WAf, 1, 2) @ Af
            rhs = np.swapaxes(WAf, 1, 2) @ bf
            sA = WA.sum(1)  # (B,3,S) per coordinate
            sb = wb.sum(1)[..., None]  # (B,3,1)
            return tuple(x.astype(np.float64) for x in (gram, rhs, sA, sb)) + (W,)

        jac_s, Pj = jac, P[..., 1:]
        if scale_mode:  # the scale column of the design matrix (:1170-1175)
            cv = -tv if scale_mode == 1 else pos
            jac_s = np.concatenate([jac, cv[..., None]], -1)
            if tj is not None:
                cj = -tj if scale_mode == 1 else P[..., 0]
                Pj = np.concatenate([P[..., 1:], cj[..., None]], -1)
            S = S + 1
        gram, rhs, sA, sb, W = block(jac_s, b, evw)
        if tj is not None:
            g2, r2, sA2, sb2, W2 = block(Pj, tj - P[..., 0], ejw)
            gram, rhs, sA, sb, W = gram + g2, rhs + r2, sA + sA2, sb + sb2, W + W2
        Ws = np.where(W == 0, 1.0, W)
        gram_c = gram - np.swapaxes(sA, 1, 2) @ sA / Ws
        rhs_c = rhs - np.swapaxes(sA, 1, 2) @ sb / Ws
        n_plain = m.S
        lam = np.concatenate([np.full(2, float(beta_reg2)), np.full(n_plain - 2, float(beta_reg))])
        if self.enable_kid:  # kid_regularizer defaults to beta_regularizer (pt/bodyfitter.py:1235-1242)
            lam = np.concatenate([lam, [float(beta_reg if kid_reg is None else kid_reg)]])
        if scale_mode:
            lam = np.concatenate([lam, [float(scale_reg)]])
            if reg_ref is not None:
                reg_ref = np.concatenate([np.asarray(reg_ref, np.float64), np.zeros((B, 1))], 1)
        if reg_ref is not None and not share_beta:  # (share_beta with a scale unknown: below)
            # the all-shared branch of lstsq_partial_share calls lstsq WITHOUT l2_regularizer_rhs
            # (pt/lstsq.py:45-47): with share_beta the ridge pulls towards zero whatever the reference
            rhs_c = rhs_c + (lam[None] * np.asarray(reg_ref, np.float64))[..., None]
        if share_beta and scale_mode:
            # shared shape, one scale per instance: lstsq_partial_share with n_shared = S (pt/lstsq.py
            # :50-90).  Regressing the shared columns and the right-hand side on the independent column
            # and solving the shared part on the residuals is the Schur complement of the scale entry.
            # The ridge rows are appended to the design matrix with weight lambda and right-hand side
            # lambda * reference (:52-61), so the reference enters the normal equations with lambda^2
            # (the plain solve adds lambda * reference, :20-21) -- restated as the reference computes it.
            if reg_ref is not None:
                rhs_c = rhs_c + (lam[None] ** 2 * np.asarray(reg_ref, np.float64))[..., None]
            Mfull = gram_c + np.diag(lam)[None]
            n = S - 1
            Mss, mv, c = Mfull[:, :n, :n], Mfull[:, :n, n:], Mfull[:, n:, n:]
            rs, rho = rhs_c[:, :n], rhs_c[:, n:]
            Msum = (Mss - mv @ np.swapaxes(mv, 1, 2) / c).sum(0)
            xs = np.linalg.solve(Msum, (rs - mv * rho / c).sum(0))
            sig = (rho - np.swapaxes(mv, 1, 2) @ xs[None]) / c
            x = np.concatenate([np.broadcast_to(xs[None], rs.shape), sig], 1)
        elif share_beta:
            # one shape for the whole batch: the regularised normal equations of all instances are summed
            # before the solve (pt/lstsq.py:24-26 via lstsq_partial_share :47-49; every instance keeps
            # its own centring, hence its own translation)
            Msum = (gram_c + np.diag(lam)[None]).sum(0)
            x = np.broadcast_to(np.linalg.solve(Msum, rhs_c.sum(0))[None], rhs_c.shape).copy()
        else:
            x = np.linalg.solve(gram_c + np.diag(lam), rhs_c)  # SPD; reference uses Cholesky (:1083-1084)
        trans = (sb / Ws - (sA / Ws) @ x)[..., 0].astype(dt)
        beta = x[..., 0].astype(dt)
        scale_corr = None
        beta_eval = beta
        if scale_mode:
            scale_corr = (beta[:, -1] + 1).astype(dt)
            beta = beta[:, :-1]
            # scale_fit: the mesh is evaluated at shape / scale (:1289-1293) while the RETURNED
            # shape_betas / kid_factor (and what the refinement is handed) stay undivided — the result
            # dict is filled before the division (:1277-1283)
            beta_eval = (beta / scale_corr[:, None]).astype(dt) if scale_mode == 2 else beta
        joints = P[..., 0] + np.einsum('bjcs,bs->bjc', P[..., 1:], beta_eval) + trans[:, None]
        verts = pos + np.einsum('bvcs,bs->bvc', jac, beta_eval) + trans[:, None]
        out = dict(shape_betas=beta[:, :n_plain], trans=trans, joints=joints.astype(dt),
                   vertices=verts.astype(dt), gram_cen=gram_c, rhs_cen=rhs_c, beta_all=beta)
        if scale_corr is not None:
            out['scale_corr'] = scale_corr
        if self.enable_kid:
            out['kid_factor'] = beta[:, n_plain]
        return out

    # -- dependent refinement, level-batched branch (pt/bodyfitter.py:1418-1544) -------------------
    def fit_global_rotations_dependent(self, tv, tj, rv, rj_true, vw, jw, G, beta, trans,
                                       rest_joints=None, scale_corr=None):
        """beta: all shape unknowns (betas + kid when enabled); ``rest_joints`` overrides the rest-pose
        joints computed from it (known-shape entry point); ``scale_corr`` (B,1,1) scales them (:1449-1450)."""
        m, dt, J, par = self.m, self.m.dtype, self.m.J, self.m.parents
        B = tv.shape[0]
        if tj is None:
            tj = np.einsum('jv,bvc->bjc', m.J_regressor_post_lbs, tv)
            rj = np.einsum('jv,bvc->bjc', m.J_regressor_post_lbs, rv)
        else:
            rj = rj_true
        if rest_joints is not None:
            j = rest_joints
        else:
            j = self.J_ext[None, :, :, 0] + np.einsum('jcs,bs->bjc', self.J_ext[:, :, 1:], beta)
        if scale_corr is not None:
            j = j * scale_corr
        jpar = np.concatenate([np.zeros((B, 1, 3), dt), j[:, par[1:]]], 1)
        bones = j - jpar
        raw, st, sa, sw = self.part_sums(tv, rv, vw)
        rots = G.copy()
        pos = np.zeros((B, J, 3), dt)
        pos[:, 0] = j[:, 0] + trans
        for k in range(self.adj_last + 1):
            for i in self.levels[k]:
                pos[:, i] = pos[:, par[i]] + np.einsum('bCc,bc->bC', rots[:, par[i]], bones[:, i])
            new = {}
            for i in self.adj_levels[k]:
                ct, ca = pos[:, i], rj_true[:, i]
                A = self._centered(raw[:, i], st[:, i], sa[:, i], sw[:, i], ct, ca)
                js = self.cas[i]
                est = tj[:, js] - ct[:, None]
                dfl = rj[:, js] - ca[:, None]
                if jw is not None:
                    dfl = dfl * jw[:, js, None]
                A = A + np.swapaxes(est, 1, 2) @ dfl
                new[i] = proj_so3(A.astype(dt)) @ G[:, i]
            for i, Rn in new.items():
                rots[:, i] = Rn
        if self.smpl_family:
            rots[:, 10] = rots[:, 7]
            rots[:, 11] = rots[:, 8]
        return rots

    # -- driver (pt/bodyfitter.py:283-549) ----------------------------------------------------------
    def fit(self, target_vertices, target_joints=None, vertex_weights=None, joint_weights=None,
            num_iter=1, beta_regularizer=1.0, beta_regularizer2=0.0, final_adjust_rots=True,
            return_stages=False, kid_regularizer=None, initial_pose_rotvecs=None,
            initial_shape_betas=None, initial_kid_factor=None, share_beta=False, scale_target=False,
            scale_fit=False, scale_regularizer=0.0):
        m, dt, J, par = self.m, self.m.dtype, self.m.J, self.m.parents
        tv = np.asarray(target_vertices, dt)
        tj = None if target_joints is None else np.asarray(target_joints, dt)
        vw = None if vertex_weights is None else np.asarray(vertex_weights, dt)
        jw = None if joint_weights is None else np.asarray(joint_weights, dt)
        B = tv.shape[0]
        if tj is None:  # (:355-361)
            mean = tv.mean(1)
            tv = tv - mean[:, None]
        else:
            mean = np.concatenate([tv, tj], 1).mean(1)
            tv = tv - mean[:, None]
            tj = tj - mean[:, None]
        stages = {}
        reg_ref = None
        if initial_pose_rotvecs is not None or initial_shape_betas is not None:  # warm start (:363-382)
            pose0 = (np.zeros((B, 3 * J), dt) if initial_pose_rotvecs is None
                     else np.asarray(initial_pose_rotvecs, dt))
            f = m.forward(pose_rotvecs=pose0, shape_betas=initial_shape_betas, kid_factor=initial_kid_factor)
            G = self.fit_global_rotations(tv, tj, f['vertices'], f['joints'], vw, jw) @ f['orientations']
        else:
            G = self.fit_global_rotations(tv, tj, self.default_mesh[None], m.J_template[None], vw, jw)
        # the ridge references go to EVERY shape solve whenever they are given, also an
        # initial_kid_factor on its own, without a warm first pass (:413-414, :448-449)
        if initial_shape_betas is not None or initial_kid_factor is not None:
            reg_ref = np.zeros((B, self.S_all), np.float64)
            if initial_shape_betas is not None:
                nbg = min(np.asarray(initial_shape_betas).shape[1], m.S)
                reg_ref[:, :nbg] = np.asarray(initial_shape_betas)[:, :nbg]
            if self.enable_kid and initial_kid_factor is not None:
                reg_ref[:, m.S] = np.asarray(initial_kid_factor).reshape(-1)
        stages['glob_rotmats_iter0'] = G.copy()
        for it in range(num_iter - 1):
            r = self.fit_shape(G, tv, tj, vw, jw, beta_regularizer, beta_regularizer2, kid_regularizer,
                               reg_ref, share_beta)
            if it == 0:
                stages['gram_cen0'], stages['rhs_cen0'] = r['gram_cen'], r['rhs_cen']
                stages['shape_betas0'], stages['trans0'] = r['shape_betas'], r['trans']
            rj = r['joints'] if tj is not None else None
            G = self.fit_global_rotations(tv, tj, r['vertices'], rj, vw, jw) @ G
        scale_mode = 1 if scale_target else 2 if scale_fit else 0  # only the LAST solve (:434-455)
        r = self.fit_shape(G, tv, tj, vw, jw, beta_regularizer, beta_regularizer2, kid_regularizer,
                           reg_ref, share_beta, scale_mode, scale_regularizer)
        if num_iter == 1:
            stages['gram_cen0'], stages['rhs_cen0'] = r['gram_cen'], r['rhs_cen']
            stages['shape_betas0'], stages['trans0'] = r['shape_betas'], r['trans']
        sc = r.get('scale_corr')
        if final_adjust_rots:  # (:462-511)
            if scale_target:
                s3 = sc[:, None, None]
                G = self.fit_global_rotations_dependent(
                    tv * s3, None if tj is None else tj * s3, r['vertices'], r['joints'], vw, jw, G,
                    r['beta_all'], r['trans'])
            elif scale_fit:
                s3 = sc[:, None, None]
                shift = (1 - s3) * r['trans'][:, None]
                G = self.fit_global_rotations_dependent(
                    tv, tj, (s3 * r['vertices'] + shift).astype(dt), (s3 * r['joints'] + shift).astype(dt),
                    vw, jw, G, r['beta_all'], r['trans'], scale_corr=s3)
            else:
                G = self.fit_global_rotations_dependent(
                    tv, tj, r['vertices'], r['joints'], vw, jw, G, r['beta_all'], r['trans']
                )
        Gpar = np.concatenate([np.broadcast_to(np.eye(3, dtype=dt), (B, 1, 3, 3)), G[:, par[1:]]], 1)
        rel = np.swapaxes(Gpar, -1, -2) @ G
        mean_out = mean * sc[:, None] if scale_target else mean / sc[:, None] if scale_fit else mean  # (:513-519)
        out = dict(
            pose_rotvecs=mat2rotvec(rel).reshape(B, J * 3),
            shape_betas=r['shape_betas'],
            trans=(r['trans'] + mean_out).astype(dt),
            orientations=G,
            relative_orientations=rel,
        )
        if self.enable_kid:
            out['kid_factor'] = r['kid_factor']
        if sc is not None:
            out['scale_corr'] = sc
        if return_stages:
            out['stages'] = stages
        return out

    # -- shape + translation for a known pose (pt/bodyfitter.py:552-653) -----------------------------
    def fit_with_known_pose(self, pose_rotvecs, target_vertices, target_joints=None,
                            vertex_weights=None, joint_weights=None, beta_regularizer=1.0,
                            beta_regularizer2=0.0, scale_regularizer=0.0, kid_regularizer=None,
                            share_beta=False, scale_target=False, scale_fit=False,
                            beta_regularizer_reference=None, kid_regularizer_reference=None):
        """One shape solve at the rotations of ``pose_rotvecs``; the options go straight into the solve
        (:623-638) and the UNSCALED target mean is added back to the translation (:642-643)."""
        if scale_target and scale_fit:
            raise ValueError('Only one of estim_scale_target and estim_scale_fit can be True')
        m, dt, J, par = self.m, self.m.dtype, self.m.J, self.m.parents
        tv = np.asarray(target_vertices, dt)
        tj = None if target_joints is None else np.asarray(target_joints, dt)
        vw = None if vertex_weights is None else np.asarray(vertex_weights, dt)
        jw = None if joint_weights is None else np.asarray(joint_weights, dt)
        B = tv.shape[0]
        if tj is None:
            mean = tv.mean(1)
            tv = tv - mean[:, None]
        else:
            mean = np.concatenate([tv, tj], 1).mean(1)
            tv, tj = tv - mean[:, None], tj - mean[:, None]
        rel = rotvec2mat(np.asarray(pose_rotvecs, dt).reshape(B, J, 3))
        glob = [rel[:, 0]]
        for i in range(1, J):
            glob.append(glob[par[i]] @ rel[:, i])
        G = np.stack(glob, 1)
        reg_ref = None
        if beta_regularizer_reference is not None or kid_regularizer_reference is not None:
            # references padded with zeros to all betas, the kid column appended (:1224-1246)
            reg_ref = np.zeros((B, self.S_all), np.float64)
            if beta_regularizer_reference is not None:
                br = np.asarray(beta_regularizer_reference, np.float64)[:, :m.S]
                reg_ref[:, :br.shape[1]] = br
            if kid_regularizer_reference is not None and self.enable_kid:
                reg_ref[:, m.S] = np.asarray(kid_regularizer_reference, np.float64).reshape(B)
        r = self.fit_shape(G, tv, tj, vw, jw, beta_regularizer, beta_regularizer2, kid_regularizer,
                           reg_ref=reg_ref, share_beta=share_beta,
                           scale_mode=1 if scale_target else 2 if scale_fit else 0, scale_reg=scale_regularizer)
        out = dict(shape_betas=r['shape_betas'], trans=(r['trans'] + mean).astype(dt), orientations=G)
        if self.enable_kid:
            out['kid_factor'] = r['kid_factor']
        if 'scale_corr' in r:
            out['scale_corr'] = r['scale_corr']
        return out

    # -- pose + translation for a known shape (pt/bodyfitter.py:655-838) -----------------------------
    def fit_with_known_shape(self, shape_betas, target_vertices, target_joints=None,
                             vertex_weights=None, joint_weights=None, kid_factor=None, num_iter=1,
                             final_adjust_rots=True, initial_pose_rotvecs=None, scale_fit=False):
        m, dt, J, par = self.m, self.m.dtype, self.m.J, self.m.parents
        tv = np.asarray(target_vertices, dt)
        tj = None if target_joints is None else np.asarray(target_joints, dt)
        vw = None if vertex_weights is None else np.asarray(vertex_weights, dt)
        jw = None if joint_weights is None else np.asarray(joint_weights, dt)
        betas = np.asarray(shape_betas, dt)
        B = tv.shape[0]
        if tj is None:  # (:710-717)
            mean = tv.mean(1)
            tv = tv - mean[:, None]
        else:
            mean = np.concatenate([tv, tj], 1).mean(1)
            tv, tj = tv - mean[:, None], tj - mean[:, None]
        pose0 = (np.zeros((B, 3 * J), dt) if initial_pose_rotvecs is None
                 else np.asarray(initial_pose_rotvecs, dt))
        f = m.forward(pose_rotvecs=pose0, shape_betas=betas, kid_factor=kid_factor)  # (:719-723)
        G = self.fit_global_rotations(tv, tj, f['vertices'], f['joints'], vw, jw) @ f['orientations']
        for _ in range(num_iter - 1):  # (:740-757)
            f = m.forward(glob_rotmats=G, shape_betas=betas, kid_factor=kid_factor)
            rj = f['joints'] if tj is not None else None
            G = self.fit_global_rotations(tv, tj, f['vertices'], rj, vw, jw) @ G
        f = m.forward(glob_rotmats=G, shape_betas=betas, kid_factor=kid_factor)
        rv, rj = f['vertices'], f['joints']
        scale, trans = fit_scale_and_translation(tv, rv, tj, rj, vw, jw, scale_fit)
        if final_adjust_rots:  # (:774-802)
            nb = min(betas.shape[1], m.S)
            rest = m.J_template + np.einsum('jcs,bs->bjc', m.J_shapedirs[:, :, :nb], betas[:, :nb])
            if kid_factor is not None:
                rest = rest + m.kid_J_shapedir[None] * np.asarray(kid_factor, dt).reshape(-1, 1, 1)
            sc = None if scale is None else scale[:, None, None]
            sv = rv if scale is None else sc * rv
            sj = rj if scale is None else sc * rj
            G = self.fit_global_rotations_dependent(
                tv, tj, (sv + trans[:, None]).astype(dt), (sj + trans[:, None]).astype(dt), vw, jw, G,
                None, trans, rest_joints=rest.astype(dt), scale_corr=sc)
        Gpar = np.concatenate([np.broadcast_to(np.eye(3, dtype=dt), (B, 1, 3, 3)), G[:, par[1:]]], 1)
        rel = np.swapaxes(Gpar, -1, -2) @ G
        out = dict(pose_rotvecs=mat2rotvec(rel).reshape(B, J * 3), trans=(trans + mean).astype(dt),
                   orientations=G, relative_orientations=rel)
        if scale is not None:
            out['scale_corr'] = scale
        return out


def fit_scale_and_translation(tv, rv, tj, rj, vw=None, jw=None, scale=False):
    """Weighted similarity alignment without rotation (pt/bodyfitter.py:1628-1681): returns
    (scale (B,) or None, trans (B,3)) with target ~ scale * reference + trans."""
    dt = tv.dtype
    if tj is None or rj is None:
        t_both, r_both = tv, rv
        w = vw if vw is not None else np.ones(tv.shape[:2], dt)
    else:
        t_both, r_both = np.concatenate([tv, tj], 1), np.concatenate([rv, rj], 1)
        if vw is not None and jw is not None:
            w = np.concatenate([vw, jw], 1)
        else:
            w = np.ones(t_both.shape[:2], dt)
    w = (w / w.sum(1, keepdims=True)).astype(dt)
    mt = (t_both * w[..., None]).sum(1)
    mr = (r_both * w[..., None]).sum(1)
    if not scale:
        return None, (mt - mr).astype(dt)
    tc, rc = t_both - mt[:, None], r_both - mr[:, None]
    ssq_r = (rc ** 2 * w[..., None]).sum((1, 2))
    ssq_t = (tc ** 2 * w[..., None]).sum((1, 2))
    sf = np.sqrt(ssq_t / ssq_r).astype(dt)
    return sf, (mt - sf[:, None] * mr).astype(dt)
