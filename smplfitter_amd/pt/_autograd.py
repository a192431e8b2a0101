"""Differentiable fit: the algorithm of ``BodyFitter.fit`` written with PyTorch operators, for inputs that require
gradients (reference behaviour: ``tests/pt/test_fitter_grad.py:31-99`` — the fit is differentiable with respect to the
target vertices / joints).  The HIP kernels have no backward pass; ``BodyFitter.fit`` routes a call here ONLY when an
input requires a gradient and the model lives on a ``cuda`` device.  Everything else — every call without gradients —
runs the HIP path; this module is never a fallback for it (without gradients the HIP path is hundreds of times faster:
this is a per-part Python loop over dense tensors, written for clarity and for autograd, not for speed).

The steps follow DESIGN.md / SURVEY.md Appendix B (the same restatement the CPU oracle checks the kernels with):
centring, part rotations (Kabsch on the multi-joint and leaf parts, swing + twist on the bone parts), the shape solve
from the weighted-mean-centred normal equations (fp32 products, fp64 combination and solve), the dependent refinement of
the adjustable parts, relative rotations and the log map.  Reference lines are cited per function
(``pt/bodyfitter.py`` / ``pt/rotation.py`` of the reference).  Supported here: target joints given or omitted, vertex and
joint weights, ``num_iter``, both ridge weights, ``final_adjust_rots``, ``enable_kid`` / ``kid_regularizer``; the
remaining options raise ``NotImplementedError`` when combined with gradients.
"""

from __future__ import annotations

from typing import Optional

import torch


def _div0(a, b):
    """a / b, 0 where b == 0 (pt/rotation.py:8-11) — with a gradient that is finite at b == 0."""
    safe = torch.where(b == 0, torch.ones_like(b), b)
    return torch.where(b == 0, torch.zeros_like(a), a / safe)


def proj_so3(A):
    """Nearest rotation: U V^T with the reflection fix on the last singular pair (pt/rotation.py:100-110)."""
    U, _, Vh = torch.linalg.svd(A)
    T = U @ Vh
    mirror = T - 2 * U[..., :, -1:] @ Vh[..., -1:, :]
    return torch.where((torch.linalg.det(T) < 0)[..., None, None], mirror, T)


def rotvec2mat(r):
    """Rodrigues (pt/rotation.py:236-258)."""
    angle = torch.linalg.norm(r, dim=-1, keepdim=True)
    axis = _div0(r, angle)
    s, c = torch.sin(angle) * axis, torch.cos(angle)
    c1 = (1.0 - c) * axis
    ax, ay, az = axis.unbind(-1)
    c1x, c1y, _ = c1.unbind(-1)
    sx, sy, sz = s.unbind(-1)
    d = c1 * axis + c
    m = torch.stack([d[..., 0], c1x * ay - sz, c1x * az + sy, c1x * ay + sz, d[..., 1], c1y * az - sx,
                     c1x * az - sy, c1y * az + sx, d[..., 2]], dim=-1)
    return m.reshape(r.shape[:-1] + (3, 3))


def mat2rotvec(R):
    """Quaternion by the four-way branch, then 2 atan2(|xyz|, w) / |xyz| * xyz (pt/rotation.py:261-289)."""
    r00, r01, r02 = R[..., 0, 0], R[..., 0, 1], R[..., 0, 2]
    r10, r11, r12 = R[..., 1, 0], R[..., 1, 1], R[..., 1, 2]
    r20, r21, r22 = R[..., 2, 0], R[..., 2, 1], R[..., 2, 2]
    trace = r00 + r11 + r22
    c0 = torch.stack([r21 - r12, r02 - r20, r10 - r01, 1 + trace], -1)
    c1 = torch.stack([(1 - r22) + (r00 - r11), r10 + r01, r02 + r20, r21 - r12], -1)
    c2 = torch.stack([r10 + r01, (1 - r22) - (r00 - r11), r21 + r12, r02 - r20], -1)
    c3 = torch.stack([r02 + r20, r21 + r12, (1 + r22) - (r00 + r11), r10 - r01], -1)
    q = torch.where((trace > 0)[..., None], c0,
                    torch.where(((r00 > r11) & (r00 > r22))[..., None], c1, torch.where((r11 > r22)[..., None], c2, c3)))
    xyz, w = q[..., :3], q[..., 3:]
    n = torch.linalg.norm(xyz, dim=-1, keepdim=True)
    return _div0(torch.full_like(n, 2.0), n) * torch.atan2(n, w) * xyz


def align_unit_vectors(a, b):
    """Rotation taking unit a to unit b; identity when (anti)parallel (pt/rotation.py:210-224)."""
    cr = torch.linalg.cross(a, b)
    sn = torch.linalg.norm(cr, dim=-1, keepdim=True)
    ang = torch.atan2(sn, (a * b).sum(-1, keepdim=True))
    return rotvec2mat(_div0(cr * ang, sn))


def _centered(raw, st, sa, sw, ct, ca):
    """raw - st ca^T - ct sa^T + sw ct ca^T (pt/bodyfitter.py:1354-1359)."""
    return (raw - st[..., :, None] * ca[..., None, :] - ct[..., :, None] * sa[..., None, :]
            + sw[..., None] * (ct[..., :, None] * ca[..., None, :]))


class TorchFit:
    """Tables of ``BodyFitter.__init__`` (pt/bodyfitter.py:25-233) for the torch restatement of one body model."""

    def __init__(self, body_model, enable_kid: bool = False):
        bm = self.bm = body_model
        self.enable_kid = enable_kid
        J, par = bm.num_joints, list(bm.kintree_parents)
        self.J, self.par = J, par
        self.smpl = bm.model_name.startswith('smpl')
        part = torch.argmax(bm.weights, dim=1).cpu()
        if self.smpl:  # toes -> feet (:36-44)
            part = torch.where(part == 10, torch.full_like(part, 7), part)
            part = torch.where(part == 11, torch.full_like(part, 8), part)
        cas = [[i] for i in range(J)]
        for i in range(1, J):
            cas[par[i]].append(i)
        self.cas = cas
        self.multi, self.bone, self.leaf = [], [], []
        for i in range(J):  # buckets (:81-97)
            if self.smpl and i in (10, 11):
                continue
            n = len(cas[i])
            (self.multi if n >= 3 else self.bone if n == 2 else self.leaf).append(i)
        if not self.smpl:
            raise NotImplementedError('non-SMPL-family models are not supported')
        self.adjustable = [1, 2, 4, 5, 7, 8, 16, 17, 18, 19]
        self.used_parts = sorted(set(self.bone + self.leaf + self.adjustable))
        self.sel = {i: torch.nonzero(part == i).reshape(-1) for i in self.used_parts}
        depth = [0] * J
        for i in range(1, J):
            depth[i] = depth[par[i]] + 1
        self.levels = [[i for i in range(J) if depth[i] == d] for d in range(1, max(depth) + 1)]
        adj = set(self.adjustable)
        self.adj_levels = [[i for i in lv if i in adj] for lv in self.levels]
        self.adj_last = max([k for k, a in enumerate(self.adj_levels) if a], default=-1)

    # constants on the device of the call (the model's buffers follow .to() / .cuda())
    def _consts(self, device):
        key = (str(device), self.bm.v_template.data_ptr())  # (rebuilt when the model's buffers move)
        if getattr(self, '_consts_key', None) == key:
            return self._consts_val
        self._consts_val = self._build_consts(device)
        self._consts_key = key
        return self._consts_val

    def _build_consts(self, device):
        bm = self.bm
        c = {k: getattr(bm, k).to(device) for k in ('v_template', 'shapedirs', 'posedirs', 'weights', 'J_template',
                                                    'J_shapedirs', 'J_regressor_post_lbs', 'kid_shapedir', 'kid_J_shapedir')}
        sd, jsd = c['shapedirs'], c['J_shapedirs']
        if self.enable_kid:  # the kid blend shape is one more shape unknown (:52-58, :1139-1149)
            sd = torch.cat([sd, c['kid_shapedir'][:, :, None]], 2)
            jsd = torch.cat([jsd, c['kid_J_shapedir'][:, :, None]], 2)
        c['sd_all'] = sd
        c['J_ext'] = torch.cat([c['J_template'][:, :, None], jsd], 2)  # (J, 3, S + 1)
        pw = [0] + self.par[1:]
        c['bone_ext'] = c['J_ext'] - c['J_ext'][pw]
        # default mesh = forward(zero pose, zero shape) (:49)
        eye = torch.eye(3, device=device).reshape(9)
        feat = eye.repeat(self.J - 1)
        vp = c['v_template'] + (c['posedirs'].reshape(-1, feat.numel()) @ feat).reshape(-1, 3)
        c['default_mesh'] = vp * c['weights'].sum(1, keepdim=True)
        c['sel'] = {i: v.to(device) for i, v in self.sel.items()}
        return c

    def _part_sums(self, c, t, a, vw):
        """Per-part sums of target x reference products (pt/bodyfitter.py:235-280)."""
        B, J = max(t.shape[0], a.shape[0]), self.J
        z33, z3, z1 = t.new_zeros(B, 3, 3), t.new_zeros(B, 3), t.new_zeros(B, 1)
        raw, st, sa, sw = [z33] * J, [z3] * J, [z3] * J, [z1] * J
        for i in self.used_parts:
            idx = c['sel'][i]
            ti, ai = t[:, idx], a[:, idx].expand(B, -1, -1)
            if vw is not None:
                w = vw[:, idx, None]
                ai, ts = ai * w, ti * w
                sw[i] = vw[:, idx].sum(1, keepdim=True)
            else:
                ts = ti
                sw[i] = t.new_full((B, 1), float(len(idx)))
            raw[i] = torch.einsum('bnr,bnc->brc', ti, ai)
            st[i], sa[i] = ts.sum(1), ai.sum(1)
        return torch.stack(raw, 1), torch.stack(st, 1), torch.stack(sa, 1), torch.stack(sw, 1)

    def _rotations(self, c, tv, tj, rv, rj, vw, jw):
        """Part rotations from the centred covariances (pt/bodyfitter.py:1321-1416)."""
        B, J = tv.shape[0], self.J
        if tj is None or rj is None:
            reg = c['J_regressor_post_lbs']
            tj, rj = torch.einsum('jv,bvc->bjc', reg, tv), torch.einsum('jv,bvc->bjc', reg, rv)
        raw, st, sa, sw = self._part_sums(c, tv, rv, vw)
        rj = rj.expand(B, -1, -1)
        R = [None] * J

        def centre(x, i):
            return x[:, self.cas[i]].mean(1)

        for i in self.leaf:
            R[i] = proj_so3(_centered(raw[:, i], st[:, i], sa[:, i], sw[:, i], centre(tj, i), centre(rj, i)))
        for i in self.multi:  # Kabsch on the part's joints only (:1361-1383)
            js = self.cas[i]
            tjs, rjs = tj[:, js], rj[:, js]
            if jw is not None:
                w = jw[:, js, None]
                rjs_w, ts, swj = rjs * w, tjs * w, jw[:, js].sum(1, keepdim=True)
            else:
                rjs_w, ts, swj = rjs, tjs, tv.new_full((B, 1), float(len(js)))
            rawj = torch.einsum('bnr,bnc->brc', tjs, rjs_w)
            R[i] = proj_so3(_centered(rawj, ts.sum(1), rjs_w.sum(1), swj, centre(tj, i), centre(rj, i)))
        for i in self.bone:  # swing from the bone, twist from the vertices (:1389-1412)
            k, ch = self.cas[i]
            b_ref, b_tgt = rj[:, ch] - rj[:, k], tj[:, ch] - tj[:, k]
            b_ref = _div0(b_ref, torch.linalg.norm(b_ref, dim=-1, keepdim=True))
            b_tgt = _div0(b_tgt, torch.linalg.norm(b_tgt, dim=-1, keepdim=True))
            Rsw = align_unit_vectors(b_ref, b_tgt)
            A = _centered(raw[:, i], st[:, i], sa[:, i], sw[:, i], centre(tj, i), centre(rj, i))
            H = Rsw @ A.transpose(-1, -2)
            trH = H[:, 0, 0] + H[:, 1, 1] + H[:, 2, 2]
            bHb = torch.einsum('br,brc,bc->b', b_tgt, H, b_tgt)
            vee = torch.stack([H[:, 1, 2] - H[:, 2, 1], H[:, 2, 0] - H[:, 0, 2], H[:, 0, 1] - H[:, 1, 0]], -1)
            ang = torch.atan2((b_tgt * vee).sum(-1), trH - bHb)
            R[i] = rotvec2mat(b_tgt * ang[:, None]) @ Rsw
        R[10], R[11] = R[7], R[8]  # toes take the feet (:147-156)
        return torch.stack(R, 1)

    def _shape(self, c, G, tv, tj, vw, jw, beta_reg, beta_reg2, kid_reg):
        """Shape + translation from the centred normal equations (pt/bodyfitter.py:840-1102)."""
        B, J, par = tv.shape[0], self.J, self.par
        S = c['sd_all'].shape[2]
        eye = torch.eye(3, device=tv.device).expand(B, 1, 3, 3)
        rel = torch.cat([eye, G[:, par[1:]]], 1).transpose(-1, -2) @ G
        P = [None] * J  # FK of the joint positions with their beta-Jacobian (:880-907)
        P[0] = c['J_ext'][0].expand(B, -1, -1)
        for lv in self.levels:
            for i in lv:
                P[i] = P[par[i]] + G[:, par[i]] @ c['bone_ext'][i]
        P = torch.stack(P, 1)
        T = P - G @ c['J_ext'][None]
        V = c['v_template'].shape[0]
        feat = rel[:, 1:].reshape(B, (J - 1) * 9)
        v_posed = c['v_template'] + (feat @ c['posedirs'].reshape(V * 3, -1).T).reshape(B, V, 3)
        Rb = torch.einsum('vj,bjk->bvk', c['weights'], G.reshape(B, J, 9)).reshape(B, V, 3, 3)
        Tb = torch.einsum('vj,bjcs->bvcs', c['weights'], T)
        pos = torch.einsum('bvCc,bvc->bvC', Rb, v_posed) + Tb[..., 0]
        jac = torch.einsum('bvCc,vcs->bvCs', Rb, c['sd_all']) + Tb[..., 1:]
        # the weights enter the solve only when both are given (with joints) / vertex weights without joints (:1018-1028)
        if tj is not None and vw is not None and jw is not None:
            evw, ejw = vw, jw
        elif tj is None and vw is not None:
            evw, ejw = vw, None
        else:
            evw, ejw = None, None

        def block(A, bb, w):  # raw normal equations of one block of points: fp32 products, fp64 results (:1598-1625)
            n = A.shape[1]
            if w is None:
                WA, wb, W = A, bb, A.new_full((B, 1, 1), float(n), dtype=torch.float64)
            else:
                WA, wb, W = A * w[:, :, None, None], bb * w[:, :, None], w.sum(1).reshape(B, 1, 1).double()
            WAf = WA.reshape(B, n * 3, S)
            gram = WAf.transpose(1, 2) @ A.reshape(B, n * 3, S)
            rhs = WAf.transpose(1, 2) @ bb.reshape(B, n * 3, 1)
            return gram.double(), rhs.double(), WA.sum(1).double(), wb.sum(1)[..., None].double(), W

        gram, rhs, sA, sb, W = block(jac, tv - pos, evw)
        if tj is not None:
            g2, r2, sA2, sb2, W2 = block(P[..., 1:], tj - P[..., 0], ejw)
            gram, rhs, sA, sb, W = gram + g2, rhs + r2, sA + sA2, sb + sb2, W + W2
        Ws = torch.where(W == 0, torch.ones_like(W), W)
        gram_c = gram - sA.transpose(1, 2) @ sA / Ws
        rhs_c = rhs - sA.transpose(1, 2) @ sb / Ws
        nb = c['shapedirs'].shape[2]
        lam = [float(beta_reg2)] * 2 + [float(beta_reg)] * (nb - 2)
        if self.enable_kid:  # kid_regularizer defaults to beta_regularizer (:1235-1242)
            lam.append(float(beta_reg if kid_reg is None else kid_reg))
        lam = torch.tensor(lam, dtype=torch.float64, device=tv.device)
        x = torch.linalg.solve(gram_c + torch.diag(lam), rhs_c)
        trans = (sb / Ws - (sA / Ws) @ x)[..., 0].to(tv.dtype)
        beta = x[..., 0].to(tv.dtype)
        joints = P[..., 0] + torch.einsum('bjcs,bs->bjc', P[..., 1:], beta) + trans[:, None]
        verts = pos + torch.einsum('bvcs,bs->bvc', jac, beta) + trans[:, None]
        return dict(beta_all=beta, trans=trans, joints=joints, vertices=verts)

    def _refine(self, c, tv, tj, rv, rj_true, vw, jw, G, beta, trans):
        """Dependent refinement of the adjustable parts, level by level (pt/bodyfitter.py:1418-1544)."""
        B, J, par = tv.shape[0], self.J, self.par
        if tj is None:
            reg = c['J_regressor_post_lbs']
            tj, rj = torch.einsum('jv,bvc->bjc', reg, tv), torch.einsum('jv,bvc->bjc', reg, rv)
        else:
            rj = rj_true
        j = c['J_ext'][None, :, :, 0] + torch.einsum('jcs,bs->bjc', c['J_ext'][:, :, 1:], beta)
        bones = j - torch.cat([j.new_zeros(B, 1, 3), j[:, par[1:]]], 1)
        raw, st, sa, sw = self._part_sums(c, tv, rv, vw)
        rots = [G[:, i] for i in range(J)]
        pos = [None] * J
        pos[0] = j[:, 0] + trans
        for k in range(self.adj_last + 1):
            for i in self.levels[k]:
                pos[i] = pos[par[i]] + torch.einsum('bCc,bc->bC', rots[par[i]], bones[:, i])
            new = {}
            for i in self.adj_levels[k]:
                ct, ca = pos[i], rj_true[:, i]
                A = _centered(raw[:, i], st[:, i], sa[:, i], sw[:, i], ct, ca)
                js = self.cas[i]
                dfl = rj[:, js] - ca[:, None]
                if jw is not None:
                    dfl = dfl * jw[:, js, None]
                A = A + (tj[:, js] - ct[:, None]).transpose(1, 2) @ dfl
                new[i] = proj_so3(A) @ G[:, i]
            for i, Rn in new.items():
                rots[i] = Rn
        rots[10], rots[11] = rots[7], rots[8]
        return torch.stack(rots, 1)

    def fit(self, target_vertices: torch.Tensor, target_joints: Optional[torch.Tensor] = None,
            vertex_weights: Optional[torch.Tensor] = None, joint_weights: Optional[torch.Tensor] = None,
            num_iter: int = 1, beta_regularizer: float = 1.0, beta_regularizer2: float = 0.0,
            kid_regularizer: Optional[float] = None, final_adjust_rots: bool = True) -> dict:
        """The driver (pt/bodyfitter.py:283-549); returns the reference's result dictionary."""
        dev = target_vertices.device
        c = self._consts(dev)
        tv = target_vertices.to(torch.float32)
        tj = None if target_joints is None else target_joints.to(torch.float32)
        vw = None if vertex_weights is None else vertex_weights.to(torch.float32)
        jw = None if joint_weights is None else joint_weights.to(torch.float32)
        B, J, par = tv.shape[0], self.J, self.par
        if tj is None:  # centring (:355-361)
            mean = tv.mean(1)
            tv = tv - mean[:, None]
        else:
            mean = torch.cat([tv, tj], 1).mean(1)
            tv, tj = tv - mean[:, None], tj - mean[:, None]
        G = self._rotations(c, tv, tj, c['default_mesh'][None], c['J_template'][None], vw, jw)
        for _ in range(num_iter - 1):
            r = self._shape(c, G, tv, tj, vw, jw, beta_regularizer, beta_regularizer2, kid_regularizer)
            G = self._rotations(c, tv, tj, r['vertices'], r['joints'] if tj is not None else None, vw, jw) @ G
        r = self._shape(c, G, tv, tj, vw, jw, beta_regularizer, beta_regularizer2, kid_regularizer)
        if final_adjust_rots:
            G = self._refine(c, tv, tj, r['vertices'], r['joints'], vw, jw, G, r['beta_all'], r['trans'])
        eye = torch.eye(3, device=dev).expand(B, 1, 3, 3)
        rel = torch.cat([eye, G[:, par[1:]]], 1).transpose(-1, -2) @ G
        nb = c['shapedirs'].shape[2]
        out = dict(pose_rotvecs=mat2rotvec(rel).reshape(B, J * 3), shape_betas=r['beta_all'][:, :nb],
                   trans=r['trans'] + mean, orientations=G, relative_orientations=rel)
        if self.enable_kid:
            out['kid_factor'] = r['beta_all'][:, nb]
        return out
