// Small fixed-size math shared by the HIP kernels (device) and the host unit-test build:
// 3x3 helpers, the SO(3) projection, swing / Rodrigues / log map, with the reference's branch
// structure (reference src/smplfitter/pt/rotation.py).  All matrices are row-major float[9].
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define SF_HD __host__ __device__ __forceinline__
// partial unrolling of loops with run-time trip counts whose bodies are LDS / table reads: the reads of
// several iterations are then issued together instead of one memory latency per iteration (the
// accumulation order is unchanged)
#define SF_UNROLL(n) _Pragma(SF_STR_(unroll n))
#define SF_UNROLL_FULL _Pragma("unroll")
#define SF_STR_(x) #x
#else
#define SF_HD inline
#define SF_UNROLL(n)
#define SF_UNROLL_FULL
#endif

// Scheduling fence for the device compiler: stops it from hoisting the LDS loads of later phases
// above earlier ones (which explodes register pressure in the per-vertex bodies). No-op on host.
#if defined(__HIP_DEVICE_COMPILE__)
#define SF_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define SF_SCHED_FENCE() ((void)0)
#endif

namespace sf {

// a / b, 0 where b == 0 (rotation.py:8-11)
SF_HD float divide_no_nan(float a, float b) { return b == 0.f ? 0.f : a / b; }

SF_HD void m3_mul(const float* a, const float* b, float* o) {  // o = a b
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c)
      o[r * 3 + c] = a[r * 3 + 0] * b[0 * 3 + c] + a[r * 3 + 1] * b[1 * 3 + c] +
                     a[r * 3 + 2] * b[2 * 3 + c];
}
SF_HD void m3_tmul(const float* a, const float* b, float* o) {  // o = a^T b
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c)
      o[r * 3 + c] = a[0 * 3 + r] * b[0 * 3 + c] + a[1 * 3 + r] * b[1 * 3 + c] +
                     a[2 * 3 + r] * b[2 * 3 + c];
}
SF_HD void m3_mult(const float* a, const float* b, float* o) {  // o = a b^T
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c)
      o[r * 3 + c] = a[r * 3 + 0] * b[c * 3 + 0] + a[r * 3 + 1] * b[c * 3 + 1] +
                     a[r * 3 + 2] * b[c * 3 + 2];
}
SF_HD void m3_vec(const float* a, const float* v, float* o) {  // o = a v
  for (int r = 0; r < 3; ++r) o[r] = a[r * 3] * v[0] + a[r * 3 + 1] * v[1] + a[r * 3 + 2] * v[2];
}
SF_HD void m3_identity(float* o) {
  for (int k = 0; k < 9; ++k) o[k] = (k % 4 == 0) ? 1.f : 0.f;
}

// Centered cross-covariance from uncentered part sums about centres (ct, ca):
//   raw - st ca^T - ct sa^T + sw (ct ca^T)          (bodyfitter.py:1354-1359, :1512-1518)
SF_HD void centered_cov(const float* raw, const float* st, const float* sa, float sw,
                        const float* ct, const float* ca, float* A) {
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c)
      A[r * 3 + c] = raw[r * 3 + c] - st[r] * ca[c] - ct[r] * sa[c] + sw * (ct[r] * ca[c]);
}

// 1/sqrt(x) and 1/x in fp64 from the hardware seed + one Newton step on the device (the IEEE
// sqrt / divide sequences cost ~20 dependent fp64 instructions each and dominate the latency of the
// joint-level kernels); plain libm on the host.  Accurate to ~1e-14, far below the fp32 output.
SF_HD double fast_rsqrt(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  double y = __builtin_amdgcn_rsq(x);
  y = y * (1.5 - 0.5 * x * y * y);
  return y * (1.5 - 0.5 * x * y * y);
#else
  return 1.0 / sqrt(x);
#endif
}
SF_HD double fast_rcp(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(x);
  r = r * (2.0 - x * r);
  return r * (2.0 - x * r);
#else
  return 1.0 / x;
#endif
}

// Nearest rotation in Frobenius norm: R = U diag(1,1,det(UV^T)) V^T of the SVD A = U S V^T
// (rotation.py:100-110 computes it with a library SVD + reflection fix; rotation.py:26-97 is the reference's own
// closed form).  Here, in fp64: the closed form through Horn's quaternion eigenproblem (below), and — for the inputs
// whose nearest rotation is not unique — cyclic Jacobi on the symmetric M = A^T A gives V and the singular-value order; U's first two
// columns are A v1, A v2 (Gram-Schmidt), and taking u3 = u1 x u2, v3 = v1 x v2 keeps both frames
// right-handed, which IS the reflection fix (R = u1 v1^T + u2 v2^T + u3 v3^T has det +1).
// Degenerate inputs: A ~ 0 -> identity; rank 1 -> an arbitrary completion (as any SVD would give).
SF_HD void proj_so3(const float* Af, float* R) {
  double a[9];
  double fro2 = 0.0;
  for (int k = 0; k < 9; ++k) {
    a[k] = (double)Af[k];
    fro2 += a[k] * a[k];
  }
  if (!(fro2 > 1e-60)) {  // also catches NaN
    if (fro2 == fro2) {
      m3_identity(R);
    } else {
      for (int k = 0; k < 9; ++k) R[k] = Af[k] * 0.f + (float)fro2;  // propagate NaN
    }
    return;
  }
  const double inv = fast_rsqrt(fro2);
  for (int k = 0; k < 9; ++k) a[k] *= inv;
  // M = A^T A (symmetric): m00 m01 m02 m11 m12 m22
  double m00 = a[0] * a[0] + a[3] * a[3] + a[6] * a[6];
  double m01 = a[0] * a[1] + a[3] * a[4] + a[6] * a[7];
  double m02 = a[0] * a[2] + a[3] * a[5] + a[6] * a[8];
  double m11 = a[1] * a[1] + a[4] * a[4] + a[7] * a[7];
  double m12 = a[1] * a[2] + a[4] * a[5] + a[7] * a[8];
  double m22 = a[2] * a[2] + a[5] * a[5] + a[8] * a[8];
#ifndef SMPLFIT_PROJ_JACOBI_ONLY
  // ---- closed form (round 6; the north star's "closed-form 3x3 SVD").  The nearest rotation is the rotation of the
  // unit quaternion q that maximises q^T N q, N = Horn's symmetric 4 x 4 matrix of A (Horn 1987; with A = sum t a^T,
  // rows = target, Horn's S is A^T).  Its eigenvalues are the signed sums of the singular values, largest
  // lambda = s1 + s2 + sgn(det A) s3 — the reflection case needs no fix-up: q always is a proper rotation — and the
  // characteristic polynomial of the trace-free N needs three invariants of the normalised A only:
  //   P(x) = x^4 - 2 |A|_F^2 x^2 - 8 det(A) x + (2 tr((A^T A)^2) - |A|_F^4),      |A|_F = 1 here.
  // lambda: Newton from the upper bound sqrt(3) (P is convex and increasing above its largest root: monotone
  // convergence, 4 steps for the near-rotations of a fit, <= 12 for ill-conditioned inputs); q: a column of
  // adj(N - lambda I) = P'(lambda) q q^T, the one with the largest diagonal entry.  ~330 fp64 operations without a
  // data-dependent sweep count, against ~600 for the Jacobi sweeps below, which stay for the inputs whose largest
  // eigenvalue is (nearly) double — rank 1, or s2 ~ s3 under a reflection: the nearest rotation is not unique there and
  // the cofactors vanish — detected by |tr adj| = |P'(lambda)| = the product of the gaps to the other eigenvalues.
  {
    const double trm2 = (m00 * m00 + m11 * m11 + m22 * m22) + 2.0 * (m01 * m01 + m02 * m02 + m12 * m12);
    const double det = a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) + a[2] * (a[3] * a[7] - a[4] * a[6]);
    const double c1 = -8.0 * det, c0 = 2.0 * trm2 - 1.0;
    double lam = 1.7320508075688772;
    bool conv = false;
    for (int it = 0; it < 12; ++it) {
      const double l2 = lam * lam;
      const double P = (l2 * l2 - 2.0 * l2) + (c1 * lam + c0), dP = 4.0 * (l2 * lam - lam) + c1;
      if (!(dP > 1e-6)) break;  // (a multiple root at the top: the sweeps below)
      const double d = P * fast_rcp(dP);
      lam -= d;
      if (fabs(d) <= 1e-11 * lam) {  // (quadratic convergence: the step after this one would be below 1e-16)
        conv = true;
        break;
      }
    }
    if (conv) {
      // K = N - lambda I (symmetric), S = A^T:  S_xy = a[3 y + x]
      const double sxx = a[0], sxy = a[3], sxz = a[6], syx = a[1], syy = a[4], syz = a[7], szx = a[2], szy = a[5], szz = a[8];
      const double k00 = (sxx + syy + szz) - lam, k01 = syz - szy, k02 = szx - sxz, k03 = sxy - syx;
      const double k11 = (sxx - syy - szz) - lam, k12 = sxy + syx, k13 = szx + sxz;
      const double k22 = (-sxx + syy - szz) - lam, k23 = syz + szy;
      const double k33 = (-sxx - syy + szz) - lam;
      // adjugate through the 2 x 2 minors of the row pairs (0, 1) and (2, 3)
      const double s0 = k00 * k11 - k01 * k01, s1 = k00 * k12 - k01 * k02, s2 = k00 * k13 - k01 * k03;
      const double s3 = k01 * k12 - k11 * k02, s4 = k01 * k13 - k11 * k03, s5 = k02 * k13 - k12 * k03;
      const double d5 = k22 * k33 - k23 * k23, d4 = k12 * k33 - k13 * k23, d3 = k12 * k23 - k13 * k22;
      const double d2 = k02 * k33 - k03 * k23, d1 = k02 * k23 - k03 * k22;
      const double a00 = k11 * d5 - k12 * d4 + k13 * d3, a01 = -k01 * d5 + k02 * d4 - k03 * d3;
      const double a02 = k13 * s5 - k23 * s4 + k33 * s3, a03 = -k12 * s5 + k22 * s4 - k23 * s3;
      const double a11 = k00 * d5 - k02 * d2 + k03 * d1, a12 = -k03 * s5 + k23 * s2 - k33 * s1;
      const double a13 = k02 * s5 - k22 * s2 + k23 * s1, a22 = k03 * s4 - k13 * s2 + k33 * s0;
      const double a23 = -k02 * s4 + k12 * s2 - k23 * s0, a33 = k02 * s3 - k12 * s1 + k22 * s0;
      // (the diagonal entries share the sign of P'(lambda); gaps: a well-separated largest eigenvalue)
      const double tra = fabs((a00 + a11) + (a22 + a33));
      if (tra > 1e-4) {
        double q0 = a00, q1 = a01, q2 = a02, q3 = a03, best = fabs(a00);
        if (fabs(a11) > best) { best = fabs(a11); q0 = a01; q1 = a11; q2 = a12; q3 = a13; }
        if (fabs(a22) > best) { best = fabs(a22); q0 = a02; q1 = a12; q2 = a22; q3 = a23; }
        if (fabs(a33) > best) { best = fabs(a33); q0 = a03; q1 = a13; q2 = a23; q3 = a33; }
        const double qn = fast_rsqrt((q0 * q0 + q1 * q1) + (q2 * q2 + q3 * q3));
        const double w = q0 * qn, x = q1 * qn, y = q2 * qn, z = q3 * qn;
        R[0] = (float)(1.0 - 2.0 * (y * y + z * z));
        R[1] = (float)(2.0 * (x * y - w * z));
        R[2] = (float)(2.0 * (x * z + w * y));
        R[3] = (float)(2.0 * (x * y + w * z));
        R[4] = (float)(1.0 - 2.0 * (x * x + z * z));
        R[5] = (float)(2.0 * (y * z - w * x));
        R[6] = (float)(2.0 * (x * z - w * y));
        R[7] = (float)(2.0 * (y * z + w * x));
        R[8] = (float)(1.0 - 2.0 * (x * x + y * y));
        return;
      }
    }
  }
#endif
  // eigenvectors as columns of V (v[r][c])
  double v00 = 1, v01 = 0, v02 = 0, v10 = 0, v11 = 1, v12 = 0, v20 = 0, v21 = 0, v22 = 1;
#define SF_JACOBI(app, aqq, apq, arp, arq, vp0, vq0, vp1, vq1, vp2, vq2)                   \
  if (fabs(apq) > 1e-140) {                                                                \
    /* tan of the Jacobi angle: t = e / (d + sgn(d) sqrt(d^2 + e^2)), d = aqq-app, e = 2 apq */ \
    const double dd = aqq - app, ee = 2.0 * apq;                                           \
    const double hh = dd * dd + ee * ee;                                                   \
    const double rt = hh * fast_rsqrt(hh);                                                 \
    const double tt = ee * fast_rcp(dd >= 0.0 ? dd + rt : dd - rt);                        \
    const double cc = fast_rsqrt(tt * tt + 1.0), ss = tt * cc;                             \
    app -= tt * apq;                                                                       \
    aqq += tt * apq;                                                                       \
    apq = 0.0;                                                                             \
    const double rp = arp, rq = arq;                                                       \
    arp = cc * rp - ss * rq;                                                               \
    arq = ss * rp + cc * rq;                                                               \
    double x, y;                                                                           \
    x = vp0; y = vq0; vp0 = cc * x - ss * y; vq0 = ss * x + cc * y;                        \
    x = vp1; y = vq1; vp1 = cc * x - ss * y; vq1 = ss * x + cc * y;                        \
    x = vp2; y = vq2; vp2 = cc * x - ss * y; vq2 = ss * x + cc * y;                        \
  }
  for (int sweep = 0; sweep < 8; ++sweep) {
    const double off = m01 * m01 + m02 * m02 + m12 * m12;
    if (off < 1e-34) break;  // trace(M) == 1 after normalisation: relative threshold
    SF_JACOBI(m00, m11, m01, m02, m12, v00, v01, v10, v11, v20, v21)  // (p,q)=(0,1), r=2
    SF_JACOBI(m00, m22, m02, m01, m12, v00, v02, v10, v12, v20, v22)  // (0,2), r=1
    SF_JACOBI(m11, m22, m12, m01, m02, v01, v02, v11, v12, v21, v22)  // (1,2), r=0
  }
#undef SF_JACOBI
  // pick the two largest eigenvalues (order: e1 >= e2 >= e3)
  double e[3] = {m00, m11, m22};
  double vc[3][3] = {{v00, v10, v20}, {v01, v11, v21}, {v02, v12, v22}};  // vc[k] = k-th eigvec
  int i1 = 0, i3 = 0;
  for (int k = 1; k < 3; ++k) {
    if (e[k] > e[i1]) i1 = k;
    if (e[k] <= e[i3]) i3 = k;
  }
  if (i1 == i3) {  // all equal
    i1 = 0;
    i3 = 2;
  }
  const int i2 = 3 - i1 - i3;
  const double* v1 = vc[i1];
  const double* v2 = vc[i2];
  const double v3[3] = {v1[1] * v2[2] - v1[2] * v2[1], v1[2] * v2[0] - v1[0] * v2[2],
                        v1[0] * v2[1] - v1[1] * v2[0]};
  double u1[3], u2[3], u3[3];
  for (int r = 0; r < 3; ++r) {
    u1[r] = a[r * 3] * v1[0] + a[r * 3 + 1] * v1[1] + a[r * 3 + 2] * v1[2];
    u2[r] = a[r * 3] * v2[0] + a[r * 3 + 1] * v2[1] + a[r * 3 + 2] * v2[2];
  }
  const double i1n = fast_rsqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);  // sigma1 >= 1/sqrt(3)
  for (int r = 0; r < 3; ++r) u1[r] *= i1n;
  const double d12 = u2[0] * u1[0] + u2[1] * u1[1] + u2[2] * u1[2];
  for (int r = 0; r < 3; ++r) u2[r] -= d12 * u1[r];
  double n2 = u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2];
  if (n2 > 1e-28) {
    const double i2n = fast_rsqrt(n2);
    for (int r = 0; r < 3; ++r) u2[r] *= i2n;
  } else {  // rank 1: any unit vector orthogonal to u1
    int k = 0;
    if (fabs(u1[1]) < fabs(u1[k])) k = 1;
    if (fabs(u1[2]) < fabs(u1[k])) k = 2;
    double ek[3] = {0, 0, 0};
    ek[k] = 1.0;
    u2[0] = u1[1] * ek[2] - u1[2] * ek[1];
    u2[1] = u1[2] * ek[0] - u1[0] * ek[2];
    u2[2] = u1[0] * ek[1] - u1[1] * ek[0];
    n2 = fast_rsqrt(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]);
    for (int r = 0; r < 3; ++r) u2[r] *= n2;
  }
  u3[0] = u1[1] * u2[2] - u1[2] * u2[1];
  u3[1] = u1[2] * u2[0] - u1[0] * u2[2];
  u3[2] = u1[0] * u2[1] - u1[1] * u2[0];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c)
      R[r * 3 + c] = (float)(u1[r] * v1[c] + u2[r] * v2[c] + u3[r] * v3[c]);
}

// Rodrigues with the reference's element arithmetic (rotation.py:236-258).
SF_HD void rotvec2mat(const float* rv, float* m) {
  const float angle = sqrtf(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
  const float ax = divide_no_nan(rv[0], angle), ay = divide_no_nan(rv[1], angle),
              az = divide_no_nan(rv[2], angle);
  const float sn = sinf(angle), cs = cosf(angle);
  const float sx = sn * ax, sy = sn * ay, sz = sn * az;
  const float c1 = 1.0f - cs;
  const float c1x = c1 * ax, c1y = c1 * ay, c1z = c1 * az;
  float tmp = c1x * ay;
  m[1] = tmp - sz;
  m[3] = tmp + sz;
  tmp = c1x * az;
  m[2] = tmp + sy;
  m[6] = tmp - sy;
  tmp = c1y * az;
  m[5] = tmp - sx;
  m[7] = tmp + sx;
  m[0] = c1x * ax + cs;
  m[4] = c1y * ay + cs;
  m[8] = c1z * az + cs;
}

// Rotation taking unit a to unit b; zero rotvec (identity) when parallel or exactly antiparallel
// (rotation.py:210-224).
SF_HD void align_unit_vectors(const float* a, const float* b, float* m) {
  const float cx = a[1] * b[2] - a[2] * b[1], cy = a[2] * b[0] - a[0] * b[2],
              cz = a[0] * b[1] - a[1] * b[0];
  const float dot = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
  const float sn = sqrtf(cx * cx + cy * cy + cz * cz);
  const float ang = atan2f(sn, dot);
  const float rv[3] = {divide_no_nan(cx * ang, sn), divide_no_nan(cy * ang, sn),
                       divide_no_nan(cz * ang, sn)};
  rotvec2mat(rv, m);
}

// Log map with the reference's 4-way quaternion branch; w may be negative (rotvec norm > pi),
// reproduced as is (rotation.py:261-289).
SF_HD void mat2rotvec(const float* r, float* rv) {
  const float r00 = r[0], r01 = r[1], r02 = r[2], r10 = r[3], r11 = r[4], r12 = r[5], r20 = r[6],
              r21 = r[7], r22 = r[8];
  const float trace = r00 + r11 + r22;
  float x, y, z, w;
  if (trace > 0.f) {
    x = r21 - r12; y = r02 - r20; z = r10 - r01; w = 1.0f + trace;
  } else if (r00 > r11 && r00 > r22) {
    x = (1.0f - r22) + (r00 - r11); y = r10 + r01; z = r02 + r20; w = r21 - r12;
  } else if (r11 > r22) {
    x = r10 + r01; y = (1.0f - r22) - (r00 - r11); z = r21 + r12; w = r02 - r20;
  } else {
    x = r02 + r20; y = r21 + r12; z = (1.0f + r22) - (r00 + r11); w = r10 - r01;
  }
  const float n = sqrtf(x * x + y * y + z * z);
  const float f = divide_no_nan(2.0f, n) * atan2f(n, w);
  rv[0] = f * x;
  rv[1] = f * y;
  rv[2] = f * z;
}

// Bone part: swing aligns the reference bone with the target bone, twist about the target bone is
// recovered in closed form from the part's centred cross-covariance A (bodyfitter.py:1389-1412).
SF_HD void swing_twist(const float* b_ref, const float* b_tgt, const float* A, float* R) {
  const float nr = sqrtf(b_ref[0] * b_ref[0] + b_ref[1] * b_ref[1] + b_ref[2] * b_ref[2]);
  const float nt = sqrtf(b_tgt[0] * b_tgt[0] + b_tgt[1] * b_tgt[1] + b_tgt[2] * b_tgt[2]);
  const float br[3] = {divide_no_nan(b_ref[0], nr), divide_no_nan(b_ref[1], nr),
                       divide_no_nan(b_ref[2], nr)};
  const float bt[3] = {divide_no_nan(b_tgt[0], nt), divide_no_nan(b_tgt[1], nt),
                       divide_no_nan(b_tgt[2], nt)};
  float Rsw[9], H[9];
  align_unit_vectors(br, bt, Rsw);
  m3_mult(Rsw, A, H);  // H = R_swing A^T
  const float trH = H[0] + H[4] + H[8];
  float Hb[3];
  m3_vec(H, bt, Hb);
  const float bHb = bt[0] * Hb[0] + bt[1] * Hb[1] + bt[2] * Hb[2];
  const float vee[3] = {H[5] - H[7], H[6] - H[2], H[1] - H[3]};
  const float ang = atan2f(bt[0] * vee[0] + bt[1] * vee[1] + bt[2] * vee[2], trH - bHb);
  const float rv[3] = {bt[0] * ang, bt[1] * ang, bt[2] * ang};
  float Rtw[9];
  rotvec2mat(rv, Rtw);
  m3_mul(Rtw, Rsw, R);
}

}  // namespace sf
