// Per-instance stages of the fit, written once for two executions:
//   * on the GPU a workgroup (one 64-lane wave for the joint-level stages) runs a stage
//     cooperatively: `cx.lane` / `cx.n` stride the loops, `cx.sync()` is the workgroup barrier and all
//     state that crosses a barrier lives in LDS;
//   * the host unit-test build (tests/hostemu) runs the very same code with n = 1, sync = no-op.
// Reference: src/smplfitter/pt/bodyfitter.py (line numbers cited per stage).
//
// Normal-equation record ("NE", S betas): [G upper triangle row-major : S(S+1)/2][r : S]
// [SA per coordinate : 3*S][Sb : 3][W : 1].
//
// Per-instance joint block ("jd") handed from the joint stage to the per-vertex kernels, per joint
// jd_stride(S) floats: [0..8] global rotation R (row-major), [9..11] T0 = P0 - R J0,
// [12 + c*jd_row(S) + s] T'[c][s] = d(P - R J)/d beta_s.  16-byte aligned rows -> ds_read_b128.
#pragma once
#include <stdint.h>

#include "sf_math.h"

namespace sf {

SF_HD constexpr int jd_row(int S) { return (S + 3) / 4 * 4; }
SF_HD constexpr int jd_stride(int S) {
  return (12 + 3 * jd_row(S)) % 8 == 4 ? 12 + 3 * jd_row(S) : 12 + 3 * jd_row(S) + 4;
}
SF_HD constexpr int ne_ng(int S) { return S * (S + 1) / 2; }
SF_HD constexpr int ne_size(int S) { return ne_ng(S) + S + 3 * S + 3; }  // W sits at [ne_size]
SF_HD constexpr int ne_g(int S, int i, int j) { return i * S - i * (i - 1) / 2 + (j - i); }

// floats per vertex of the packed constants record (must match HostTables::cstride())
SF_HD constexpr int cpack_stride(int S, int KW) {
  return ((3 * S + KW + KW / 4 + 3) / 4 * 4) % 8 == 4 ? (3 * S + KW + KW / 4 + 3) / 4 * 4
                                                      : (3 * S + KW + KW / 4 + 3) / 4 * 4 + 4;
}

// Position of pose-feature p inside the (Kp) GEMM K axis: parity-major, so that the fp32 MFMA
// 32x32x2 operand of lane (row, k parity) is one contiguous run (16-byte loads).  The posedirs rows
// (HostTables::pdT) are stored in the same order.
SF_HD constexpr int rp_pos(int p, int Kp) { return (p & 1) * (Kp / 2) + (p >> 1); }

constexpr int kPsum = 16;  // part-sum record: raw 9, s_t 3, s_a 3, s_w 1
constexpr int kScaleExtras = 6;  // scaled solve: entries after u in the extra-sum record (tt, tb, bb, St)

struct alignas(16) F4 {
  float x, y, z, w;
};
SF_HD F4 ld4(const float* p) { return *reinterpret_cast<const F4*>(p); }

// Tables a joint-level stage reads (device or host pointers).
struct JointTabs {
  int J, S, num_levels, adj_last_level, P, Kp;
  int n_kid;  // 1: the last of the S shape unknowns is the kid blend shape
  int n_pad;  // zero shape directions in front of it (HostTables::n_pad): pinned to 0 by a unit ridge
  const int32_t *parents, *fk_js, *fk_level_start, *cas_start, *cas_flat, *part_type, *toe_src;
  const int32_t *adj_level_start, *adj_parts;
  const float *j_ext, *bone_ext;  // (J,3,S+1)
  const int32_t* fk_jp;           // per position of fk_js: joint | parent << 16
  const float* bone_lv;           // (len(fk_js),3,S+1): bone_ext rows in fk_js order
  const float *cs_joint;  // (J,3,S) sum_v w_vj shapedirs_v   (closed-form vertex-block SA)
  const float *cw_joint;  // (J)     sum_v w_vj
  // pair-Gram constants (HostTables::pair_*, diag_*)
  int np;
  const int32_t* pair_j;
  const float *pair_c1, *pair_c2, *pair_c3, *diag_g0, *diag_c2, *diag_c3;
};

// ---------------------------------------------------------------------------------------------
// LDS carve for the joint-level stages (floats).
// ---------------------------------------------------------------------------------------------
struct JointScratch {
  float *tj, *rj;  // (J,3) centred target joints / reference joints
  float *R;        // (J,9) fitted part rotations, later new rotations in the refinement
  float *G;        // (J,9) global rotations
  float *P;        // (J,3,S+1) FK positions with beta-Jacobian          (joint stage only)
  float *T;        // (J,3,S+1) joint stage; (J,3) bones in the refinement
  float *aux;      // (J,3) joints from betas
  float *pos;      // (J,3)
};
// kind 0: joint stage / forward stage (P, T full).  kind 1: refinement (no P, T = (J,3)).
SF_HD int joint_scratch_floats(int J, int S, int kind) {
  const int S1 = S + 1;
  const int f = J * 3 * 2 + J * 9 * 2 + (kind == 0 ? J * 3 * S1 * 2 : J * 3) + J * 3 * 2;
  return (f + 3) / 4 * 4;
}
SF_HD JointScratch carve_joint_scratch(float* base, int J, int S, int kind) {
  const int S1 = S + 1;
  JointScratch s;
  s.tj = base;
  s.rj = s.tj + J * 3;
  s.R = s.rj + J * 3;
  s.G = s.R + J * 9;
  if (kind == 0) {
    s.P = s.G + J * 9;
    s.T = s.P + J * 3 * S1;
    s.aux = s.T + J * 3 * S1;
  } else {
    s.P = nullptr;
    s.T = s.G + J * 9;
    s.aux = s.T + J * 3;
  }
  s.pos = s.aux + J * 3;
  return s;
}
// solve stage scratch: (NE+1) + S*S + S + kSolveParts*S doubles, then S+3 floats
constexpr int kSolveParts = 6;  // partial sums of the T' part of Jac^T b (see solve_stage)
constexpr int kSolvePanel = 16;  // columns per panel of the blocked factorisation (general path)
SF_HD int solve_scratch_floats(int S) {
  return 2 * (ne_size(S) + 1 + S * S + S + kSolveParts * S) + ((S + 3 + 3) / 4 * 4);
}

#define SF_FOR(i, count) for (int i = cx.lane; i < (count); i += cx.n)
// timing experiments (tools/stage_stamps.sh, -DSMPLFIT_STAGE_STAMPS): cycle stamps at the sync points of the joint stages
#ifdef SMPLFIT_STAGE_STAMPS
#define SF_STAMP(k) cx.stamp(k)
#else
#define SF_STAMP(k)
#endif

// ---------------------------------------------------------------------------------------------
// Joint block of the normal equations (+ the closed-form vertex SA) with a lane per JOINT: every lane accumulates the
// terms of its joints' three rows in registers, then one wave sum per entry (cx.sum_to_last: the total on the last lane,
// which stores it).  The entry-per-lane form in joint_stage below walks the 3 J rows serially in every lane, once per
// branch the wave's lanes diverge into: 37 k of the 93 k cycles of k_joint_stage for SMPL, 78 k of 160 k for the
// SMPL-X-shaped model (tools/stage_stamps.sh); this form: see DESIGN.md.  Two phases (the Gramian, then everything
// else) keep the live accumulators under the registers of four waves per SIMD.  Same sums as the generic form; on the
// device the order of the additions differs (a tree over the lanes instead of one chain).
// ---------------------------------------------------------------------------------------------
template <int S, class Ctx>
SF_HD void joint_gram_by_joint(Ctx& cx, const JointTabs& tb, const JointScratch& sh, const float* jw, bool joint_block,
                               bool weighted, bool closed_form, float* gramj_out) {
  constexpr int S1 = S + 1, NG = ne_ng(S), NE = ne_size(S);
  const int J = tb.J;
  const bool last = cx.lane == cx.n - 1;
  {
    // row i of the Gramian is kept from column i & ~1 (an even column) to SE: the products of a row then pair up with
    // ALIGNED pairs of the joint row's values (packed fp32 without shuffled copies of them); the extra entries (column
    // i - 1 of the odd rows, the zero padding of an odd S) are computed and not used
    constexpr int SE = (S + 1) & ~1;
    float g[S][SE];
    SF_UNROLL_FULL
    for (int i = 0; i < S; ++i)
      SF_UNROLL_FULL
      for (int k = i & ~1; k < SE; ++k) g[i][k] = 0.f;
    if (joint_block) {
      for (int j = cx.lane; j < J; j += cx.n) {
        const float w = weighted ? jw[j] : 1.0f;
        SF_UNROLL_FULL
        for (int c = 0; c < 3; ++c) {
          const float* pr = sh.P + (j * 3 + c) * S1 + 1;
          float p[SE];
          SF_UNROLL_FULL
          for (int i = 0; i < SE; ++i) p[i] = i < S ? pr[i] : 0.f;
          SF_UNROLL_FULL
          for (int i = 0; i < S; ++i) {
            const float wp = w * p[i];
            SF_UNROLL_FULL
            for (int k = i & ~1; k < SE; ++k) g[i][k] += wp * p[k];
          }
          SF_SCHED_FENCE();  // (one row's reads at a time: registers)
        }
      }
    }
    SF_UNROLL_FULL
    for (int i = 0; i < S; ++i)
      SF_UNROLL_FULL
      for (int i2 = i; i2 < S; ++i2) {
        const float t = cx.sum_to_last(g[i][i2]);
        if (last) gramj_out[ne_g(S, i, i2)] = t;
      }
  }
  cx.sync();  // (also keeps the second phase's reads out of the first phase's registers)
  {
    float r[S], sa[3][S], sb[3] = {0.f, 0.f, 0.f}, W = 0.f;
    SF_UNROLL_FULL
    for (int i = 0; i < S; ++i) r[i] = sa[0][i] = sa[1][i] = sa[2][i] = 0.f;
    for (int j = cx.lane; j < J; j += cx.n) {
      const float w = weighted ? jw[j] : 1.0f;
      const float cw = closed_form ? tb.cw_joint[j] : 0.f;
      SF_UNROLL_FULL
      for (int c = 0; c < 3; ++c) {
        SF_SCHED_FENCE();
        if (joint_block) {
          const float* pr = sh.P + (j * 3 + c) * S1;
          const float d = sh.tj[j * 3 + c] - pr[0];
          SF_UNROLL_FULL
          for (int i = 0; i < S; ++i) {
            const float wp = w * pr[1 + i];
            r[i] += wp * d;
            sa[c][i] += wp;
          }
          sb[c] += d * w;
        }
        if (closed_form) {  // vertex-block SA with unit weights, sum_v Jac_v[c][i] = sum_j (G_j CS_j)[c][i] + cw_j T'_j[c][i]
          const float* Gj = sh.G + j * 9 + c * 3;
          const float* cs = tb.cs_joint + j * 3 * S;
          const float* tr = sh.T + (j * 3 + c) * S1 + 1;
          SF_UNROLL_FULL
          for (int i = 0; i < S; ++i) sa[c][i] += (Gj[0] * cs[i] + Gj[1] * cs[S + i] + Gj[2] * cs[2 * S + i]) + cw * tr[i];
        }
        SF_SCHED_FENCE();
      }
      if (joint_block) W += w;
    }
    SF_UNROLL_FULL
    for (int i = 0; i < S; ++i) {
      const float t = cx.sum_to_last(r[i]);
      if (last) gramj_out[NG + i] = t;
    }
    SF_UNROLL_FULL
    for (int c = 0; c < 3; ++c) {
      SF_UNROLL_FULL
      for (int i = 0; i < S; ++i) {
        const float t = cx.sum_to_last(sa[c][i]);
        if (last) gramj_out[NG + S + c * S + i] = t;
      }
      const float t = cx.sum_to_last(sb[c]);
      if (last) gramj_out[NG + 4 * S + c] = t;
    }
    const float t = cx.sum_to_last(W);
    if (last) gramj_out[NE] = t;
  }
}

// ---------------------------------------------------------------------------------------------
// Stage J — part rotations (+ optional shape-solve prologue).
//   rotations: _fit_global_rotations, bodyfitter.py:1321-1416 (buckets :81-97, toes :147-156)
//   prologue : _fit_shape, bodyfitter.py:869-916 (relative rotations, level-batched FK with the
//              beta-Jacobian, T = P - G J_ext, pose feature) and the joint block of the normal
//              equations, _gram_block :1598-1625 called at :1051-1053.
// psum (J,16): part sums of (target, reference) vertices.  Gprev == nullptr -> identity.
// ---------------------------------------------------------------------------------------------
template <class Ctx>
SF_HD void joint_stage(Ctx& cx, const JointTabs& tb, const JointScratch& sh, const float* psum,
                       const float* tjc, const float* rj_in, const float* Gprev, const float* jw,
                       bool fit_rotations, bool do_prologue, bool joint_block,
                       bool joint_block_weighted, bool vertex_sa_closed_form, float* Gout,
                       float* rp_out, float* jd_out, float* pext_out, float* gramj_out,
                       float* GT_out = nullptr, int gt_pitch = 0) {
  // GT_out (round 6): the instance's column of an instance-innermost copy of G — element (j, k) at
  // GT_out[(j * 9 + k) * gt_pitch] — for the batch-major prologue kernel k_prologue_bm, which then follows instead of
  // the prologue below (do_prologue = false)
  const int J = tb.J, S = tb.S, S1 = S + 1;
  SF_STAMP(0);
  SF_FOR(k, J * 3) {
    sh.tj[k] = tjc[k];
    sh.rj[k] = rj_in ? rj_in[k] : 0.f;
  }
  cx.sync();
  if (!fit_rotations) {  // rotations given by the caller (shape-solve entry point): G = Gprev
    SF_FOR(k, J * 9) {
      sh.G[k] = Gprev[k];
      Gout[k] = Gprev[k];
      if (GT_out) GT_out[k * gt_pitch] = Gprev[k];
    }
  }
  SF_FOR(j, J) {
    if (!fit_rotations) break;
    const int type = tb.part_type[j];
    float R[9];
    m3_identity(R);
    if (type != 0) {
      const int c0 = tb.cas_start[j], n = tb.cas_start[j + 1] - c0;
      // children-mean centres: rows of center_matrix hold 1/n (bodyfitter.py:124-129)
      const float inv = 1.0f / (float)n;
      float ct[3] = {0, 0, 0}, ca[3] = {0, 0, 0};
      for (int q = 0; q < n; ++q) {
        const int m = tb.cas_flat[c0 + q];
        for (int c = 0; c < 3; ++c) {
          ct[c] += inv * sh.tj[m * 3 + c];
          ca[c] += inv * sh.rj[m * 3 + c];
        }
      }
      float A[9];
      if (type == 1) {  // multi-joint part: Kabsch on its joints only (:1361-1383)
        float raw[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, st[3] = {0, 0, 0}, sa[3] = {0, 0, 0}, sw = 0;
        for (int q = 0; q < n; ++q) {
          const int m = tb.cas_flat[c0 + q];
          const float w = jw ? jw[m] : 1.0f;
          const float* t = sh.tj + m * 3;
          const float a[3] = {sh.rj[m * 3] * w, sh.rj[m * 3 + 1] * w, sh.rj[m * 3 + 2] * w};
          for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) raw[r * 3 + c] += t[r] * a[c];
            st[r] += t[r] * w;
            sa[r] += a[r];
          }
          sw += w;
        }
        centered_cov(raw, st, sa, sw, ct, ca, A);
      } else {
        const float* ps = psum + j * kPsum;
        centered_cov(ps, ps + 9, ps + 12, ps[15], ct, ca, A);
      }
      if (type == 2) {  // bone part: swing + twist (:1389-1412)
        const int k0 = tb.cas_flat[c0], k1 = tb.cas_flat[c0 + 1];
        float br[3], bt[3];
        for (int c = 0; c < 3; ++c) {
          br[c] = sh.rj[k1 * 3 + c] - sh.rj[k0 * 3 + c];
          bt[c] = sh.tj[k1 * 3 + c] - sh.tj[k0 * 3 + c];
        }
        swing_twist(br, bt, A, R);
      } else {  // multi-joint and leaf parts: one Kabsch projection (:1386-1387), single call site
        proj_so3(A, R);
      }
    }
    for (int k = 0; k < 9; ++k) sh.R[j * 9 + k] = R[k];
  }
  cx.sync();
  SF_STAMP(1);
  SF_FOR(j, J) {  // toes take the feet; compose with the previous rotations (:422-433)
    if (!fit_rotations) break;
    const int src = tb.toe_src[j] >= 0 ? tb.toe_src[j] : j;
    float G[9];
    if (Gprev) {
      m3_mul(sh.R + src * 9, Gprev + j * 9, G);
    } else {
      for (int k = 0; k < 9; ++k) G[k] = sh.R[src * 9 + k];
    }
    for (int k = 0; k < 9; ++k) {
      sh.G[j * 9 + k] = G[k];
      Gout[j * 9 + k] = G[k];
      if (GT_out) GT_out[(j * 9 + k) * gt_pitch] = G[k];
    }
  }
  cx.sync();
  SF_STAMP(2);
  if (!do_prologue) return;

  // relative rotations -> pose feature (:869-876, :913); root row of P (:889-891)
  SF_FOR(j, J) {
    if (j > 0) {
      float rel[9];
      m3_tmul(sh.G + tb.parents[j] * 9, sh.G + j * 9, rel);
      for (int k = 0; k < 9; ++k) rp_out[rp_pos((j - 1) * 9 + k, tb.Kp)] = rel[k];
    }
  }
  // first padding feature = 1: its posedirs row holds v_template (the GEMM's bias, see sf_tables.cpp)
  SF_FOR(k, tb.Kp - tb.P) rp_out[rp_pos(tb.P + k, tb.Kp)] = k == 0 ? 1.f : 0.f;
  SF_FOR(k, 3 * S1) sh.P[k] = tb.j_ext[k];
  cx.sync();
  SF_STAMP(3);
  // level-batched FK of positions and their beta-Jacobian (:892-907)
  for (int lv = 0; lv < tb.num_levels; ++lv) {
    const int l0 = tb.fk_level_start[lv], nl = tb.fk_level_start[lv + 1] - l0;
    SF_FOR(idx, nl * S1) {
      // (joint, parent) and the bone rows come from tables in level order: ONE round trip per level instead of the
      // chain fk_js -> parents -> bone_ext (cycle stamps: 1.7 k cycles per level, three dependent table reads)
      const int q = l0 + idx / S1, s = idx % S1;
      const int jp = tb.fk_jp[q], j = jp & 0xffff, p = jp >> 16;
      const float* Gp = sh.G + p * 9;
      const float* be = tb.bone_lv + q * 3 * S1;
      const float b0 = be[s], b1 = be[S1 + s], b2 = be[2 * S1 + s];
      for (int c = 0; c < 3; ++c)
        sh.P[(j * 3 + c) * S1 + s] =
            sh.P[(p * 3 + c) * S1 + s] + (Gp[c * 3] * b0 + Gp[c * 3 + 1] * b1 + Gp[c * 3 + 2] * b2);
    }
    cx.sync();
  }
  SF_STAMP(4);
  // T = P - G J_ext (:909-911); joint block for the vertex kernels; P for the solve stage
  const int stride = jd_stride(S), row = jd_row(S);
  // (the table values of a lane's NEXT item are requested before the current one is processed: the loop is a chain of
  // memory round trips otherwise, 2.4 k cycles per pass of the wave)
  const int cnt_js = J * S1;
  float n0 = 0.f, n1 = 0.f, n2 = 0.f;
  if (cx.lane < cnt_js) {
    const int j = cx.lane / S1, s = cx.lane % S1;
    const float* je = tb.j_ext + j * 3 * S1;
    n0 = je[s], n1 = je[S1 + s], n2 = je[2 * S1 + s];
  }
  SF_FOR(idx, cnt_js) {
    const int j = idx / S1, s = idx % S1;
    const float* Gj = sh.G + j * 9;
    const float e0 = n0, e1 = n1, e2 = n2;
    {
      const int nx = idx + cx.n < cnt_js ? idx + cx.n : cnt_js - 1;  // (clamped: no branch around the loads)
      const int jn = nx / S1, sn = nx % S1;
      const float* je = tb.j_ext + jn * 3 * S1;
      n0 = je[sn], n1 = je[S1 + sn], n2 = je[2 * S1 + sn];
    }
    for (int c = 0; c < 3; ++c) {
      const float p = sh.P[(j * 3 + c) * S1 + s];
      const float tv = p - (Gj[c * 3] * e0 + Gj[c * 3 + 1] * e1 + Gj[c * 3 + 2] * e2);
      sh.T[(j * 3 + c) * S1 + s] = tv;
      pext_out[(j * 3 + c) * S1 + s] = p;
      if (s == 0)
        jd_out[j * stride + 9 + c] = tv;
      else
        jd_out[j * stride + 12 + c * row + (s - 1)] = tv;
    }
  }
  SF_FOR(idx, J * 9) jd_out[(idx / 9) * stride + idx % 9] = sh.G[idx];
  SF_FOR(idx, J * 3 * (row - S)) {  // zero the row padding (read, never used, by the vertex kernel)
    const int j = idx / (3 * (row - S)), r = idx % (3 * (row - S));
    jd_out[j * stride + 12 + (r / (row - S)) * row + S + r % (row - S)] = 0.f;
  }
  // joint block of the normal equations, fp32 sums (:1051-1053, _gram_block :1598-1625)
  const int NG = ne_ng(S), NE = ne_size(S);
  cx.sync();
  SF_STAMP(5);
  if (S == 10 || S == 11) {  // (the shapes the batch-major path serves; more unknowns: the Gramian does not fit registers)
    if (S == 10)
      joint_gram_by_joint<10>(cx, tb, sh, jw, joint_block, joint_block_weighted, vertex_sa_closed_form, gramj_out);
    else
      joint_gram_by_joint<11>(cx, tb, sh, jw, joint_block, joint_block_weighted, vertex_sa_closed_form, gramj_out);
    SF_STAMP(6);
    return;
  }
  SF_FOR(e, NE + 1) {
    float acc = 0.f;
    if (vertex_sa_closed_form && e >= NG + S && e < NG + 4 * S) {
      // vertex-block SA with unit weights, sum_v Jac_v[c][i] = sum_j (G_j CS_j)[c][i] + cw_j T'_j[c][i]
      const int c = (e - NG - S) / S, i = (e - NG - S) % S;
      SF_UNROLL(4)
      for (int j = 0; j < J; ++j) {
        const float* Gj = sh.G + j * 9 + c * 3;
        const float* cs = tb.cs_joint + j * 3 * S + i;
        acc += (Gj[0] * cs[0] + Gj[1] * cs[S] + Gj[2] * cs[2 * S]) +
               tb.cw_joint[j] * sh.T[(j * 3 + c) * S1 + 1 + i];
      }
    }
    if (joint_block) {
      if (e < NG) {
        int i = 0, r = e;
        while (r >= S - i) {
          r -= S - i;
          ++i;
        }
        const int jj = i + r;
        SF_UNROLL(8)
        for (int q = 0; q < J * 3; ++q) {
          const float w = joint_block_weighted ? jw[q / 3] : 1.0f;
          acc += (w * sh.P[q * S1 + 1 + i]) * sh.P[q * S1 + 1 + jj];
        }
      } else if (e < NG + S) {
        const int i = e - NG;
        SF_UNROLL(8)
        for (int q = 0; q < J * 3; ++q) {
          const float w = joint_block_weighted ? jw[q / 3] : 1.0f;
          acc += (w * sh.P[q * S1 + 1 + i]) * (sh.tj[q] - sh.P[q * S1]);
        }
      } else if (e < NG + 4 * S) {
        const int c = (e - NG - S) / S, i = (e - NG - S) % S;
        SF_UNROLL(8)
        for (int j = 0; j < J; ++j) {
          const float w = joint_block_weighted ? jw[j] : 1.0f;
          acc += w * sh.P[(j * 3 + c) * S1 + 1 + i];
        }
      } else if (e < NE) {
        const int c = e - NG - 4 * S;
        for (int j = 0; j < J; ++j) {
          const float w = joint_block_weighted ? jw[j] : 1.0f;
          acc += (sh.tj[j * 3 + c] - sh.P[(j * 3 + c) * S1]) * w;
        }
      } else {
        if (joint_block_weighted)
          for (int j = 0; j < J; ++j) acc += jw[j];
        else
          acc = (float)J;
      }
    }
    gramj_out[e] = acc;
  }
  SF_STAMP(6);
}

// ---------------------------------------------------------------------------------------------
// Symmetric positive definite solve M x = b of the leading n x n block of M (lower triangle, row pitch ld), in place:
// x holds b on entry and the solution on exit.  rd: n doubles of scratch (1 / D_k).  panel: null, or
// kSolvePanel * n + kSolvePanel doubles of LDS for the blocked form (M in global memory: the general path).
// ---------------------------------------------------------------------------------------------
template <class Ctx>
SF_HD void ldlt_solve(Ctx& cx, double* M, int ld, int n, double* x, double* rd, double* panel) {
  // the reference's Cholesky solve (:1083-1084) as an in-place M = L D L^T (L unit lower: the same factorisation
  // without the square roots; L_ik D_k is left in M[i][k], D_k on the diagonal) — ONE sync per column, every lane
  // updating one entry of the trailing triangle, instead of three syncs and a lane-0 step per column — and
  // column-oriented substitutions spread over the lanes instead of two serial triangular loops on lane 0
  // (cycle stamps, S = 10: 26 k -> 8 k of the stage's 50 k cycles).  fp64 throughout, as before.
  // (rd may alias scratch the caller has finished reading: the cx.sync() that opens the first column orders those
  // reads before the first write of rd[0])
  // A pivot D_k <= 0 (a Gramian that is not positive definite: a singular system with both regularisers at 0) becomes
  // NaN, which reaches every unknown — the failure signal of the Cholesky factorisation this replaces (sqrt of a
  // negative number; the reference ignores cholesky_ex's info and returns what it gets, :1083).
  if (panel) {
    // Many unknowns: M (S x S doubles) is in global memory, and the column-by-column form below reads and writes the
    // whole trailing triangle once per column — S^3 / 6 x 16 bytes through L2 per instance (11.5 ms per launch at
    // S = 300, B = 256: half of such a fit).  Blocked, right-looking: a panel of kSolvePanel columns is factorised in LDS
    // (transposed: PT[q][r] = M[k0 + r][k0 + q], lanes along the rows), the forward substitution of its columns runs
    // on the copy in LDS, and the trailing triangle takes ONE rank-kSolvePanel update per panel.  Same arithmetic
    // (L_ik D_k left in M, 1 / D_k in rd; a non-positive pivot turns into NaN), the sums of a trailing entry grouped by panel.
    double* PT = panel;                        // [kSolvePanel][n]
    double* rdl = panel + kSolvePanel * n;     // [kSolvePanel] 1 / D of the panel's columns
    for (int k0 = 0; k0 < n; k0 += kSolvePanel) {
      const int wk = n - k0 < kSolvePanel ? n - k0 : kSolvePanel, rows = n - k0;
      cx.sync();  // (the previous panel's trailing update is complete)
      SF_FOR(idx, rows * wk) {
        const int r = idx / wk, q = idx - r * wk;
        PT[q * n + r] = r >= q ? M[(k0 + r) * ld + k0 + q] : 0.0;
      }
      for (int q = 0; q < wk; ++q) {
        cx.sync();
        const double dk = PT[q * n + q];
        const double rdk = dk > 0.0 ? 1.0 / dk : (dk - dk) / (dk - dk);
        if (cx.lane == 0) {
          rd[k0 + q] = rdk;
          rdl[q] = rdk;
        }
        const double yq = x[k0 + q] * rdk;  // forward substitution with column k0 + q (x[k0 + q] is final: every earlier column has been applied)
        const int nc = wk - 1 - q;          // the panel's columns to the right
        SF_FOR(idx, (rows - q - 1) * (nc + 1)) {
          const int r = q + 1 + idx / (nc + 1), c = idx % (nc + 1);
          if (c == nc) {
            x[k0 + r] -= PT[q * n + r] * yq;
          } else {
            const int q2 = q + 1 + c;
            if (r >= q2) PT[q2 * n + r] -= (PT[q * n + r] * rdk) * PT[q * n + q2];
          }
        }
      }
      cx.sync();
      SF_FOR(idx, rows * wk) {  // the factorised panel back to M (the back substitution reads it)
        const int r = idx / wk, q = idx - r * wk;
        if (r >= q) M[(k0 + r) * ld + k0 + q] = PT[q * n + r];
      }
      // trailing triangle: a group of (up to) 64 lanes takes a ROW a — its 16 scaled panel values stay in registers — and
      // the lanes the columns b2 <= a, four entries per lane requested together (one entry at a time every update
      // waited for its own round trip to M: 2.4 of the stage's 3.0 ms at S = 300)
      const int t0 = k0 + wk, nt = n - t0;
      const int gl = cx.n < 64 ? cx.n : 64, ng = cx.n / gl, g = cx.lane / gl, l = cx.lane - g * gl;
      for (int a = g; a < nt; a += ng) {
        double la[kSolvePanel];
        SF_UNROLL_FULL
        for (int q = 0; q < kSolvePanel; ++q) la[q] = q < wk ? PT[q * n + wk + a] * rdl[q] : 0.0;
        double* Mrow = M + (size_t)(t0 + a) * ld + t0;
        for (int b0 = l; b0 <= a; b0 += 4 * gl) {
          double acc[4];
          SF_UNROLL_FULL
          for (int u = 0; u < 4; ++u) acc[u] = b0 + u * gl <= a ? Mrow[b0 + u * gl] : 0.0;
          SF_UNROLL_FULL
          for (int q = 0; q < kSolvePanel; ++q)
            if (q < wk) {
              SF_UNROLL_FULL
              for (int u = 0; u < 4; ++u) {
                const int b2 = b0 + u * gl <= a ? b0 + u * gl : 0;
                acc[u] -= la[q] * PT[q * n + wk + b2];
              }
            }
          SF_UNROLL_FULL
          for (int u = 0; u < 4; ++u)
            if (b0 + u * gl <= a) Mrow[b0 + u * gl] = acc[u];
        }
      }
    }
    cx.sync();
    SF_STAMP(3);
  } else {
  for (int k = 0; k < n; ++k) {
    cx.sync();
    const double dk = M[k * ld + k];
    const double rdk = dk > 0.0 ? 1.0 / dk : (dk - dk) / (dk - dk);  // NaN for a non-positive pivot (0 / 0, also on a NaN pivot)
    if (cx.lane == 0) rd[k] = rdk;
    const int m = n - 1 - k;  // trailing rows i = k + 1 + a, columns j = k + 1 + b, b <= a
    SF_FOR(idx, m * m) {
      const int a = idx / m, b2 = idx % m;
      if (b2 <= a) {
        const int i2 = k + 1 + a, j2 = k + 1 + b2;
        M[i2 * ld + j2] -= (M[i2 * ld + k] * rdk) * M[j2 * ld + k];
      }
    }
  }
  SF_STAMP(3);
  for (int k = 0; k < n; ++k) {  // forward: y = L^-1 b
    cx.sync();
    const double yk = x[k] * rd[k];
    SF_FOR(i2, n) if (i2 > k) x[i2] -= M[i2 * ld + k] * yk;
  }
  }
  cx.sync();
  SF_FOR(i2, n) x[i2] *= rd[i2];  // z = D^-1 y
  for (int k = n - 1; k > 0; --k) {  // back: x = L^-T z
    cx.sync();
    const double xk = x[k];
    SF_FOR(i2, k) x[i2] -= (M[k * ld + i2] * rd[i2]) * xk;
  }
  cx.sync();
}

// ---------------------------------------------------------------------------------------------
// Stage S — combine the vertex and joint blocks in fp64, centre, regularise, Cholesky-solve,
// translation; joints and per-joint skinning translations at the solution.
//   _fit_shape_gram, bodyfitter.py:1054-1101.
// gramv: NE+1 doubles (vertex block incl. W), gramj: NE+1 floats (joint block incl. W).
// reg_ref (S) or null: values the ridge pulls towards (warm-started fit).
// mode 0: solve this instance.  share_beta (pt/lstsq.py:24-26): mode 1 stops after the regularised,
// centred system and writes it to cen[S*S + S] (lower triangle of M, then the right-hand side) for
// the sum over the batch; mode 2 solves the summed system read from cen (shared by all instances)
// and finishes with this instance's own translation.
// Outputs: beta (S), trans (3), rjoints (J,3), jb (J,4) = T0 + T' beta (skinning translation).
// ---------------------------------------------------------------------------------------------
// Ridge weight of shape unknown i (bodyfitter.py:1064-1071; kid unknown :1235-1242); padding unknowns: 1.
SF_HD double ridge_weight(const JointTabs& tb, int i, float beta_reg, float beta_reg2, float kid_reg) {
  const int S = tb.S;
  if (i >= S - tb.n_kid) return (double)kid_reg;
  if (i >= S - tb.n_kid - tb.n_pad) return 1.0;
  return (double)(i < 2 ? beta_reg2 : beta_reg);
}

template <class Ctx>
SF_HD void solve_stage(Ctx& cx, const JointTabs& tb, float* scratch, const double* gramv,
                       const float* gramj, const float* pext, const float* jd, const float* mb,
                       float beta_reg, float beta_reg2, float kid_reg, float* beta_out, float* trans_out,
                       float* rjoints_out, float* jb_out, const float* reg_ref = nullptr, int mode = 0,
                       double* cen = nullptr, float* jbT_out = nullptr, double* panel = nullptr) {
  // panel (general path: M lives in global memory): kSolvePanel * S + kSolvePanel doubles of LDS for the blocked
  // factorisation below
  // jbT_out: the instance's column of the instance-innermost copy of jb ((J, 4) rows of 64 instances; element
  // (j, c) at jbT_out[(j * 4 + c) * 64]) read by the batch-major LBS kernel
  const int J = tb.J, S = tb.S, S1 = S + 1;
  const int NG = ne_ng(S), NE = ne_size(S);
  double* sum = reinterpret_cast<double*>(scratch);  // NE+1   (scratch is 8-byte aligned)
  double* M = sum + NE + 1;                          // S*S (lower triangle used)
  double* x = M + S * S;                             // S
  double* r2p = x + S;                               // kSolveParts * S
  float* aux = reinterpret_cast<float*>(r2p + kSolveParts * S);  // S+3
  SF_STAMP(0);
  if (mb) {
    // pair-Gram form: the residual kernel delivers r1 = sum_v S_v^T (Rt_v^T b_v) and the per-joint residual moments
    // mb_j = sum_v w_vj b_v; the T' part of Jac^T b is sum_j T'_j^T mb_j.  kSolveParts lanes per unknown sum a run of
    // joints each (one dependent memory round trip per joint of the run instead of one per joint of the model: the
    // rows live in global memory), in a fixed order on every target.
    const int stride = jd_stride(S), row = jd_row(S), run = (J + kSolveParts - 1) / kSolveParts;
    SF_FOR(t, kSolveParts * S) {
      const int i = t % S, part = t / S;
      const int j1 = (part + 1) * run < J ? (part + 1) * run : J;
      double r2 = 0.0;
      for (int j = part * run; j < j1; ++j)
        for (int c = 0; c < 3; ++c) r2 += (double)jd[j * stride + 12 + c * row + i] * (double)mb[j * 3 + c];
      r2p[t] = r2;
    }
    cx.sync();
  }
  SF_FOR(e, NE + 1) {
    double v = gramv[e] + (double)gramj[e];
    if (mb && e >= NG && e < NG + S) {
      double r2 = 0.0;
      for (int part = 0; part < kSolveParts; ++part) r2 += r2p[part * S + (e - NG)];
      v += r2;
    }
    sum[e] = v;
  }
  cx.sync();
  SF_STAMP(1);
  double W = sum[NE];
  if (W == 0.0) W = 1.0;  // w_sum_safe (:1060)
  const double* SA = sum + NG + S;
  const double* Sb = sum + NG + 4 * S;
  SF_FOR(idx, S * S) {
    const int i = idx / S, j = idx % S;
    if (j <= i) {  // lower triangle: M[i][j] = G[j][i] - sum_c SA[c][i] SA[c][j] / W (+ lambda)
      double g = sum[ne_g(S, j, i)];
      g -= (SA[i] * SA[j] + SA[S + i] * SA[S + j] + SA[2 * S + i] * SA[2 * S + j]) / W;
      if (i == j)  // (:1064-1071; kid unknown :1235-1242)
        g += ridge_weight(tb, i, beta_reg, beta_reg2, kid_reg);
      M[i * S + j] = g;
    }
  }
  SF_FOR(i, S) {
    double r = sum[NG + i] - (SA[i] * Sb[0] + SA[S + i] * Sb[1] + SA[2 * S + i] * Sb[2]) / W;
    if (reg_ref)  // ridge towards reference values: + lambda_i ref_i (:1072-1081, :1224-1255)
      r += ridge_weight(tb, i, beta_reg, beta_reg2, kid_reg) * (double)reg_ref[i];
    x[i] = r;
  }
  cx.sync();
  if (mode == 1) {
    SF_FOR(k, S * S) cen[k] = (k % S <= k / S) ? M[k] : 0.0;
    SF_FOR(i, S) cen[S * S + i] = x[i];
    return;
  }
  if (mode == 2) {
    SF_FOR(k, S * S) M[k] = cen[k];
    SF_FOR(i, S) x[i] = cen[S * S + i];
    cx.sync();
  }
  SF_STAMP(2);
  // rd ALIASES r2p: the partial sums of r2 were read by the sum loop above, and the cx.sync() that opens the
  // factorisation is what orders those reads before the first write of rd[0].
  ldlt_solve(cx, M, S, S, x, r2p, panel);  // (rd ALIASES r2p: see below)
  SF_STAMP(4);
  // translation (:1086-1088) and outputs, cast to fp32 (:1088-1089)
  float* betaf = aux;       // S
  float* transf = aux + S;  // 3
  SF_FOR(i, S) {
    betaf[i] = (float)x[i];
    beta_out[i] = (float)x[i];
  }
  SF_FOR(c, 3) {
    double v = Sb[c] / W;
    for (int i = 0; i < S; ++i) v -= (SA[c * S + i] / W) * x[i];
    transf[c] = (float)v;
    trans_out[c] = (float)v;
  }
  cx.sync();
  const int stride = jd_stride(S), row = jd_row(S);
  SF_FOR(idx, J * 3) {
    const int j = idx / 3, c = idx % 3;
    // joints = P0 + P' beta + trans (:1093-1098)
    float acc = 0.f;
    for (int s = 0; s < S; ++s) acc += pext[idx * S1 + 1 + s] * betaf[s];
    rjoints_out[idx] = pext[idx * S1] + acc + transf[c];
    // skinning translation at the solution, T0 + T' beta (vertex re-evaluation :1099-1101)
    float tb0 = 0.f;
    const float* tr = jd + j * stride + 12 + c * row;
    for (int s = 0; s < S; ++s) tb0 += tr[s] * betaf[s];
    jb_out[j * 4 + c] = jd[j * stride + 9 + c] + tb0;
    if (jbT_out) jbT_out[(j * 4 + c) * 64] = jd[j * stride + 9 + c] + tb0;
  }
  SF_STAMP(5);
}

// ---------------------------------------------------------------------------------------------
// Stage S' — the LAST shape solve of fit(scale_target=True) / fit(scale_fit=True): one more unknown
// sigma = scale - 1 (_fit_shape_general, bodyfitter.py:1104-1319; driver :434-455).  The (S+1)-unknown
// normal equations are assembled from the ordinary record (gramv + gramj, as in stage S) and the extra
// sums of the vertices (vextra: [u : S][tt][tb][bb][St : 3], scale_extras_vertex) and of the joints
// (computed here from P and the target joints):
//   scale_target (mode 1), c = -t:      g = -u,  h = tt,  q = -tb,  Sc = -St
//   scale_fit    (mode 2), c = t - b:   g = u - r,  h = tt - 2 tb + bb,  q = tb - bb,  Sc = St - Sb
// scratch: doubles [NE+1 | (S+1)^2 | S+1 | S | 8 | S+1] then floats.  Outputs: beta_out (S, UNDIVIDED — what the
// reference returns and hands to the refinement, :1277-1283), beta_eval (S, divided by the scale for
// scale_fit: the shape the mesh is evaluated at, :1289-1293), trans, scale, joints, jb.
// share_beta with a scale unknown (lstsq_partial_share with n_shared = S, pt/lstsq.py:50-90: the shape
// is shared by the batch, every instance keeps its scale): regressing out the independent column is
// the Schur complement of the scale entry.  share 1 writes this instance's reduced S x S system
// M_ss - m m^T / c and right-hand side r_s - m rho / c to cen[S*S + S] (layout of stage S) for the sum
// over the batch; share 2 solves the summed system read from cen and recovers this instance's
// sigma = (rho - m . x_s) / c.  In this branch the reference appends the ridge as ROWS of the design
// matrix with weight lambda and right-hand side lambda * reference (:52-61), so reg_ref enters with
// lambda^2 — restated as the reference computes it.
// ---------------------------------------------------------------------------------------------
// (general path: one scratch slot per instance serves both solve stages)
SF_HD int gen_solve_scratch_floats(int S);
SF_HD int scaled_solve_scratch_floats(int S) {
  return 2 * (ne_size(S) + 1 + (S + 1) * (S + 1) + (S + 1) + S + 8 + (S + 1)) + ((S + 8) / 4 * 4);
}

SF_HD int gen_solve_scratch_floats(int S) {
  const int a = solve_scratch_floats(S), b = scaled_solve_scratch_floats(S);
  return a > b ? a : b;
}

template <class Ctx>
SF_HD void scaled_solve_stage(Ctx& cx, const JointTabs& tb, float* scratch, const double* gramv,
                              const float* gramj, const float* vextra, const float* pext, const float* jd,
                              const float* mb, const float* tj, const float* jw, bool joint_block,
                              int mode, float beta_reg, float beta_reg2, float kid_reg, float scale_reg,
                              const float* reg_ref, float* beta_out, float* beta_eval, float* trans_out,
                              float* scale_out, float* rjoints_out, float* jb_out, int share = 0,
                              double* cen = nullptr, double* panel = nullptr, const double* vextra_d = nullptr) {
  // vextra_d: the vertices' extra sums as doubles (general path: k_gen_accum_mfma) instead of vextra
  // panel (general path: the scratch, M included, is in global memory): LDS for ldlt_solve's blocked form — the
  // (S + 1)-unknown system is then solved as L D L^T with column-oriented substitutions instead of the Cholesky
  // factorisation and the two serial triangular loops on lane 0 below (S^2 dependent round trips to global memory)
  const int J = tb.J, S = tb.S, S1 = S + 1, N = S + 1;
  const int NG = ne_ng(S), NE = ne_size(S);
  double* sum = reinterpret_cast<double*>(scratch);  // NE+1
  double* M = sum + NE + 1;                          // N*N (lower triangle)
  double* x = M + N * N;                             // N
  double* u = x + N;                                 // S: sum w Jac^T t, vertices + joints
  double* ex = u + S;                                // [tt, tb, bb, St(3)] vertices + joints
  double* rdv = ex + 8;                              // N: 1 / D_k of ldlt_solve (panel form)
  float* aux = reinterpret_cast<float*>(rdv + N);    // S+4 floats
  SF_FOR(e, NE + 1) {
    double v = gramv[e] + (double)gramj[e];
    if (mb && e >= NG && e < NG + S) {  // pair-Gram form of the vertex block (see stage S)
      const int i = e - NG, stride = jd_stride(S), row = jd_row(S);
      double r2 = 0.0;
      for (int j = 0; j < J; ++j)
        for (int c = 0; c < 3; ++c) r2 += (double)jd[j * stride + 12 + c * row + i] * (double)mb[j * 3 + c];
      v += r2;
    }
    sum[e] = v;
  }
  // the extra sums: vertices + joints
  SF_FOR(i, S + kScaleExtras) {
    double v = vextra_d ? vextra_d[i] : (double)vextra[i];
    if (joint_block) {
      float acc = 0.f;
      for (int j = 0; j < J; ++j) {
        const float w = jw ? jw[j] : 1.0f;
        for (int c = 0; c < 3; ++c) {
          const float t = tj[j * 3 + c], p0 = pext[(j * 3 + c) * S1], b = t - p0;
          float term;
          if (i < S) term = pext[(j * 3 + c) * S1 + 1 + i] * t;
          else if (i == S) term = t * t;
          else if (i == S + 1) term = t * b;
          else if (i == S + 2) term = b * b;
          else term = (c == i - S - 3) ? t : 0.f;
          acc += w * term;
        }
      }
      v += (double)acc;
    }
    if (i < S) u[i] = v; else ex[i - S] = v;
  }
  cx.sync();
  double W = sum[NE];
  if (W == 0.0) W = 1.0;
  const double* SA = sum + NG + S;
  const double* Sb = sum + NG + 4 * S;
  const double tt = ex[0], tbv = ex[1], bb = ex[2];
  const double h = mode == 1 ? tt : tt - 2.0 * tbv + bb;
  const double q = mode == 1 ? -tbv : tbv - bb;
  double Sc[3];
  for (int c = 0; c < 3; ++c) Sc[c] = mode == 1 ? -ex[3 + c] : ex[3 + c] - Sb[c];
  auto lam = [&](int i) -> double {
    return i == S ? (double)scale_reg : ridge_weight(tb, i, beta_reg, beta_reg2, kid_reg);
  };
  auto gcol = [&](int i) -> double { return mode == 1 ? -u[i] : u[i] - sum[NG + i]; };  // g[i]
  SF_FOR(idx, N * N) {
    const int i = idx / N, j = idx % N;
    if (j <= i) {
      double g;
      if (i < S) {
        g = sum[ne_g(S, j, i)] - (SA[i] * SA[j] + SA[S + i] * SA[S + j] + SA[2 * S + i] * SA[2 * S + j]) / W;
      } else if (j < S) {
        g = gcol(j) - (Sc[0] * SA[j] + Sc[1] * SA[S + j] + Sc[2] * SA[2 * S + j]) / W;
      } else {
        g = h - (Sc[0] * Sc[0] + Sc[1] * Sc[1] + Sc[2] * Sc[2]) / W;
      }
      if (i == j) g += lam(i);
      M[i * N + j] = g;
    }
  }
  SF_FOR(i, N) {
    double r;
    if (i < S) {
      r = sum[NG + i] - (SA[i] * Sb[0] + SA[S + i] * Sb[1] + SA[2 * S + i] * Sb[2]) / W;
      if (reg_ref) r += (share ? lam(i) * lam(i) : lam(i)) * (double)reg_ref[i];
    } else {
      r = q - (Sc[0] * Sb[0] + Sc[1] * Sb[1] + Sc[2] * Sb[2]) / W;
    }
    x[i] = r;
  }
  cx.sync();
  if (share == 1) {  // Schur complement of the scale entry: row S of M is (m, c), x[S] is rho
    const double c = M[S * N + S], rho = x[S];
    SF_FOR(k, S * S) {
      const int i = k / S, j = k % S;
      cen[k] = j <= i ? M[i * N + j] - M[S * N + i] * M[S * N + j] / c : 0.0;
    }
    SF_FOR(i, S) cen[S * S + i] = x[i] - M[S * N + i] * rho / c;
    return;
  }
  int n = N;  // size of the system solved here
  if (share == 2) {  // the summed reduced system; row S of M and x[S] stay this instance's own
    SF_FOR(k, S * S) {
      const int i = k / S, j = k % S;
      if (j <= i) M[i * N + j] = cen[k];
    }
    SF_FOR(i, S) x[i] = cen[S * S + i];
    n = S;
    cx.sync();
  }
  if (panel) {
    ldlt_solve(cx, M, N, n, x, rdv, panel);
    if (share == 2) {  // this instance's scale from the shared shape (row S of M and x[S] are untouched: n == S)
      cx.sync();
      if (cx.lane == 0) {
        double v = x[S];
        for (int j = 0; j < S; ++j) v -= M[S * N + j] * x[j];
        x[S] = v / M[S * N + S];
      }
    }
    cx.sync();
  } else {
    for (int k = 0; k < n; ++k) {  // Cholesky, column by column
      if (cx.lane == 0) M[k * N + k] = sqrt(M[k * N + k]);
      cx.sync();
      SF_FOR(i, n) if (i > k) M[i * N + k] /= M[k * N + k];
      cx.sync();
      SF_FOR(idx, n * n) {
        const int i = idx / n, j = idx % n;
        if (j > k && i >= j) M[i * N + j] -= M[i * N + k] * M[j * N + k];
      }
      cx.sync();
    }
    if (cx.lane == 0) {
      for (int i = 0; i < n; ++i) {
        double v = x[i];
        for (int k = 0; k < i; ++k) v -= M[i * N + k] * x[k];
        x[i] = v / M[i * N + i];
      }
      for (int i = n - 1; i >= 0; --i) {
        double v = x[i];
        for (int k = i + 1; k < n; ++k) v -= M[k * N + i] * x[k];
        x[i] = v / M[i * N + i];
      }
      if (share == 2) {  // this instance's scale from the shared shape
        double v = x[S];
        for (int j = 0; j < S; ++j) v -= M[S * N + j] * x[j];
        x[S] = v / M[S * N + S];
      }
    }
    cx.sync();
  }
  float* betaf = aux;            // S: the shape the mesh is evaluated at
  float* transf = aux + S;       // 3
  const float scale = (float)x[S] + 1.0f;  // new_scale_corr (:1286)
  SF_FOR(i, S) {
    const float bu = (float)x[i];
    beta_out[i] = bu;
    const float be = mode == 2 ? bu / scale : bu;  // (:1289-1293)
    betaf[i] = be;
    beta_eval[i] = be;
  }
  SF_FOR(c, 3) {  // translation from ALL unknowns, sigma included (:1270-1272)
    double v = Sb[c] / W;
    for (int i = 0; i < S; ++i) v -= (SA[c * S + i] / W) * x[i];
    v -= (Sc[c] / W) * x[S];
    transf[c] = (float)v;
    trans_out[c] = (float)v;
  }
  if (cx.lane == 0) scale_out[0] = scale;
  cx.sync();
  const int stride = jd_stride(S), row = jd_row(S);
  SF_FOR(idx, J * 3) {
    const int j = idx / 3, c = idx % 3;
    float acc = 0.f;
    for (int s = 0; s < S; ++s) acc += pext[idx * S1 + 1 + s] * betaf[s];
    rjoints_out[idx] = pext[idx * S1] + acc + transf[c];
    float tb0 = 0.f;
    const float* tr = jd + j * stride + 12 + c * row;
    for (int s = 0; s < S; ++s) tb0 += tr[s] * betaf[s];
    jb_out[j * 4 + c] = jd[j * stride + 9 + c] + tb0;
  }
}

// ---------------------------------------------------------------------------------------------
// Stage R — dependent rotation refinement, level-batched branch (bodyfitter.py:1418-1544) and the
// epilogue (:513-539): trans += mean, relative rotations, mat2rotvec.
//   psum: part sums of (target, final reference vertices); rj_true: joints of the final solve.
//   rj_reg/tj: joints entering the joint term (equal to rj_true / target joints when joints are
//   given; regressed ones otherwise, :1433-1438).
// ---------------------------------------------------------------------------------------------
template <class Ctx>
SF_HD void refine_stage(Ctx& cx, const JointTabs& tb, const JointScratch& sh, const float* psum,
                        const float* tj_in, const float* rj_joint_term, const float* rj_true,
                        const float* jw, const float* Gprev, const float* beta, const float* trans,
                        const float* mean, bool final_adjust, float* pose_out, float* beta_out,
                        float* trans_out, float* kid_out, float* orient_out, float* rel_out,
                        const float* scale = nullptr) {
  const int J = tb.J, S = tb.S, S1 = S + 1;
  SF_FOR(k, J * 9) {
    sh.G[k] = Gprev[k];
    sh.R[k] = Gprev[k];
  }
  if (final_adjust) {
    SF_FOR(k, J * 3) {
      sh.tj[k] = tj_in[k];
      sh.rj[k] = rj_joint_term[k];
      // joints from betas (:1441-1445)
      float acc = 0.f;
      for (int s = 0; s < S; ++s) acc += tb.j_ext[k * S1 + 1 + s] * beta[s];
      sh.aux[k] = tb.j_ext[k * S1] + acc;
      if (scale) sh.aux[k] *= scale[0];  // scale_corr of the known-shape fit (:1449-1450)
    }
    cx.sync();
    SF_FOR(k, J * 3) {  // bones (:1452-1460) into T (scratch), root position (:1481)
      const int j = k / 3, c = k % 3;
      sh.T[k] = sh.aux[k] - (j > 0 ? sh.aux[tb.parents[j] * 3 + c] : 0.f);
      if (j == 0) sh.pos[k] = sh.aux[k] + trans[c];
    }
    cx.sync();
    for (int lv = 0; lv <= tb.adj_last_level; ++lv) {
      const int l0 = tb.fk_level_start[lv], nl = tb.fk_level_start[lv + 1] - l0;
      SF_FOR(q, nl) {  // FK of this level from final parents (:1491-1498)
        const int j = tb.fk_js[l0 + q], p = tb.parents[j];
        float rb[3];
        m3_vec(sh.R + p * 9, sh.T + j * 3, rb);
        for (int c = 0; c < 3; ++c) sh.pos[j * 3 + c] = sh.pos[p * 3 + c] + rb[c];
      }
      cx.sync();
      const int a0 = tb.adj_level_start[lv], na = tb.adj_level_start[lv + 1] - a0;
      SF_FOR(q, na) {  // refine the adjustable parts of the level (:1505-1537)
        const int i = tb.adj_parts[a0 + q];
        const float* ct = sh.pos + i * 3;
        const float* ca = rj_true + i * 3;
        const float* ps = psum + i * kPsum;
        float A[9];
        centered_cov(ps, ps + 9, ps + 12, ps[15], ct, ca, A);
        for (int m0 = tb.cas_start[i]; m0 < tb.cas_start[i + 1]; ++m0) {
          const int m = tb.cas_flat[m0];
          const float w = jw ? jw[m] : 1.0f;
          float est[3], dfl[3];
          for (int c = 0; c < 3; ++c) {
            est[c] = sh.tj[m * 3 + c] - ct[c];
            dfl[c] = (sh.rj[m * 3 + c] - ca[c]) * w;
          }
          for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) A[r * 3 + c] += est[r] * dfl[c];
        }
        float Rf[9], Rn[9];
        proj_so3(A, Rf);
        m3_mul(Rf, sh.G + i * 9, Rn);
        for (int k = 0; k < 9; ++k) sh.R[i * 9 + k] = Rn[k];
      }
      cx.sync();
    }
    SF_FOR(j, J) {  // toes copy the feet (:1539-1543)
      if (tb.toe_src[j] >= 0)
        for (int k = 0; k < 9; ++k) sh.R[j * 9 + k] = sh.R[tb.toe_src[j] * 9 + k];
    }
  }
  cx.sync();
  SF_FOR(j, J) {  // relative rotations and log map (:523-539)
    float rel[9];
    if (j > 0) {
      m3_tmul(sh.R + tb.parents[j] * 9, sh.R + j * 9, rel);
    } else {
      for (int k = 0; k < 9; ++k) rel[k] = sh.R[k];
    }
    float rv[3];
    mat2rotvec(rel, rv);
    for (int c = 0; c < 3; ++c) pose_out[j * 3 + c] = rv[c];
    if (orient_out)
      for (int k = 0; k < 9; ++k) orient_out[j * 9 + k] = sh.R[j * 9 + k];
    if (rel_out)
      for (int k = 0; k < 9; ++k) rel_out[j * 9 + k] = rel[k];
  }
  if (beta_out) SF_FOR(i, S - tb.n_kid - tb.n_pad) beta_out[i] = beta[i];
  if (tb.n_kid && kid_out && cx.lane == 0) kid_out[0] = beta[S - 1];
  SF_FOR(c, 3) trans_out[c] = trans[c] + mean[c];  // (:519)
}

// ---------------------------------------------------------------------------------------------
// Stage A — scale and translation between the target and the posed reference
//   fit_scale_and_translation, bodyfitter.py:1628-1681, as called by fit_with_known_shape (:764-772),
//   followed by the substitution reference <- scale * reference + trans that the refinement is
//   handed (:774-802), applied to the part sums and joints instead of re-skinning the mesh:
//     raw' = s raw + s_t trans^T,  s_a' = s s_a + s_w trans,  joints' = s joints + trans.
// Points: the V real vertices (SoA slots [0,V) of tvs / rverts, weights vws or null) and, when tj is
// given, the J joints (weights jw or null).  The reference's per-instance scale (its (B,) * (B,3)
// broadcast at :1675-1676 only runs for B == 1) is applied per instance.
// red: cx.n * 8 floats of scratch.  rj_reg (regressed reference joints) may be null.
// ---------------------------------------------------------------------------------------------
template <class Ctx>
SF_HD void scale_trans_stage(Ctx& cx, int J, int V, int Vp, float* red, const float* tvs,
                             const float* rverts, const float* vws, const float* tj, float* rj,
                             const float* jw, bool with_scale, float* psum, float* rj_reg,
                             const float* reg_rowsum, float* trans_out, float* scale_out) {
  // pass 1: weighted sums  [W, t(3), r(3)]
  float a[7] = {0, 0, 0, 0, 0, 0, 0};
  SF_FOR(i, V) {
    const float w = vws ? vws[i] : 1.0f;
    a[0] += w;
    for (int c = 0; c < 3; ++c) {
      a[1 + c] += w * tvs[c * Vp + i];
      a[4 + c] += w * rverts[c * Vp + i];
    }
  }
  if (tj) SF_FOR(j, J) {
    const float w = jw ? jw[j] : 1.0f;
    a[0] += w;
    for (int c = 0; c < 3; ++c) {
      a[1 + c] += w * tj[j * 3 + c];
      a[4 + c] += w * rj[j * 3 + c];
    }
  }
  for (int k = 0; k < 7; ++k) red[cx.lane * 8 + k] = a[k];
  cx.sync();
  float tot[7];
  for (int k = 0; k < 7; ++k) {  // every lane sums the partials in the same order
    float s = 0.f;
    for (int l = 0; l < cx.n; ++l) s += red[l * 8 + k];
    tot[k] = s;
  }
  cx.sync();
  const float invW = 1.0f / tot[0];
  const float mt[3] = {tot[1] * invW, tot[2] * invW, tot[3] * invW};
  const float mr[3] = {tot[4] * invW, tot[5] * invW, tot[6] * invW};
  float sc = 1.0f;
  if (with_scale) {  // pass 2: centred second moments (:1665-1672)
    float q[2] = {0, 0};
    SF_FOR(i, V) {
      const float w = vws ? vws[i] : 1.0f;
      for (int c = 0; c < 3; ++c) {
        const float dt = tvs[c * Vp + i] - mt[c], dr = rverts[c * Vp + i] - mr[c];
        q[0] += w * dt * dt;
        q[1] += w * dr * dr;
      }
    }
    if (tj) SF_FOR(j, J) {
      const float w = jw ? jw[j] : 1.0f;
      for (int c = 0; c < 3; ++c) {
        const float dt = tj[j * 3 + c] - mt[c], dr = rj[j * 3 + c] - mr[c];
        q[0] += w * dt * dt;
        q[1] += w * dr * dr;
      }
    }
    red[cx.lane * 8] = q[0];
    red[cx.lane * 8 + 1] = q[1];
    cx.sync();
    float st = 0.f, sr = 0.f;
    for (int l = 0; l < cx.n; ++l) {
      st += red[l * 8];
      sr += red[l * 8 + 1];
    }
    cx.sync();
    sc = sqrtf(st / sr);
  }
  const float tr[3] = {mt[0] - sc * mr[0], mt[1] - sc * mr[1], mt[2] - sc * mr[2]};
  SF_FOR(j, J) {
    float* ps = psum + j * kPsum;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) ps[r * 3 + c] = sc * ps[r * 3 + c] + ps[9 + r] * tr[c];
    for (int c = 0; c < 3; ++c) {
      ps[12 + c] = sc * ps[12 + c] + ps[15] * tr[c];
      rj[j * 3 + c] = sc * rj[j * 3 + c] + tr[c];
      if (rj_reg) rj_reg[j * 3 + c] = sc * rj_reg[j * 3 + c] + reg_rowsum[j] * tr[c];
    }
  }
  SF_FOR(c, 3) trans_out[c] = tr[c];
  if (cx.lane == 0 && scale_out) scale_out[0] = sc;
}

// ---------------------------------------------------------------------------------------------
// Stage F — forward-kinematics prologue of BodyModel.forward (bodymodel.py:216-284): rotations from
// rotvecs (or given global rotations), joints from betas, FK, pose feature and per-joint skinning
// translations T = pos - G j (+ trans folded in by the vertex kernel).
// ---------------------------------------------------------------------------------------------
template <class Ctx>
SF_HD void forward_joint_stage(Ctx& cx, const JointTabs& tb, const JointScratch& sh,
                               const float* pose_rotvecs, const float* glob_in, const float* betas,
                               int nb, const float* kid, const float* trans, float* rp_out, float* jd_out,
                               float* joints_out, float* orient_out, const float* rel_in = nullptr) {
  // rel_in (J,3,3): relative rotation matrices given directly (forward's rel_rotmats, bodymodel.py:230-234) — the
  // same kinematic chain as pose_rotvecs without the exponential map
  const int J = tb.J, S = tb.S, S1 = S + 1;
  if (glob_in) {
    SF_FOR(k, J * 9) sh.G[k] = glob_in[k];
  } else {
    SF_FOR(j, J) {
      float m[9];
      if (rel_in) {
        for (int k = 0; k < 9; ++k) m[k] = rel_in[j * 9 + k];
      } else if (pose_rotvecs) {
        rotvec2mat(pose_rotvecs + j * 3, m);
      } else {
        m3_identity(m);
      }
      for (int k = 0; k < 9; ++k) sh.R[j * 9 + k] = m[k];
    }
    cx.sync();
    if (cx.lane == 0)
      for (int k = 0; k < 9; ++k) sh.G[k] = sh.R[k];
    cx.sync();
    // sequential FK of rotations (bodymodel.py:230-234), one level at a time
    for (int lv = 0; lv < tb.num_levels; ++lv) {
      const int l0 = tb.fk_level_start[lv], nl = tb.fk_level_start[lv + 1] - l0;
      SF_FOR(q, nl) {
        const int j = tb.fk_js[l0 + q];
        float g[9];
        m3_mul(sh.G + tb.parents[j] * 9, sh.R + j * 9, g);
        for (int k = 0; k < 9; ++k) sh.G[j * 9 + k] = g[k];
      }
      cx.sync();
    }
  }
  cx.sync();
  SF_FOR(k, J * 3) {  // joints from betas (bodymodel.py:258-264)
    float acc = 0.f;
    for (int s = 0; s < nb; ++s) acc += tb.j_ext[k * S1 + 1 + s] * betas[s];
    if (kid && tb.n_kid) acc += tb.j_ext[k * S1 + S] * kid[0];  // kid_J_shapedir (bodymodel.py:263)
    sh.aux[k] = tb.j_ext[k * S1] + acc;
  }
  SF_FOR(j, J) {  // pose feature (bodymodel.py:244-251, :286)
    if (j > 0) {
      float rel[9];
      if (glob_in) {
        m3_tmul(sh.G + tb.parents[j] * 9, sh.G + j * 9, rel);
      } else {
        for (int k = 0; k < 9; ++k) rel[k] = sh.R[j * 9 + k];
      }
      for (int k = 0; k < 9; ++k) rp_out[rp_pos((j - 1) * 9 + k, tb.Kp)] = rel[k];
    }
  }
  // first padding feature = 1: its posedirs row holds v_template (the GEMM's bias, see sf_tables.cpp)
  SF_FOR(k, tb.Kp - tb.P) rp_out[rp_pos(tb.P + k, tb.Kp)] = k == 0 ? 1.f : 0.f;
  cx.sync();
  SF_FOR(c, 3) sh.pos[c] = sh.aux[c];
  cx.sync();
  for (int lv = 0; lv < tb.num_levels; ++lv) {  // FK of positions (bodymodel.py:266-275)
    const int l0 = tb.fk_level_start[lv], nl = tb.fk_level_start[lv + 1] - l0;
    SF_FOR(q, nl) {
      const int j = tb.fk_js[l0 + q], p = tb.parents[j];
      const float bone[3] = {sh.aux[j * 3] - sh.aux[p * 3], sh.aux[j * 3 + 1] - sh.aux[p * 3 + 1],
                             sh.aux[j * 3 + 2] - sh.aux[p * 3 + 2]};
      float rb[3];
      m3_vec(sh.G + p * 9, bone, rb);
      for (int c = 0; c < 3; ++c) sh.pos[j * 3 + c] = sh.pos[p * 3 + c] + rb[c];
    }
    cx.sync();
  }
  const int stride = jd_stride(S);
  SF_FOR(j, J) {  // translations = pos - G j (bodymodel.py:297)
    float gj[3];
    m3_vec(sh.G + j * 9, sh.aux + j * 3, gj);
    for (int c = 0; c < 3; ++c) {
      jd_out[j * stride + 9 + c] = sh.pos[j * 3 + c] - gj[c];
      joints_out[j * 3 + c] = sh.pos[j * 3 + c] + (trans ? trans[c] : 0.f);
    }
    for (int k = 0; k < 9; ++k) {
      jd_out[j * stride + k] = sh.G[j * 9 + k];
      if (orient_out) orient_out[j * 9 + k] = sh.G[j * 9 + k];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Per-vertex bodies (lane = vertex).  jd = the instance's joint block (LDS on the GPU).
// rec = the vertex's packed constants record (HostTables::cpackA/B, LDS on the GPU), cpack_stride
// floats: [shapedirs s-major: rec[s*3 + a], 3S floats][KW weights][KW/4 words of 4 joint ids].
// ---------------------------------------------------------------------------------------------
template <int KW>
struct Skin {
  int j[KW];
  float w[KW];
};

template <int S, int KW>
SF_HD Skin<KW> skin_from_rec(const float* rec) {
  Skin<KW> s;
#pragma unroll
  for (int k = 0; k < KW; ++k) s.w[k] = rec[3 * S + k];
#pragma unroll
  for (int q = 0; q < KW / 4; ++q) {
    uint32_t u;
    const float f = rec[3 * S + KW + q];
    __builtin_memcpy(&u, &f, 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) s.j[q * 4 + k] = (int)((u >> (8 * k)) & 0xffu);
  }
  return s;
}

// blended rotation and translation from 3 F4 per joint: R0..R8, T0
template <int S, int KW>
SF_HD void blend_rt(const float* jd, const Skin<KW>& sk, float* Rt, float* T0) {
  constexpr int STRIDE = jd_stride(S);
#pragma unroll
  for (int k = 0; k < 9; ++k) Rt[k] = 0.f;
  T0[0] = T0[1] = T0[2] = 0.f;
#pragma unroll
  for (int k = 0; k < KW; ++k) {
    const float* p = jd + sk.j[k] * STRIDE;
    const F4 a = ld4(p), b = ld4(p + 4), c = ld4(p + 8);
    const float w = sk.w[k];
    Rt[0] += w * a.x; Rt[1] += w * a.y; Rt[2] += w * a.z; Rt[3] += w * a.w;
    Rt[4] += w * b.x; Rt[5] += w * b.y; Rt[6] += w * b.z; Rt[7] += w * b.w;
    Rt[8] += w * c.x; T0[0] += w * c.y; T0[1] += w * c.z; T0[2] += w * c.w;
  }
}

// Normal-equation accumulation of one vertex (_fit_shape_gram, bodyfitter.py:999-1048):
//   Rt = sum_j w_j G_j;  pos = Rt v_posed + sum_j w_j T0_j;  b = t - pos
//   Jac[c][s] = Rt[c][:] . shapedirs[:, s] + sum_j w_j T'_j[c][s]
//   G += w Jac^T Jac, r += w Jac^T b, Sb[c] += w b[c]; SA[c] += w Jac[c] only when WEIGHTED — with
//   unit weights SA = sum_v Jac_v is target-independent and the joint stage evaluates it in closed
//   form from per-joint constants (JointTabs::cs_joint / cw_joint).
// acc: NE floats in the NE layout (W is handled by the caller).
template <int S, int KW, bool WEIGHTED>
SF_HD void shape_accum_vertex(const float* jd, const float* rec, const float* vp, const float* tv,
                              float wv, float* priv, float* acc) {
  constexpr int STRIDE = jd_stride(S), ROW = jd_row(S), NG = ne_ng(S), SD = 3 * S;
  const Skin<KW> sk = skin_from_rec<S, KW>(rec);
  {
    float Rt[9], T0[3];
    blend_rt<S, KW>(jd, sk, Rt, T0);
    // blended rotation row + residual of each coordinate go to a lane-private 12-float slot
    // (LDS on the GPU) so that the loop over the 3 coordinates below can stay ROLLED: fully unrolled
    // the compiler hoists every LDS load of all three rows (~190 live registers) and spills.
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float pos = (Rt[c * 3] * vp[0] + Rt[c * 3 + 1] * vp[1] + Rt[c * 3 + 2] * vp[2]) + T0[c];
      const float bc = tv[c] - pos;
      priv[c * 4] = Rt[c * 3];
      priv[c * 4 + 1] = Rt[c * 3 + 1];
      priv[c * 4 + 2] = Rt[c * 3 + 2];
      priv[c * 4 + 3] = bc;
      acc[NG + 4 * S + c] += WEIGHTED ? wv * bc : bc;
    }
  }
  float sd[SD + 3];  // shapedirs of the vertex, s-major: sd[s*3 + a]
#pragma unroll
  for (int q = 0; q < (SD + 3) / 4; ++q) {
    const F4 t = ld4(rec + 4 * q);
    if (4 * q < SD) sd[4 * q] = t.x;
    if (4 * q + 1 < SD) sd[4 * q + 1] = t.y;
    if (4 * q + 2 < SD) sd[4 * q + 2] = t.z;
    if (4 * q + 3 < SD) sd[4 * q + 3] = t.w;
  }
#pragma unroll 1
  for (int c = 0; c < 3; ++c) {
    const F4 rb = ld4(priv + c * 4);
    float a[ROW];
#pragma unroll
    for (int s = 0; s < S; ++s) a[s] = rb.x * sd[s * 3] + rb.y * sd[s * 3 + 1] + rb.z * sd[s * 3 + 2];
#pragma unroll
    for (int s = S; s < ROW; ++s) a[s] = 0.f;
#pragma unroll
    for (int k = 0; k < KW; ++k) {
      const float* p = jd + sk.j[k] * STRIDE + 12 + c * ROW;
      const float w = sk.w[k];
#pragma unroll
      for (int q = 0; q < ROW / 4; ++q) {
        const F4 t = ld4(p + 4 * q);
        a[4 * q] += w * t.x; a[4 * q + 1] += w * t.y; a[4 * q + 2] += w * t.z; a[4 * q + 3] += w * t.w;
      }
    }
    const float bc = rb.w;
#pragma unroll
    for (int i = 0; i < S; ++i) {
      const float wa = WEIGHTED ? wv * a[i] : a[i];
#pragma unroll
      for (int j = i; j < S; ++j) acc[ne_g(S, i, j)] += wa * a[j];
      acc[NG + i] += wa * bc;
      if (WEIGHTED) {  // SA[c][i]: the only c-indexed accumulators; branches keep the indices static
        if (c == 0) acc[NG + S + i] += wa;
        else if (c == 1) acc[NG + 2 * S + i] += wa;
        else acc[NG + 3 * S + i] += wa;
      }
    }
  }
}

// Extra sums of the scaled shape solve (scale_target / scale_fit, _fit_shape_general
// bodyfitter.py:1170-1175): with the scale column c = -t (scale_target) or c = pos = t - b (scale_fit)
// every new entry of the normal equations follows from
//   u = sum w Jac^T t,  tt = sum w |t|^2,  tb = sum w t.b,  bb = sum w |b|^2,  St = sum w t
// (see scaled_solve_stage).  acc: [u : S][tt][tb][bb][St : 3].
template <int S, int KW>
SF_HD void scale_extras_vertex(const float* jd, const float* rec, const float* vp, const float* tv,
                               float wv, float* acc) {
  constexpr int STRIDE = jd_stride(S), ROW = jd_row(S);
  const Skin<KW> sk = skin_from_rec<S, KW>(rec);
  float Rt[9], T0[3];
  blend_rt<S, KW>(jd, sk, Rt, T0);
  float bq = 0.f, tq = 0.f, tbq = 0.f;
  for (int c = 0; c < 3; ++c) {
    const float pos = (Rt[c * 3] * vp[0] + Rt[c * 3 + 1] * vp[1] + Rt[c * 3 + 2] * vp[2]) + T0[c];
    const float b = tv[c] - pos;
    tq += tv[c] * tv[c];
    tbq += tv[c] * b;
    bq += b * b;
    acc[S + 3 + c] += wv * tv[c];
    const float wt = wv * tv[c];
    for (int s = 0; s < S; ++s) {
      float a = Rt[c * 3] * rec[s * 3] + Rt[c * 3 + 1] * rec[s * 3 + 1] + Rt[c * 3 + 2] * rec[s * 3 + 2];
      for (int k = 0; k < KW; ++k) a += sk.w[k] * jd[sk.j[k] * STRIDE + 12 + c * ROW + s];
      acc[s] += wt * a;
    }
  }
  acc[S] += wv * tq;
  acc[S + 1] += wv * tbq;
  acc[S + 2] += wv * bq;
}

// Vertex at the solved shape (bodyfitter.py:1099-1101; LBS of bodymodel.py:288-306):
//   v = Rt (v_posed + shapedirs beta) + sum_j w_j jb_j + trans
// jb: (J,4) per-joint skinning translation at the solution; beta: S values (zero beyond nb).
template <int S, int KW>
SF_HD void lbs_vertex(const float* jd, const float* jb, const float* rec, const float* vp,
                      const float* beta, const float* trans, float* out) {
  constexpr int STRIDE = jd_stride(S), SD = 3 * S;
  const Skin<KW> sk = skin_from_rec<S, KW>(rec);
  float Rt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, Tb[3] = {0, 0, 0};
#pragma unroll
  for (int k = 0; k < KW; ++k) {
    const float* p = jd + sk.j[k] * STRIDE;
    const F4 a = ld4(p), b = ld4(p + 4);
    const float r8 = p[8];
    const F4 t = ld4(jb + sk.j[k] * 4);
    const float w = sk.w[k];
    Rt[0] += w * a.x; Rt[1] += w * a.y; Rt[2] += w * a.z; Rt[3] += w * a.w;
    Rt[4] += w * b.x; Rt[5] += w * b.y; Rt[6] += w * b.z; Rt[7] += w * b.w;
    Rt[8] += w * r8;
    Tb[0] += w * t.x; Tb[1] += w * t.y; Tb[2] += w * t.z;
  }
  float vs[3] = {vp[0], vp[1], vp[2]};
#pragma unroll
  for (int q = 0; q < (SD + 3) / 4; ++q) {
    const F4 t = ld4(rec + 4 * q);
    const float f[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = 4 * q + e;
      if (k < SD) vs[k % 3] += f[e] * beta[k / 3];
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c)
    out[c] = (Rt[c * 3] * vs[0] + Rt[c * 3 + 1] * vs[1] + Rt[c * 3 + 2] * vs[2]) + Tb[c] +
             trans[c];
}

// Residual pass of the pair-Gram form (unit weights).  With G and SA evaluated in closed form from
// the rotations, the vertex pass only needs Jac^T b, split as
//   r1 = sum_v S_v^T (Rt_v^T b_v)          (accumulated here, S sums per lane)
//   r2 = sum_j T'_j^T (sum_v w_vj b_v)     (the per-joint moments are scattered by the caller)
// acc: [r1 : S][Sb : 3]; b_out: the residual (for the scatter).
template <int S, int KW>
SF_HD void residual_vertex(const float* jd, const float* rec, const float* vp, const float* tv,
                           float* acc, float* b_out) {
  constexpr int SD = 3 * S;
  const Skin<KW> sk = skin_from_rec<S, KW>(rec);
  float Rt[9], T0[3];
  blend_rt<S, KW>(jd, sk, Rt, T0);
  float b[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float pos = (Rt[c * 3] * vp[0] + Rt[c * 3 + 1] * vp[1] + Rt[c * 3 + 2] * vp[2]) + T0[c];
    b[c] = tv[c] - pos;
    b_out[c] = b[c];
    acc[S + c] += b[c];
  }
  const float u[3] = {Rt[0] * b[0] + Rt[3] * b[1] + Rt[6] * b[2], Rt[1] * b[0] + Rt[4] * b[1] + Rt[7] * b[2],
                      Rt[2] * b[0] + Rt[5] * b[1] + Rt[8] * b[2]};
#pragma unroll
  for (int q = 0; q < (SD + 3) / 4; ++q) {
    const F4 t = ld4(rec + 4 * q);
    const float f[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = 4 * q + e;
      if (k < SD) acc[k / 3] += f[e] * u[k % 3];
    }
  }
}

// Pair-Gram stage: G = sum_v Jac_v^T Jac_v for unit weights, from the rotations alone.
//   Jac_v = sum_j w_vj R_j (S_v + D_j),  D_j = R_j^T T'_j   =>
//   G = G0 + sum_j [C2_jj^T D_j + D_j^T C2_jj + C3_jj D_j^T D_j] + sum_{j<j'} (f_jj' + f_jj'^T),
//   f_jj'[i][i'] = sum_a ( sum_a' Q[a][a'] C1[a][a'][i][i'] + C2[a][i] U[a][i'] + D_j[a][i] V[a][i'] ),
//   Q = R_j^T R_j',  U = Q D_j',  V = Q C2 + C3 U.
// scratch: Qs (np*9) then Ds (J*3*S) floats.  Output: NG doubles (upper triangle, NE order).
template <class Ctx>
SF_HD void pair_gram_stage(Ctx& cx, const JointTabs& tb, float* scratch, const float* jd,
                           double* G_out) {
#define SF_FOR2(i, count) for (int i = cx.lane; i < (count); i += cx.n)
  const int J = tb.J, S = tb.S, np = tb.np, stride = jd_stride(S), row = jd_row(S);
  float* Qs = scratch;
  float* Ds = scratch + np * 9;
  SF_FOR2(idx, J * 3 * S) {
    const int j = idx / (3 * S), a = (idx / S) % 3, i = idx % S;
    const float* R = jd + j * stride;
    const float* T = jd + j * stride + 12;
    Ds[idx] = R[a] * T[i] + R[3 + a] * T[row + i] + R[6 + a] * T[2 * row + i];
  }
  SF_FOR2(idx, np * 9) {
    const int p = idx / 9, a = (idx / 3) % 3, a2 = idx % 3;
    const float* R1 = jd + tb.pair_j[2 * p] * stride;
    const float* R2 = jd + tb.pair_j[2 * p + 1] * stride;
    Qs[idx] = R1[a] * R2[a2] + R1[3 + a] * R2[3 + a2] + R1[6 + a] * R2[6 + a2];
  }
  cx.sync();
  const int NG = ne_ng(S);
  SF_FOR2(e, NG) {
    int i = 0, r = e;
    while (r >= S - i) {
      r -= S - i;
      ++i;
    }
    const int i2 = i + r;
    float acc = tb.diag_g0[i * S + i2];
    for (int j = 0; j < J; ++j) {
      const float* D = Ds + j * 3 * S;
      const float* c2 = tb.diag_c2 + j * 3 * S;
      const float c3 = tb.diag_c3[j];
      for (int a = 0; a < 3; ++a)
        acc += c2[a * S + i] * D[a * S + i2] + D[a * S + i] * (c2[a * S + i2] + c3 * D[a * S + i2]);
    }
    for (int p = 0; p < np; ++p) {
      const float* Q = Qs + p * 9;
      const float* D1 = Ds + tb.pair_j[2 * p] * 3 * S;
      const float* D2 = Ds + tb.pair_j[2 * p + 1] * 3 * S;
      const float* c1 = tb.pair_c1 + (size_t)p * 9 * S * S;
      const float* c2 = tb.pair_c2 + p * 3 * S;
      const float c3 = tb.pair_c3[p];
      // f[i][i2] + f[i2][i]
      for (int pass = 0; pass < 2; ++pass) {
        const int x = pass == 0 ? i : i2, y = pass == 0 ? i2 : i;
        float f = 0.f;
        for (int a = 0; a < 3; ++a) {
          const float U = Q[a * 3] * D2[y] + Q[a * 3 + 1] * D2[S + y] + Q[a * 3 + 2] * D2[2 * S + y];
          const float Vv = (Q[a * 3] * c2[y] + Q[a * 3 + 1] * c2[S + y] + Q[a * 3 + 2] * c2[2 * S + y]) + c3 * U;
          f += (Q[a * 3] * c1[((a * 3) * S + x) * S + y] + Q[a * 3 + 1] * c1[((a * 3 + 1) * S + x) * S + y] +
                Q[a * 3 + 2] * c1[((a * 3 + 2) * S + x) * S + y]) +
               c2[a * S + x] * U + D1[a * S + x] * Vv;
        }
        acc += f;
      }
    }
    G_out[e] = (double)acc;
  }
#undef SF_FOR2
}

// Part-sum accumulation of one vertex (_part_sums, bodyfitter.py:257-280): acc is a kPsum record.
SF_HD void partsum_vertex(const float* t, const float* a_in, float w, bool weighted, float* acc) {
  float a[3] = {a_in[0], a_in[1], a_in[2]};
  float ts[3] = {t[0], t[1], t[2]};
  if (weighted) {
    for (int c = 0; c < 3; ++c) {
      a[c] *= w;
      ts[c] *= w;
    }
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int c = 0; c < 3; ++c) acc[r * 3 + c] += t[r] * a[c];
    acc[9 + r] += ts[r];
    acc[12 + r] += a[r];
  }
  acc[15] += weighted ? w : 1.0f;
}

#undef SF_FOR

}  // namespace sf
