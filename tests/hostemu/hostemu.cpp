// TEST-ONLY host emulation of the HIP pipeline (never shipped, never loaded by the package).
//
// The device code keeps its arithmetic in headers shared with this file (csrc/sf_math.h,
// csrc/sf_stages.h, csrc/sf_tables.cpp): the joint-level stages run here with a one-lane context
// (lane = 0, n = 1, sync = no-op) and the per-vertex bodies run in plain loops in the same order of
// kernels as run_fit() in csrc/smplfit_hip.hip.  This lets the build container (no GPU) check the
// table builder, the stage logic and the orchestration against the oracle and the golden vectors
// before any GPU minute is spent; wave reductions / LDS staging / MFMA remain GPU-only code that the
// -m gpu parity tests cover.  Built by tests/test_hostemu.py with g++.
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../smplfitter_amd/csrc/sf_stages.h"
#include "../../smplfitter_amd/csrc/sf_tables.h"

namespace {

struct HostCtx {
  int lane = 0, n = 1;
  void sync() const {}
  float sum_to_last(float v) const { return v; }  // (one lane: the partial sum is the sum)
};

sf::JointTabs make_tabs(const sf::HostTables& t) {
  sf::JointTabs jt;
  jt.J = t.J; jt.S = t.S; jt.num_levels = t.num_levels(); jt.adj_last_level = t.adj_last_level;
  jt.P = t.P; jt.Kp = t.Kp;
  jt.n_kid = t.n_kid;
  jt.n_pad = t.n_pad;
  jt.parents = t.parents.data(); jt.fk_js = t.fk_js.data();
  jt.fk_level_start = t.fk_level_start.data(); jt.cas_start = t.cas_start.data();
  jt.cas_flat = t.cas_flat.data(); jt.part_type = t.part_type.data(); jt.toe_src = t.toe_src.data();
  jt.adj_level_start = t.adj_level_start.data(); jt.adj_parts = t.adj_parts.data();
  jt.j_ext = t.j_ext.data(); jt.bone_ext = t.bone_ext.data();
  jt.fk_jp = t.fk_jp.data(); jt.bone_lv = t.bone_lv.data();
  jt.cs_joint = t.cs_joint.data(); jt.cw_joint = t.cw_joint.data();
  jt.np = (int)t.pair_c3.size(); jt.pair_j = t.pair_j.data(); jt.pair_c1 = t.pair_c1.data();
  jt.pair_c2 = t.pair_c2.data(); jt.pair_c3 = t.pair_c3.data(); jt.diag_g0 = t.diag_g0.data();
  jt.diag_c2 = t.diag_c2.data(); jt.diag_c3 = t.diag_c3.data();
  return jt;
}

template <int S, int KW>
struct Emu {
  const sf::HostTables& t;
  sf::JointTabs jt;
  int B;
  std::vector<float> tvs, vws, vposed, rp, mean, tjc, psum, G, jd, pext, gramj, beta, trans, jb,
      rjoints, rverts, tjreg, rjreg, scratch, regref;  // regref: (B,S) ridge reference, empty = none
  std::vector<double> gramv;
  sf::JointScratch sh;
  std::vector<float> solve_scratch;

  Emu(const sf::HostTables& tt, int b) : t(tt), jt(make_tabs(tt)), B(b) {
    const int J = t.J, Vp = t.Vp, NE1 = sf::ne_size(S) + 1;
    tvs.assign((size_t)B * 3 * Vp, 0.f); vws.assign((size_t)B * Vp, 0.f);
    vposed.assign((size_t)B * 3 * Vp, 0.f); rp.assign((size_t)B * t.Kp, 0.f);
    mean.assign(B * 3, 0.f); tjc.assign((size_t)B * J * 3, 0.f);
    psum.assign((size_t)B * J * sf::kPsum, 0.f); G.assign((size_t)B * J * 9, 0.f);
    jd.assign((size_t)B * J * sf::jd_stride(S) + 4, 0.f); pext.assign((size_t)B * J * 3 * (S + 1), 0.f);
    gramj.assign((size_t)B * NE1, 0.f); gramv.assign((size_t)B * NE1, 0.0);
    beta.assign(B * S, 0.f); trans.assign(B * 3, 0.f); jb.assign((size_t)B * J * 4, 0.f);
    rjoints.assign((size_t)B * J * 3, 0.f); rverts.assign((size_t)B * 3 * Vp, 0.f);
    tjreg.assign((size_t)B * J * 3, 0.f); rjreg.assign((size_t)B * J * 3, 0.f);
    scratch.assign(sf::joint_scratch_floats(J, S, 0) + 8, 0.f);
    solve_scratch.assign(sf::solve_scratch_floats(S) + 8, 0.f);
    // 16-byte align the scratch base
    float* base = scratch.data();
    while ((uintptr_t)base & 15) ++base;
    sh = sf::carve_joint_scratch(base, J, S, 0);
    cpack_store.assign(t.cpackA.size() + 4, 0.f);
    float* cp = cpack_store.data();
    while ((uintptr_t)cp & 15) ++cp;
    std::memcpy(cp, t.cpackA.data(), t.cpackA.size() * sizeof(float));
    cpack_aligned = cp;
  }

  const float* rec(int slot) const {  // 16-byte aligned copy of the packed constants
    return cpack_aligned + (size_t)slot * sf::cpack_stride(S, KW);
  }
  std::vector<float> cpack_store;
  const float* cpack_aligned = nullptr;

  float* solve_base() {
    float* p = solve_scratch.data();
    while ((uintptr_t)p & 15) ++p;
    return p;
  }

  float* jd_b(int b) {  // 16-byte aligned per-instance joint block (stride is a multiple of 4 floats)
    float* p = jd.data();
    while ((uintptr_t)p & 15) ++p;
    return p + (size_t)b * t.J * sf::jd_stride(S);
  }

  void k0(const float* tv, const float* tj, const float* vw) {
    const int V = t.V, J = t.J, Vp = t.Vp;
    for (int b = 0; b < B; ++b) {
      const float* tvb = tv + (size_t)b * V * 3;
      float s[3] = {0, 0, 0};
      for (int v = 0; v < V; ++v) for (int c = 0; c < 3; ++c) s[c] += tvb[v * 3 + c];
      if (tj) for (int j = 0; j < J; ++j) for (int c = 0; c < 3; ++c) s[c] += tj[((size_t)b * J + j) * 3 + c];
      const float n = (float)(V + (tj ? J : 0));
      float mu[3];
      for (int c = 0; c < 3; ++c) { mu[c] = s[c] / n; mean[b * 3 + c] = mu[c]; }
      if (tj) for (int k = 0; k < J * 3; ++k) tjc[(size_t)b * J * 3 + k] = tj[(size_t)b * J * 3 + k] - mu[k % 3];
      float* ps = psum.data() + (size_t)b * J * sf::kPsum;
      std::fill(ps, ps + J * sf::kPsum, 0.f);
      for (int i = 0; i < Vp; ++i) {
        const int o = t.perm[i];
        float tt[3] = {0, 0, 0}, w = 0.f;
        if (o >= 0) {
          for (int c = 0; c < 3; ++c) tt[c] = tvb[o * 3 + c] - mu[c];
          if (vw) w = vw[(size_t)b * V + o];
        }
        for (int c = 0; c < 3; ++c) tvs[((size_t)b * 3 + c) * Vp + i] = tt[c];
        if (vw) vws[(size_t)b * Vp + i] = w;
        if (i < t.n_used) {
          const float a[3] = {t.dm[i], t.dm[Vp + i], t.dm[2 * Vp + i]};
          sf::partsum_vertex(tt, a, vw ? w : 1.f, vw != nullptr, ps + t.slot_part[i] * sf::kPsum);
        }
      }
    }
  }

  void regress(const float* src, bool shared_src, float* out, int nb) {
    const int J = t.J, Vp = t.Vp;
    for (int b = 0; b < nb; ++b) {
      const float* s = shared_src ? src : src + (size_t)b * 3 * Vp;
      for (int j = 0; j < J; ++j) {
        float a[3] = {0, 0, 0};
        for (int k = t.reg_start[j]; k < t.reg_start[j + 1]; ++k)
          for (int c = 0; c < 3; ++c) a[c] += t.reg_val[k] * s[c * Vp + t.reg_slot[k]];
        for (int c = 0; c < 3; ++c) out[((size_t)b * J + j) * 3 + c] = a[c];
      }
    }
  }

  void k1(const float* tj, const float* rj, bool rj_shared, const float* Gprev, const float* jw,
          bool fit_rot, bool prologue, bool jblock, bool jblock_w, bool sa_closed) {
    const int J = t.J, NE1 = sf::ne_size(S) + 1;
    HostCtx cx;
    for (int b = 0; b < B; ++b)
      sf::joint_stage(cx, jt, sh, psum.data() + (size_t)b * J * sf::kPsum, tj + (size_t)b * J * 3,
                      rj ? (rj_shared ? rj : rj + (size_t)b * J * 3) : nullptr,
                      Gprev ? Gprev + (size_t)b * J * 9 : nullptr, jw ? jw + (size_t)b * J : nullptr,
                      fit_rot, prologue, jblock, jblock_w, sa_closed, G.data() + (size_t)b * J * 9,
                      rp.data() + (size_t)b * t.Kp, jd_b(b), pext.data() + (size_t)b * J * 3 * (S + 1),
                      gramj.data() + (size_t)b * NE1);
  }

  void gemm() {  // k-ordered fp32 fma chain per output, like the MFMA path
    const int N = 3 * t.Vp;
    for (int b = 0; b < B; ++b)
      for (int n = 0; n < N; ++n) {
        float acc = 0.f;  // the template comes in through the padding row of posedirs
        for (int k = 0; k < t.Kp; ++k) acc = fmaf(rp[(size_t)b * t.Kp + k], t.pdT[(size_t)k * N + n], acc);
        vposed[(size_t)b * N + n] = acc;
      }
  }

  std::vector<float> mbj;  // (B,J,3) per-joint residual moments of the pair-Gram form
  bool use_pair_gram = false;

  void k3_pair(void) {  // unit weights: residual pass + pair-Gram (mirrors k_residual / k_pair_gram)
    const int Vp = t.Vp, NE = sf::ne_size(S), NG = sf::ne_ng(S), J = t.J;
    mbj.assign((size_t)B * J * 3, 0.f);
    std::vector<float> pscratch((size_t)jt.np * 9 + (size_t)J * 3 * S + 8);
    HostCtx cx;
    for (int b = 0; b < B; ++b) {
      std::vector<double> dacc(NE + 1, 0.0);
      for (const auto& g : t.gtiles) {  // per-tile fp32 partials, like the per-lane partials on the GPU
        float acc[S + 3];
        for (int k = 0; k < S + 3; ++k) acc[k] = 0.f;
        for (int l = 0; l < g.count; ++l) {
          const int i = g.start + l;
          const float vp[3] = {vposed[((size_t)b * 3) * Vp + i], vposed[((size_t)b * 3 + 1) * Vp + i],
                               vposed[((size_t)b * 3 + 2) * Vp + i]};
          const float tv[3] = {tvs[((size_t)b * 3) * Vp + i], tvs[((size_t)b * 3 + 1) * Vp + i],
                               tvs[((size_t)b * 3 + 2) * Vp + i]};
          float bo[3];
          sf::residual_vertex<S, KW>(jd_b(b), rec(i), vp, tv, acc, bo);
          const sf::Skin<KW> sk = sf::skin_from_rec<S, KW>(rec(i));
          for (int k = 0; k < KW; ++k)
            for (int c = 0; c < 3; ++c) mbj[((size_t)b * J + sk.j[k]) * 3 + c] += sk.w[k] * bo[c];
        }
        for (int k = 0; k < S; ++k) dacc[NG + k] += (double)acc[k];
        for (int c = 0; c < 3; ++c) dacc[NG + 4 * S + c] += (double)acc[S + c];
      }
      dacc[NE] = (double)t.V;
      sf::pair_gram_stage(cx, jt, pscratch.data(), jd_b(b), dacc.data());
      for (int k = 0; k <= NE; ++k) gramv[(size_t)b * (NE + 1) + k] = dacc[k];
    }
    use_pair_gram = true;
  }

  void k3(bool weighted) {
    if (!weighted) { k3_pair(); return; }
    use_pair_gram = false;
    const int Vp = t.Vp, NE = sf::ne_size(S);
    for (int b = 0; b < B; ++b) {
      float acc[sf::ne_size(S) + 1];
      alignas(16) float priv[12];
      for (int k = 0; k <= NE; ++k) acc[k] = 0.f;
      std::vector<double> dacc(NE + 1, 0.0);
      // accumulate in chunks so fp32 partial sums stay short, like the per-lane partials on the GPU
      for (int i0 = 0; i0 < Vp; i0 += 32) {
        for (int k = 0; k <= NE; ++k) acc[k] = 0.f;
        for (int i = i0; i < i0 + 32 && i < Vp; ++i) {
          const float vp[3] = {vposed[((size_t)b * 3) * Vp + i], vposed[((size_t)b * 3 + 1) * Vp + i],
                               vposed[((size_t)b * 3 + 2) * Vp + i]};
          const float tv[3] = {tvs[((size_t)b * 3) * Vp + i], tvs[((size_t)b * 3 + 1) * Vp + i],
                               tvs[((size_t)b * 3 + 2) * Vp + i]};
          float wv = 1.f;
          if (weighted) { wv = vws[(size_t)b * Vp + i]; acc[NE] += wv; }
          if (weighted)
            sf::shape_accum_vertex<S, KW, true>(jd_b(b), rec(i), vp, tv, wv, priv, acc);
          else
            sf::shape_accum_vertex<S, KW, false>(jd_b(b), rec(i), vp, tv, wv, priv, acc);
        }
        for (int k = 0; k <= NE; ++k) dacc[k] += (double)acc[k];
      }
      if (!weighted) dacc[NE] = (double)t.V;
      for (int k = 0; k <= NE; ++k) gramv[(size_t)b * (NE + 1) + k] = dacc[k];
    }
  }

  bool share_beta = false;  // one shape for the batch: assemble, sum in instance order, solve the sum
  smplfit_share_allreduce_fn share_allreduce = nullptr;  // as smplfit_fit_args.share_allreduce, host memory
  void* share_user = nullptr;
  // row B of cen = sum of rows 0..B-1 in the order of k_share_reduce, then the sum over the ranks
  void share_sum(double* cen, int NC) {
    for (int e = 0; e < NC; ++e) {
      double a[4] = {0, 0, 0, 0};
      int b = 0;
      for (; b + 3 < B; b += 4)
        for (int q = 0; q < 4; ++q) a[q] += cen[(size_t)(b + q) * NC + e];
      for (; b < B; ++b) a[0] += cen[(size_t)b * NC + e];
      cen[(size_t)B * NC + e] = (a[0] + a[1]) + (a[2] + a[3]);
    }
    if (share_allreduce) share_allreduce(share_user, cen + (size_t)B * NC, NC, nullptr);
  }

  void k4(float reg, float reg2, float kid_reg) {
    const int J = t.J, NE1 = sf::ne_size(S) + 1, NC = S * S + S;
    HostCtx cx;
    std::vector<double> cen((size_t)(B + 1) * NC, 0.0);
    auto stage = [&](int b, int mode, double* c) {
      sf::solve_stage(cx, jt, solve_base(), gramv.data() + (size_t)b * NE1, gramj.data() + (size_t)b * NE1,
                      pext.data() + (size_t)b * J * 3 * (S + 1), jd_b(b),
                      use_pair_gram ? mbj.data() + (size_t)b * J * 3 : nullptr, reg, reg2, kid_reg,
                      beta.data() + (size_t)b * S, trans.data() + (size_t)b * 3,
                      rjoints.data() + (size_t)b * J * 3, jb.data() + (size_t)b * J * 4,
                      // the all-shared solve ignores the ridge reference (pt/lstsq.py:45-47)
                      (regref.empty() || mode != 0) ? nullptr : regref.data() + (size_t)b * S, mode, c);
    };
    if (!share_beta) {
      for (int b = 0; b < B; ++b) stage(b, 0, nullptr);
      return;
    }
    for (int b = 0; b < B; ++b) stage(b, 1, cen.data() + (size_t)b * NC);
    share_sum(cen.data(), NC);
    for (int b = 0; b < B; ++b) stage(b, 2, cen.data() + (size_t)B * NC);
  }

  void vertex(int b, int i, const float* be, int nb, const float* tr, float* v) {
    const int Vp = t.Vp;
    const float vp[3] = {vposed[((size_t)b * 3) * Vp + i], vposed[((size_t)b * 3 + 1) * Vp + i],
                         vposed[((size_t)b * 3 + 2) * Vp + i]};
    float bb[S];
    for (int s = 0; s < S; ++s) bb[s] = (be && s < nb) ? be[s] : 0.f;
    alignas(16) float jbl[sf::kMaxJoints * 4];
    std::memcpy(jbl, jb.data() + (size_t)b * t.J * 4, sizeof(float) * t.J * 4);
    sf::lbs_vertex<S, KW>(jd_b(b), jbl, rec(i), vp, bb, tr, v);
  }

  void k5(bool weighted, bool store) {
    const int J = t.J, Vp = t.Vp;
    for (int b = 0; b < B; ++b) {
      float* ps = psum.data() + (size_t)b * J * sf::kPsum;
      std::fill(ps, ps + J * sf::kPsum, 0.f);
      for (int i = 0; i < (store ? t.V : t.n_used); ++i) {
        float v[3];
        vertex(b, i, beta.data() + (size_t)b * S, S, trans.data() + (size_t)b * 3, v);
        if (store) for (int c = 0; c < 3; ++c) rverts[((size_t)b * 3 + c) * Vp + i] = v[c];
        if (i < t.n_used) {
          const float tt[3] = {tvs[((size_t)b * 3) * Vp + i], tvs[((size_t)b * 3 + 1) * Vp + i],
                               tvs[((size_t)b * 3 + 2) * Vp + i]};
          sf::partsum_vertex(tt, v, weighted ? vws[(size_t)b * Vp + i] : 1.f, weighted,
                             ps + t.slot_part[i] * sf::kPsum);
        }
      }
    }
  }
};

thread_local std::string g_err;

struct Warm {  // warm start of fit (mirrors FitOptions::init_* in smplfit_hip.hip)
  const float* pose = nullptr;
  const float* betas = nullptr;
  int nb = 0;
  const float* kid = nullptr;
  int share_beta = 0;
  smplfit_share_allreduce_fn share_allreduce = nullptr;
  void* share_user = nullptr;
  int scale_mode = 0;  // 1 scale_target, 2 scale_fit
  float scale_reg = 0.f;
  float* scale_out = nullptr;
};
thread_local Warm g_warm;  // set by hostemu_fit_warm around hostemu_fit

template <int S, int KW>
int fit_impl(const sf::HostTables& t, const float* tv, const float* tj, const float* vw,
             const float* jw, int B, int num_iter, float reg, float reg2, float kid_reg, int final_adjust,
             float* pose, float* betas, float* trans, float* kid, float* orient, float* G0_out) {
  const Warm w = g_warm;
  Emu<S, KW> e(t, B);
  e.share_beta = g_warm.share_beta != 0;
  e.share_allreduce = g_warm.share_allreduce;
  e.share_user = g_warm.share_user;
  const bool joints = tj != nullptr;
  const bool eff_v = joints ? (vw && jw) : (vw != nullptr);
  const bool eff_j = joints && vw && jw;
  e.k0(tv, tj, vw);
  const float* tj_rot = e.tjc.data();
  std::vector<float> jtemplate((size_t)t.J * 3);
  for (int k = 0; k < t.J * 3; ++k) jtemplate[k] = t.j_ext[(size_t)k * (S + 1)];
  std::vector<float> rj0((size_t)t.J * 3);
  if (!joints) {
    e.regress(e.tvs.data(), false, e.tjreg.data(), B);
    tj_rot = e.tjreg.data();
    e.regress(t.dm.data(), true, rj0.data(), 1);
  } else {
    rj0 = jtemplate;
  }
  const bool warm = w.pose || w.betas;
  if (warm || w.kid) {  // k_fill_shape: the ridge reference also for an initial_kid_factor on its own
    const int nbe = std::min(w.nb, t.num_betas());
    for (int b = 0; b < B; ++b) {
      for (int s = 0; s < S - t.n_kid; ++s)
        e.beta[(size_t)b * S + s] = (w.betas && s < nbe) ? w.betas[(size_t)b * w.nb + s] : 0.f;
      if (t.n_kid) e.beta[(size_t)b * S + S - 1] = w.kid ? w.kid[b] : 0.f;
      for (int c = 0; c < 3; ++c) e.trans[(size_t)b * 3 + c] = 0.f;
    }
    if (w.betas || w.kid) e.regref = e.beta;
  }
  if (warm) {
    HostCtx cx;
    for (int b = 0; b < B; ++b) {
      sf::forward_joint_stage(cx, e.jt, e.sh, w.pose ? w.pose + (size_t)b * t.J * 3 : nullptr, nullptr,
                              e.beta.data() + (size_t)b * S, S, nullptr, nullptr,
                              e.rp.data() + (size_t)b * t.Kp, e.jd_b(b),
                              e.rjoints.data() + (size_t)b * t.J * 3, e.G.data() + (size_t)b * t.J * 9);
      for (int k = 0; k < t.J * 3; ++k)
        e.jb[(size_t)b * t.J * 4 + (k / 3) * 4 + k % 3] = e.jd_b(b)[(k / 3) * sf::jd_stride(S) + 9 + k % 3];
    }
    e.gemm();
    e.k5(vw != nullptr, true);
    if (!joints) e.regress(e.rverts.data(), false, e.rjreg.data(), B);
    std::vector<float> Gprev = e.G;
    e.k1(tj_rot, joints ? e.rjoints.data() : e.rjreg.data(), false, Gprev.data(), jw, true, true, joints,
         eff_j, !eff_v);
  } else {
    e.k1(tj_rot, rj0.data(), true, nullptr, jw, true, true, joints, eff_j, !eff_v);
  }
  if (G0_out) std::memcpy(G0_out, e.G.data(), sizeof(float) * (size_t)B * t.J * 9);
  std::vector<float> scale, beta_und, tjs;
  for (int it = 0; it < num_iter; ++it) {
    e.gemm();
    e.k3(eff_v);
    const bool last = it + 1 == num_iter;
    if (w.scale_mode && last) {  // mirrors k_scale_extras + k_shape_solve_scaled
      HostCtx cx0;
      const int NE1 = sf::ne_size(S) + 1, NX = S + sf::kScaleExtras, Vp = t.Vp;
      scale.assign(B, 1.f);
      beta_und.assign((size_t)B * S, 0.f);
      std::vector<float> scratch(sf::scaled_solve_scratch_floats(S) + 8);
      float* sb = scratch.data();
      while ((uintptr_t)sb & 15) ++sb;
      std::vector<float> vextra((size_t)B * 32, 0.f);
      for (int b = 0; b < B; ++b) {
        std::vector<double> dacc(NX, 0.0);
        for (int i0 = 0; i0 < t.V; i0 += 64) {  // per-lane fp32 partials as on the GPU, then summed
          for (int i = i0; i < i0 + 64 && i < t.V; ++i) {
            float acc[NX];
            for (int k = 0; k < NX; ++k) acc[k] = 0.f;
            const float vp[3] = {e.vposed[((size_t)b * 3) * Vp + i], e.vposed[((size_t)b * 3 + 1) * Vp + i],
                                 e.vposed[((size_t)b * 3 + 2) * Vp + i]};
            const float tvv[3] = {e.tvs[((size_t)b * 3) * Vp + i], e.tvs[((size_t)b * 3 + 1) * Vp + i],
                                  e.tvs[((size_t)b * 3 + 2) * Vp + i]};
            sf::scale_extras_vertex<S, KW>(e.jd_b(b), e.rec(i), vp, tvv, eff_v ? e.vws[(size_t)b * Vp + i] : 1.f, acc);
            for (int k = 0; k < NX; ++k) dacc[k] += (double)acc[k];
          }
        }
        for (int k = 0; k < NX; ++k) vextra[(size_t)b * 32 + k] = (float)dacc[k];
      }
      auto stage = [&](int b, int share, double* cen) {
        sf::scaled_solve_stage(cx0, e.jt, sb, e.gramv.data() + (size_t)b * NE1, e.gramj.data() + (size_t)b * NE1,
                               vextra.data() + (size_t)b * 32, e.pext.data() + (size_t)b * t.J * 3 * (S + 1),
                               e.jd_b(b), e.use_pair_gram ? e.mbj.data() + (size_t)b * t.J * 3 : nullptr,
                               joints ? e.tjc.data() + (size_t)b * t.J * 3 : nullptr,
                               eff_j ? jw + (size_t)b * t.J : nullptr, joints, w.scale_mode, reg, reg2, kid_reg,
                               w.scale_reg, e.regref.empty() ? nullptr : e.regref.data() + (size_t)b * S,
                               beta_und.data() + (size_t)b * S, e.beta.data() + (size_t)b * S,
                               e.trans.data() + (size_t)b * 3, scale.data() + b,
                               e.rjoints.data() + (size_t)b * t.J * 3, e.jb.data() + (size_t)b * t.J * 4, share, cen);
      };
      if (!e.share_beta) {
        for (int b = 0; b < B; ++b) stage(b, 0, nullptr);
      } else {  // shared shape, own scale: mirrors the three launches of enqueue_solve
        const int NC = S * S + S;
        std::vector<double> cen((size_t)(B + 1) * NC, 0.0);
        for (int b = 0; b < B; ++b) stage(b, 1, cen.data() + (size_t)b * NC);
        e.share_sum(cen.data(), NC);
        for (int b = 0; b < B; ++b) stage(b, 2, cen.data() + (size_t)B * NC);
      }
    } else {
      e.k4(reg, reg2, kid_reg);
    }
    if (last && !final_adjust) break;
    e.k5(vw != nullptr, !joints);
    if (!joints) e.regress(e.rverts.data(), false, e.rjreg.data(), B);
    if (last) break;
    std::vector<float> Gprev = e.G;
    e.k1(tj_rot, joints ? e.rjoints.data() : e.rjreg.data(), false, Gprev.data(), jw, true, true,
         joints, eff_j, !eff_v);
  }
  HostCtx cx;
  const float* tj_ref = tj_rot;
  if (w.scale_mode) {  // mirrors k_scale_refs
    const int J = t.J;
    tjs.assign(tj_rot, tj_rot + (size_t)B * J * 3);
    for (int b = 0; b < B; ++b) {
      const float sc = scale[b];
      const float tr[3] = {(1.f - sc) * e.trans[b * 3], (1.f - sc) * e.trans[b * 3 + 1], (1.f - sc) * e.trans[b * 3 + 2]};
      if (final_adjust)
        for (int j = 0; j < J; ++j) {
          float* ps = e.psum.data() + ((size_t)b * J + j) * sf::kPsum;
          if (w.scale_mode == 1) {
            for (int k = 0; k < 12; ++k) ps[k] *= sc;
            for (int c = 0; c < 3; ++c) tjs[((size_t)b * J + j) * 3 + c] = sc * tj_rot[((size_t)b * J + j) * 3 + c];
          } else {
            for (int r = 0; r < 3; ++r)
              for (int c = 0; c < 3; ++c) ps[r * 3 + c] = sc * ps[r * 3 + c] + ps[9 + r] * tr[c];
            for (int c = 0; c < 3; ++c) {
              ps[12 + c] = sc * ps[12 + c] + ps[15] * tr[c];
              float& rj = e.rjoints[((size_t)b * J + j) * 3 + c];
              rj = sc * rj + tr[c];
              if (!joints) {
                float& rr = e.rjreg[((size_t)b * J + j) * 3 + c];
                rr = sc * rr + t.reg_rowsum[j] * tr[c];
              }
            }
          }
        }
      for (int s2 = 0; s2 < S; ++s2) e.beta[(size_t)b * S + s2] = beta_und[(size_t)b * S + s2];
      for (int c = 0; c < 3; ++c) e.mean[b * 3 + c] = w.scale_mode == 1 ? e.mean[b * 3 + c] * sc : e.mean[b * 3 + c] / sc;
      if (w.scale_out) w.scale_out[b] = sc;
    }
    if (w.scale_mode == 1 && final_adjust) tj_ref = tjs.data();
  }
  for (int b = 0; b < B; ++b)
    sf::refine_stage(cx, e.jt, e.sh, e.psum.data() + (size_t)b * t.J * sf::kPsum,
                     tj_ref + (size_t)b * t.J * 3,
                     (joints ? e.rjoints.data() : e.rjreg.data()) + (size_t)b * t.J * 3,
                     e.rjoints.data() + (size_t)b * t.J * 3, jw ? jw + (size_t)b * t.J : nullptr,
                     e.G.data() + (size_t)b * t.J * 9, e.beta.data() + (size_t)b * S,
                     e.trans.data() + (size_t)b * 3, e.mean.data() + (size_t)b * 3, final_adjust != 0,
                     pose + (size_t)b * t.J * 3, betas + (size_t)b * t.num_betas(), trans + (size_t)b * 3,
                     kid ? kid + b : nullptr, orient ? orient + (size_t)b * t.J * 9 : nullptr, nullptr,
                     w.scale_mode == 2 ? scale.data() + b : nullptr);
  return 0;
}

// mirrors run_fit_known_shape (smplfit_hip.hip)
template <int S, int KW>
int known_shape_impl(const sf::HostTables& t, const float* betas, int nb, const float* kid,
                     const float* init_pose, const float* tv, const float* tj, const float* vw,
                     const float* jw, int B, int num_iter, int final_adjust, int scale_fit, float* pose,
                     float* trans, float* scale_out, float* orient) {
  Emu<S, KW> e(t, B);
  const bool joints = tj != nullptr;
  if (!joints && !t.has_regressor) { g_err = "no regressor"; return -1; }
  e.k0(tv, tj, vw);
  const float* tj_rot = e.tjc.data();
  if (!joints) {
    e.regress(e.tvs.data(), false, e.tjreg.data(), B);
    tj_rot = e.tjreg.data();
  }
  const int nbe = std::min(nb, t.num_betas());
  for (int b = 0; b < B; ++b) {
    for (int s = 0; s < S - t.n_kid; ++s) e.beta[(size_t)b * S + s] = s < nbe ? betas[(size_t)b * nb + s] : 0.f;
    if (t.n_kid) e.beta[(size_t)b * S + S - 1] = kid ? kid[b] : 0.f;
    for (int c = 0; c < 3; ++c) e.trans[(size_t)b * 3 + c] = 0.f;
  }
  HostCtx cx;
  std::vector<float> scale((size_t)B, 1.f);
  for (int it = 0; it <= num_iter; ++it) {
    for (int b = 0; b < B; ++b) {
      std::vector<float> Gin(e.G.begin() + (size_t)b * t.J * 9, e.G.begin() + (size_t)(b + 1) * t.J * 9);
      sf::forward_joint_stage(cx, e.jt, e.sh, (it == 0 && init_pose) ? init_pose + (size_t)b * t.J * 3 : nullptr,
                              it == 0 ? nullptr : Gin.data(), e.beta.data() + (size_t)b * S, S, nullptr,
                              nullptr, e.rp.data() + (size_t)b * t.Kp, e.jd_b(b),
                              e.rjoints.data() + (size_t)b * t.J * 3, e.G.data() + (size_t)b * t.J * 9);
      for (int k = 0; k < t.J * 3; ++k)
        e.jb[(size_t)b * t.J * 4 + (k / 3) * 4 + k % 3] = e.jd_b(b)[(k / 3) * sf::jd_stride(S) + 9 + k % 3];
    }
    e.gemm();
    e.k5(vw != nullptr, true);
    if (!joints) e.regress(e.rverts.data(), false, e.rjreg.data(), B);
    if (it == num_iter) break;
    std::vector<float> Gprev = e.G;
    e.k1(tj_rot, joints ? e.rjoints.data() : e.rjreg.data(), false, Gprev.data(), jw, true, false, false,
         false, false);
  }
  const bool wv = joints ? (vw && jw) : (vw != nullptr);
  const float* jwe = (joints && vw && jw) ? jw : nullptr;
  std::vector<float> red(8);
  for (int b = 0; b < B; ++b) {
    sf::scale_trans_stage(cx, t.J, t.V, t.Vp, red.data(), e.tvs.data() + (size_t)b * 3 * t.Vp,
                          e.rverts.data() + (size_t)b * 3 * t.Vp, wv ? e.vws.data() + (size_t)b * t.Vp : nullptr,
                          joints ? e.tjc.data() + (size_t)b * t.J * 3 : nullptr,
                          e.rjoints.data() + (size_t)b * t.J * 3, jwe ? jwe + (size_t)b * t.J : nullptr,
                          scale_fit != 0, e.psum.data() + (size_t)b * t.J * sf::kPsum,
                          joints ? nullptr : e.rjreg.data() + (size_t)b * t.J * 3, t.reg_rowsum.data(),
                          e.trans.data() + (size_t)b * 3, scale.data() + b);
    if (scale_fit && scale_out) scale_out[b] = scale[b];
    sf::refine_stage(cx, e.jt, e.sh, e.psum.data() + (size_t)b * t.J * sf::kPsum,
                     tj_rot + (size_t)b * t.J * 3,
                     (joints ? e.rjoints.data() : e.rjreg.data()) + (size_t)b * t.J * 3,
                     e.rjoints.data() + (size_t)b * t.J * 3, jw ? jw + (size_t)b * t.J : nullptr,
                     e.G.data() + (size_t)b * t.J * 9, e.beta.data() + (size_t)b * S,
                     e.trans.data() + (size_t)b * 3, e.mean.data() + (size_t)b * 3, final_adjust != 0,
                     pose + (size_t)b * t.J * 3, nullptr, trans + (size_t)b * 3, nullptr,
                     orient ? orient + (size_t)b * t.J * 9 : nullptr, nullptr,
                     scale_fit ? scale.data() + b : nullptr);
  }
  return 0;
}

template <int S, int KW>
int forward_impl(const sf::HostTables& t, const float* pose, const float* glob, const float* betas,
                 int nb, const float* trans, const float* kid, int B, float* verts, float* joints,
                 float* orient) {
  Emu<S, KW> e(t, B);
  HostCtx cx;
  const float zero3[3] = {0, 0, 0};
  for (int b = 0; b < B; ++b) {
    sf::forward_joint_stage(cx, e.jt, e.sh, pose ? pose + (size_t)b * t.J * 3 : nullptr,
                            glob ? glob + (size_t)b * t.J * 9 : nullptr,
                            betas ? betas + (size_t)b * nb : nullptr, betas ? nb : 0,
                            kid ? kid + b : nullptr, trans ? trans + (size_t)b * 3 : nullptr, e.rp.data() + (size_t)b * t.Kp,
                            e.jd_b(b), joints + (size_t)b * t.J * 3,
                            orient ? orient + (size_t)b * t.J * 9 : nullptr);
    for (int k = 0; k < t.J * 3; ++k)
      e.jb[(size_t)b * t.J * 4 + (k / 3) * 4 + k % 3] = e.jd_b(b)[(k / 3) * sf::jd_stride(S) + 9 + k % 3];
  }
  if (verts) {
    e.gemm();
    for (int b = 0; b < B; ++b)
      for (int i = 0; i < t.V; ++i) {
        float v[3];
        float bb[S];
        for (int q = 0; q < S; ++q) bb[q] = (betas && q < nb) ? betas[(size_t)b * nb + q] : 0.f;
        if (kid && t.n_kid) bb[S - 1] = kid[b];
        e.vertex(b, i, bb, S, trans ? trans + (size_t)b * 3 : zero3, v);
        for (int c = 0; c < 3; ++c) verts[((size_t)b * t.V + t.perm[i]) * 3 + c] = v[c];
      }
  }
  return 0;
}

}  // namespace

extern "C" {

const char* hostemu_last_error() { return g_err.c_str(); }

// proj_so3 / mat2rotvec / rotvec2mat / align_unit_vectors on n inputs (primitive goldens)
void hostemu_proj_so3(const float* A, float* R, int n) { for (int i = 0; i < n; ++i) sf::proj_so3(A + i * 9, R + i * 9); }
void hostemu_mat2rotvec(const float* R, float* rv, int n) { for (int i = 0; i < n; ++i) sf::mat2rotvec(R + i * 9, rv + i * 3); }
void hostemu_rotvec2mat(const float* rv, float* R, int n) { for (int i = 0; i < n; ++i) sf::rotvec2mat(rv + i * 3, R + i * 9); }
void hostemu_align(const float* a, const float* b, float* R, int n) { for (int i = 0; i < n; ++i) sf::align_unit_vectors(a + i * 3, b + i * 3, R + i * 9); }

int hostemu_fit(const smplfit_model_desc* d, const float* tv, const float* tj, const float* vw,
                const float* jw, int B, int num_iter, float reg, float reg2, float kid_reg,
                int final_adjust, float* pose, float* betas, float* trans, float* kid, float* orient,
                float* G0_out) {
  sf::HostTables t;
  bool unsup = false;
  g_err = sf::build_tables(*d, t, &unsup);
  if (!g_err.empty()) return -1;
  if (t.S == 10 && t.KW == 4)
    return fit_impl<10, 4>(t, tv, tj, vw, jw, B, num_iter, reg, reg2, kid_reg, final_adjust, pose, betas, trans, kid, orient, G0_out);
  if (t.S == 10 && t.KW == 8)
    return fit_impl<10, 8>(t, tv, tj, vw, jw, B, num_iter, reg, reg2, kid_reg, final_adjust, pose, betas, trans, kid, orient, G0_out);
  if (t.S == 16 && t.KW == 4)
    return fit_impl<16, 4>(t, tv, tj, vw, jw, B, num_iter, reg, reg2, kid_reg, final_adjust, pose, betas, trans, kid, orient, G0_out);
  if (t.S == 11 && t.KW == 4)
    return fit_impl<11, 4>(t, tv, tj, vw, jw, B, num_iter, reg, reg2, kid_reg, final_adjust, pose, betas, trans, kid, orient, G0_out);
  if (t.S == 17 && t.KW == 4)
    return fit_impl<17, 4>(t, tv, tj, vw, jw, B, num_iter, reg, reg2, kid_reg, final_adjust, pose, betas, trans, kid, orient, G0_out);
  g_err = "hostemu: unsupported (S, KW)";
  return -2;
}

// the cross-rank sum of a sharded share_beta fit for the NEXT hostemu_fit_warm calls of this thread
thread_local smplfit_share_allreduce_fn g_share_allreduce = nullptr;
thread_local void* g_share_user = nullptr;
void hostemu_set_share_allreduce(smplfit_share_allreduce_fn fn, void* user) {
  g_share_allreduce = fn;
  g_share_user = user;
}

int hostemu_fit_warm(const smplfit_model_desc* d, const float* tv, const float* tj, const float* vw,
                     const float* jw, int B, int num_iter, float reg, float reg2, float kid_reg,
                     int final_adjust, const float* init_pose, const float* init_betas, int init_nb,
                     const float* init_kid, int share_beta, int scale_mode, float scale_reg,
                     float* scale_out, float* pose, float* betas, float* trans, float* kid, float* orient) {
  g_warm.share_beta = share_beta;
  g_warm.share_allreduce = share_beta ? g_share_allreduce : nullptr;
  g_warm.share_user = g_share_user;
  g_warm.scale_mode = scale_mode;
  g_warm.scale_reg = scale_reg;
  g_warm.scale_out = scale_out;
  g_warm.pose = init_pose;
  g_warm.betas = init_betas;
  g_warm.nb = init_betas ? init_nb : 0;
  g_warm.kid = init_kid;
  const int rc = hostemu_fit(d, tv, tj, vw, jw, B, num_iter, reg, reg2, kid_reg, final_adjust, pose, betas,
                             trans, kid, orient, nullptr);
  g_warm = Warm();
  return rc;
}

int hostemu_fit_known_shape(const smplfit_model_desc* d, const float* betas, int nb, const float* kid,
                            const float* init_pose, const float* tv, const float* tj, const float* vw,
                            const float* jw, int B, int num_iter, int final_adjust, int scale_fit,
                            float* pose, float* trans, float* scale_out, float* orient) {
  sf::HostTables t;
  bool unsup = false;
  g_err = sf::build_tables(*d, t, &unsup);
  if (!g_err.empty()) return -1;
#define HE_KS(S_, KW_) \
  if (t.S == S_ && t.KW == KW_) \
    return known_shape_impl<S_, KW_>(t, betas, nb, kid, init_pose, tv, tj, vw, jw, B, num_iter, final_adjust, scale_fit, pose, trans, scale_out, orient)
  HE_KS(10, 4); HE_KS(10, 8); HE_KS(16, 4); HE_KS(11, 4); HE_KS(17, 4);
#undef HE_KS
  g_err = "hostemu: unsupported (S, KW)";
  return -2;
}

int hostemu_forward(const smplfit_model_desc* d, const float* pose, const float* glob,
                    const float* betas, int nb, const float* trans, const float* kid, int B,
                    float* verts, float* joints, float* orient) {
  sf::HostTables t;
  bool unsup = false;
  g_err = sf::build_tables(*d, t, &unsup);
  if (!g_err.empty()) return -1;
  if (t.S == 10 && t.KW == 4) return forward_impl<10, 4>(t, pose, glob, betas, nb, trans, kid, B, verts, joints, orient);
  if (t.S == 10 && t.KW == 8) return forward_impl<10, 8>(t, pose, glob, betas, nb, trans, kid, B, verts, joints, orient);
  if (t.S == 16 && t.KW == 4) return forward_impl<16, 4>(t, pose, glob, betas, nb, trans, kid, B, verts, joints, orient);
  if (t.S == 11 && t.KW == 4) return forward_impl<11, 4>(t, pose, glob, betas, nb, trans, kid, B, verts, joints, orient);
  if (t.S == 17 && t.KW == 4) return forward_impl<17, 4>(t, pose, glob, betas, nb, trans, kid, B, verts, joints, orient);
  g_err = "hostemu: unsupported (S, KW)";
  return -2;
}

}  // extern "C"
