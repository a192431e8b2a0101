// libsmplfit_hip.so — HIP kernels (gfx950 / CDNA4) and the C-ABI of include/smplfit.h.
//
// Kernel inventory.  Wave-per-instance kernels (every configuration; grid = instances unless noted):
//   k_center_sort_partsum(_lds)  K0  mean-centre targets, re-order vertices by body part (SoA), part
//                              sums against the template mesh      [HBM-bound streaming + wave sums]
//   k_joint_stage          K1  part rotations (SO(3) projections, swing-twist), shape prologue
//                              (FK + beta-Jacobian, pose feature, joint normal equations)  [1 wave]
//   k_posedirs_gemm(_as)   K2  v_posed = v_template + pose_feature . posedirs, fp32 MFMA 32x32x2,
//                              A-stationary (SMPL) or 128x128 LDS-tiled; plain or instance-innermost
//                              output                                                    [MFMA-bound]
//   k_shape_accum          K3  per-vertex blended rotation / position / shape Jacobian from LDS
//                              joint block, 98 normal-equation sums per instance         [VALU-bound]
//   k_residual, k_pair_gram    the same block in pair-Gram form (SMPLFIT_SHAPE_FORM=pair)
//   k_shape_solve          K4  fp64 centring + 10x10 Cholesky + translation (share_beta: + k_share_reduce;
//                              scale options: k_scale_extras + k_shape_solve_scaled + k_scale_refs)
//   k_lbs_partsum          K5  vertices at the solved shape (LBS) fused with the part sums of the
//                              next rotation pass — the re-evaluated mesh never reaches HBM
//   k_refine_epilogue      K6  dependent rotation refinement + relative rotations + log map [1 wave]
//   k_forward_joint, k_lbs_partsum<MODE 2>   BodyModel.forward;  k_scale_trans  known-shape alignment
// Batch-major kernels (LANE = INSTANCE; the default vertex block where they apply, see bm_applies):
//   k_transpose_targets, k_residual_bm, k_pair_gram_bm, k_gram_combine_bm, k_lbs_partsum_bm,
//   k_psum_combine — grid = (vertex group | unit chunk) x instance blocks of 64.
// Everything is enqueued on the caller's stream; no host synchronisation, no allocation.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/smplfit.h"
#include "sf_stages.h"
#include "sf_tables.h"

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}

#define SF_HIP_TRY(expr)                                                                   \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess)                                                                  \
      return fail(SMPLFIT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));     \
  } while (0)

// ------------------------------------------------------------------------------------------------
// device-side model
// ------------------------------------------------------------------------------------------------
struct DevModel {
  int V, J, S, P, Vp, Kp, KW, n_used, nseg;
  sf::JointTabs jt;
  const int32_t* perm;      // (Vp)
  const int32_t* segments;  // (nseg,3)
  const int32_t* part_seg_start;  // (J+1) first segment of each part (empty range: unused part)
  const float *vt, *dm, *sd, *wval, *pdT, *pdSw, *vtN, *j_template, *cpackA, *cpackB, *gblob;
  const int32_t* gtiles;  // (ngt,3) start, count, part
  int ngt;
  const uint32_t* widx;
  const int32_t *reg_start, *reg_slot;
  const float* reg_val;
  const float* reg_rowsum;
  // batch-major vertex kernels (HostTables::groups / brec / bwd)
  int ngroups, ngroups_used;
  const int32_t* groups;  // (ngroups, kGroupRec): start, count, part, used, nq, joints[12]
  const float* brec;
  const float* pair_c1x;
  // per joint: the resP rows (group * kResRec + 16 + 3 * slot) holding its residual moments
  const int32_t *mb_start, *mb_row;
};

constexpr int kGroupRec = 5 + sf::kGroupJoints;

}  // namespace

constexpr int kMaxChunks = 4;  // batch chunks of one fit call run on the caller's stream + 3 side streams

struct smplfit_handle {
  sf::HostTables t;
  DevModel d{};
  std::vector<void*> allocs;
  bool has_device = false;
  // fork/join resources of the chunked fit (see smplfit_fit_f32); guarded by `mu`
  hipStream_t side[kMaxChunks - 1] = {};
  hipEvent_t ev_fork = nullptr, ev_join[kMaxChunks - 1] = {};
  bool have_streams = false;
  mutable std::mutex mu;
};

namespace {

struct DevCtx {
  int lane, n;
  __device__ __forceinline__ void sync() const { __syncthreads(); }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// DPP wave-64 sum (6 VALU ops, no LDS traffic); the total lands in lane 63.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_step(float v) {
  const int x = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false);
  return v + __builtin_bit_cast(float, x);
}
__device__ __forceinline__ float wave_sum_last(float v) {
  v = dpp_step<0x111, 0xf>(v);  // row_shr:1
  v = dpp_step<0x112, 0xf>(v);  // row_shr:2
  v = dpp_step<0x114, 0xf>(v);  // row_shr:4
  v = dpp_step<0x118, 0xf>(v);  // row_shr:8   -> lane 15 of each row holds the row total
  v = dpp_step<0x142, 0xa>(v);  // row_bcast:15 into rows 1,3
  v = dpp_step<0x143, 0xc>(v);  // row_bcast:31 into rows 2,3 -> lane 63 holds the wave total
  return v;
}

// Per-call workspace carve (device pointers).
struct Workspace {
  float* tvs;      // (B,3,Vp)   centred targets, sorted slots, SoA
  float* vws;      // (B,Vp)     vertex weights, sorted slots (when given)
  float* vposed;   // (Mp,3*Vp)  GEMM output
  float* rp;       // (Mp,Kp)    pose features (GEMM A)
  float* mean;     // (B,3)
  float* tjc;      // (B,J,3)    centred target joints
  float* psum;     // (B,J,16)
  float* G;        // (B,J,9)
  float* jd;       // (B,J,jd_stride)
  float* pext;     // (B,J,3,S+1)
  float* gramj;    // (B,NE+1)
  double* gramv;   // (B,NE+1)
  float* beta;     // (B,S)
  float* trans;    // (B,3)
  float* jb;       // (B,J,4)
  float* rjoints;  // (B,J,3)
  float* rverts;   // (B,3,Vp) re-evaluated vertices (joints-omitted path only)
  float* tjreg;    // (B,J,3) regressed target joints (joints-omitted path)
  float* rjreg;    // (B,J,3) regressed reference joints
  float* mbj;      // (B,J,3) per-joint residual moments (pair-Gram form)
  float* scale;    // (B) scale_corr of the known-shape fit
  float* regref;   // (B,S) ridge reference of the warm-started fit
  double* cen;     // (B, S*S+S) centred regularised systems of a share_beta fit; row B = their sum
  float* vextra;   // (B,32) extra vertex sums of the scaled solve (scale_extras_vertex)
  float* beta_out; // (B,S) undivided shape of the scaled solve (ws.beta holds the evaluated one)
  float* tjs;      // (B,J,3) target joints times the scale (scale_target refinement)
  // batch-major path: streams with the instance index innermost (lane = instance reads coalesce)
  float* vpT;      // (Mp/64, 3*Vp, 64) v_posed, written by the GEMM
  float* tT;       // (Mp/64, 3*Vp, 64) centred targets, transposed from tvs
  float* psumP;    // (ngroups, 16, Mp) part sums per vertex group
  float* resP;     // (ngroups, kResRec, Mp) residual-pass sums per vertex group
  float* gramP;    // (kGramChunks, NG, Mp) pair-Gram partial sums
};

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

size_t carve(const sf::HostTables& t, int B, char* base, Workspace* w) {
  const size_t Mp = align_up((size_t)B, 128);
  const size_t Vp = t.Vp, J = t.J, S = t.S, NE1 = t.ne() + 1;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off = align_up(off + bytes, 256);
    return base ? base + o : nullptr;
  };
  Workspace ws;
  ws.tvs = (float*)take((size_t)B * 3 * Vp * 4);
  ws.vws = (float*)take((size_t)B * Vp * 4);
  ws.vposed = (float*)take(Mp * 3 * Vp * 4);
  ws.rp = (float*)take(Mp * t.Kp * 4);
  ws.mean = (float*)take((size_t)B * 3 * 4);
  ws.tjc = (float*)take((size_t)B * J * 3 * 4);
  ws.psum = (float*)take((size_t)B * J * sf::kPsum * 4);
  ws.G = (float*)take((size_t)B * J * 9 * 4);
  ws.jd = (float*)take((size_t)B * J * sf::jd_stride(S) * 4);
  ws.pext = (float*)take((size_t)B * J * 3 * (S + 1) * 4);
  ws.gramj = (float*)take((size_t)B * NE1 * 4);
  ws.gramv = (double*)take((size_t)B * NE1 * 8);
  ws.beta = (float*)take((size_t)B * S * 4);
  ws.trans = (float*)take((size_t)B * 3 * 4);
  ws.jb = (float*)take((size_t)B * J * 4 * 4);
  ws.rjoints = (float*)take((size_t)B * J * 3 * 4);
  ws.rverts = (float*)take((size_t)B * 3 * Vp * 4);
  ws.tjreg = (float*)take((size_t)B * J * 3 * 4);
  ws.rjreg = (float*)take((size_t)B * J * 3 * 4);
  ws.mbj = (float*)take((size_t)B * J * 3 * 4);
  ws.scale = (float*)take((size_t)B * 4);
  ws.regref = (float*)take((size_t)B * S * 4);
  ws.cen = (double*)take(((size_t)B + 1) * (S * S + S) * 8);
  ws.vextra = (float*)take((size_t)B * 32 * 4);
  ws.beta_out = (float*)take((size_t)B * S * 4);
  ws.tjs = (float*)take((size_t)B * J * 3 * 4);
  ws.vpT = (float*)take(Mp * 3 * Vp * 4);
  ws.tT = (float*)take(Mp * 3 * Vp * 4);
  ws.psumP = (float*)take((size_t)t.groups.size() * 16 * Mp * 4);
  ws.resP = (float*)take((size_t)t.groups.size() * (16 + 3 * sf::kGroupJoints) * Mp * 4);
  ws.gramP = (float*)take((size_t)32 * (NE1 - 1) * Mp * 4);  // kGramChunks x NG (<= NE) x Mp
  if (w) *w = ws;
  return off;
}

// ------------------------------------------------------------------------------------------------
// K0: centre + sort + template part sums.  grid B, block 256.
// reference: fit() centring bodyfitter.py:355-361; _part_sums :235-280 against default_mesh_tf.
// dynamic LDS: 4 waves x J x 16 floats + 20 (no static LDS: keeps the dynamic base 16-B aligned).
// ------------------------------------------------------------------------------------------------
template <bool WEIGHTED>
__global__ __launch_bounds__(256) void k_center_sort_partsum(DevModel m, const float* __restrict__ tv,
                                                             const float* __restrict__ tj,
                                                             const float* __restrict__ vw,
                                                             Workspace ws) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int V = m.V, J = m.J, Vp = m.Vp;
  float(*red)[4] = reinterpret_cast<float(*)[4]>(smem + 4 * J * sf::kPsum);  // [4][4]
  float* mu = smem + 4 * J * sf::kPsum + 16;                                // [4]
  const float* tvb = tv + (size_t)b * V * 3;
  // ---- mean over the V (+J) points
  float sx = 0.f, sy = 0.f, sz = 0.f;
  for (int v = tid; v < V; v += 256) {
    sx += tvb[v * 3];
    sy += tvb[v * 3 + 1];
    sz += tvb[v * 3 + 2];
  }
  if (tj) {
    for (int j = tid; j < J; j += 256) {
      sx += tj[((size_t)b * J + j) * 3];
      sy += tj[((size_t)b * J + j) * 3 + 1];
      sz += tj[((size_t)b * J + j) * 3 + 2];
    }
  }
  sx = wave_sum(sx);
  sy = wave_sum(sy);
  sz = wave_sum(sz);
  if (lane == 0) {
    red[wave][0] = sx;
    red[wave][1] = sy;
    red[wave][2] = sz;
  }
  for (int k = tid; k < 4 * J * sf::kPsum; k += 256) smem[k] = 0.f;
  __syncthreads();
  if (tid < 3) {
    const float n = (float)(V + (tj ? J : 0));
    const float s = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    mu[tid] = s / n;
    ws.mean[b * 3 + tid] = s / n;
  }
  __syncthreads();
  const float m0 = mu[0], m1 = mu[1], m2 = mu[2];
  if (tj)
    for (int k = tid; k < J * 3; k += 256)
      ws.tjc[(size_t)b * J * 3 + k] = tj[(size_t)b * J * 3 + k] - mu[k % 3];
  // ---- used parts: gather, centre, store sorted SoA, accumulate part sums per segment
  float* tvs = ws.tvs + (size_t)b * 3 * Vp;
  float* vws = ws.vws + (size_t)b * Vp;
  const int s_begin = (int)((long)m.nseg * wave / 4), s_end = (int)((long)m.nseg * (wave + 1) / 4);
  float acc[sf::kPsum];
#pragma unroll
  for (int k = 0; k < sf::kPsum; ++k) acc[k] = 0.f;
  for (int s = s_begin; s < s_end; ++s) {
    const int start = m.segments[s * 3], count = m.segments[s * 3 + 1], part = m.segments[s * 3 + 2];
    if (lane < count) {
      const int i = start + lane, o = m.perm[i];
      const float t[3] = {tvb[o * 3] - m0, tvb[o * 3 + 1] - m1, tvb[o * 3 + 2] - m2};
      tvs[i] = t[0];
      tvs[Vp + i] = t[1];
      tvs[2 * Vp + i] = t[2];
      const float a[3] = {m.dm[i], m.dm[Vp + i], m.dm[2 * Vp + i]};
      float w = 1.f;
      if (WEIGHTED) {
        w = vw[(size_t)b * V + o];
        vws[i] = w;
      }
      sf::partsum_vertex(t, a, w, WEIGHTED, acc);
    }
    const bool flush = (s + 1 == s_end) || (m.segments[(s + 1) * 3 + 2] != part);
    if (flush) {
#pragma unroll
      for (int k = 0; k < sf::kPsum; ++k) {
        const float r = wave_sum(acc[k]);
        if (lane == 0) smem[(wave * J + part) * sf::kPsum + k] = r;
        acc[k] = 0.f;
      }
    }
  }
  // ---- the remaining (unused-part and padding) slots: store only
  for (int i = m.n_used + tid; i < Vp; i += 256) {
    const int o = m.perm[i];
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, w = 0.f;
    if (o >= 0) {
      t0 = tvb[o * 3] - m0;
      t1 = tvb[o * 3 + 1] - m1;
      t2 = tvb[o * 3 + 2] - m2;
      if (WEIGHTED) w = vw[(size_t)b * V + o];
    }
    tvs[i] = t0;
    tvs[Vp + i] = t1;
    tvs[2 * Vp + i] = t2;
    if (WEIGHTED) vws[i] = w;
  }
  __syncthreads();
  for (int k = tid; k < J * sf::kPsum; k += 256)
    ws.psum[(size_t)b * J * sf::kPsum + k] =
        (smem[k] + smem[J * sf::kPsum + k]) + (smem[2 * J * sf::kPsum + k] + smem[3 * J * sf::kPsum + k]);
}

// ------------------------------------------------------------------------------------------------
// K0 (LDS-staged form): one 1024-thread workgroup per instance.  The instance's (V,3) target row is
// read from HBM exactly once, fully coalesced, into LDS (83 KB SMPL / 126 KB SMPL-X); the mean, the
// gather into part-sorted SoA order and the per-part sums against the template all run out of LDS.
// (Real SMPL vertex order is not part-sorted: gathering 12-byte vertices straight from global
// memory pulled ~8x the row through the fabric.)   part_seg_start: (J+1) first segment of each part.
// dynamic LDS: 3 VL floats + 64, VL = vertices staged in LDS.  For SMPL the whole row (82.7 KB) would
// allow one workgroup per CU only; staging the first VL = 6784 vertices (80 KB) and reading the last
// 1.5 % of the row from L2 in the gather lets TWO workgroups share a CU, so one loads while the other
// gathers / sums.
// ------------------------------------------------------------------------------------------------
template <bool WEIGHTED>
__global__ __launch_bounds__(1024) void k_center_sort_partsum_lds(DevModel m, const float* __restrict__ tv,
                                                                 const float* __restrict__ tj,
                                                                 const float* __restrict__ vw,
                                                                 Workspace ws, int VL) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int V = m.V, J = m.J, Vp = m.Vp, n3 = 3 * V, nl = 3 * VL;
  float* raw = smem;                       // [3 VL]
  float* red = smem + ((nl + 3) & ~3);     // [16][3] + mu[3]
  const float* tvb = tv + (size_t)b * n3;
  auto at = [&](int k) { return k < nl ? raw[k] : tvb[k]; };  // element k of the row
  // ---- coalesced row load (rows are 8-byte aligned: 3V*4 is a multiple of 8 when V is even)
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  if ((n3 & 1) == 0) {
    const float2* src = reinterpret_cast<const float2*>(tvb);
    float2* dst = reinterpret_cast<float2*>(raw);
    for (int k = tid; k < n3 / 2; k += 1024) {
      const float2 x = src[k];
      if (2 * k + 1 < nl) dst[k] = x;  // nl is even here (VL even or VL == V)
      const int r = (2 * k) % 3;  // coordinate of x.x; x.y is (r+1)%3
      s0 += (r == 0 ? x.x : 0.f) + (r == 2 ? x.y : 0.f);
      s1 += (r == 1 ? x.x : 0.f) + (r == 0 ? x.y : 0.f);
      s2 += (r == 2 ? x.x : 0.f) + (r == 1 ? x.y : 0.f);
    }
  } else {
    for (int k = tid; k < n3; k += 1024) {
      const float x = tvb[k];
      if (k < nl) raw[k] = x;
      const int r = k % 3;
      s0 += r == 0 ? x : 0.f;
      s1 += r == 1 ? x : 0.f;
      s2 += r == 2 ? x : 0.f;
    }
  }
  if (tj) {
    for (int j = tid; j < J; j += 1024) {
      s0 += tj[((size_t)b * J + j) * 3];
      s1 += tj[((size_t)b * J + j) * 3 + 1];
      s2 += tj[((size_t)b * J + j) * 3 + 2];
    }
  }
  s0 = wave_sum_last(s0);
  s1 = wave_sum_last(s1);
  s2 = wave_sum_last(s2);
  if (lane == 63) {
    red[wave * 3] = s0;
    red[wave * 3 + 1] = s1;
    red[wave * 3 + 2] = s2;
  }
  __syncthreads();
  if (tid < 3) {
    float s = 0.f;
    for (int w = 0; w < 16; ++w) s += red[w * 3 + tid];
    const float mean = s / (float)(V + (tj ? J : 0));
    red[48 + tid] = mean;
    ws.mean[b * 3 + tid] = mean;
  }
  __syncthreads();
  const float m0 = red[48], m1 = red[49], m2 = red[50];
  if (tj)
    for (int k = tid; k < J * 3; k += 1024)
      ws.tjc[(size_t)b * J * 3 + k] = tj[(size_t)b * J * 3 + k] - red[48 + k % 3];
  // ---- gather from LDS into sorted SoA order, coalesced stores
  float* tvs = ws.tvs + (size_t)b * 3 * Vp;
  float* vws = ws.vws + (size_t)b * Vp;
  for (int i = tid; i < Vp; i += 1024) {
    const int o = m.perm[i];
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, w = 0.f;
    if (o >= 0) {
      t0 = at(o * 3) - m0;
      t1 = at(o * 3 + 1) - m1;
      t2 = at(o * 3 + 2) - m2;
      if (WEIGHTED) w = vw[(size_t)b * V + o];
    }
    tvs[i] = t0;
    tvs[Vp + i] = t1;
    tvs[2 * Vp + i] = t2;
    if (WEIGHTED) vws[i] = w;
  }
  // ---- per-part sums against the template: one wave per part
  for (int p = wave; p < J; p += 16) {
    const int s_begin = m.part_seg_start[p], s_end = m.part_seg_start[p + 1];
    float* ps = ws.psum + ((size_t)b * J + p) * sf::kPsum;
    float acc[sf::kPsum];
#pragma unroll
    for (int k = 0; k < sf::kPsum; ++k) acc[k] = 0.f;
    for (int s = s_begin; s < s_end; ++s) {
      const int start = m.segments[s * 3], count = m.segments[s * 3 + 1];
      if (lane < count) {
        const int i = start + lane, o = m.perm[i];
        const float t[3] = {at(o * 3) - m0, at(o * 3 + 1) - m1, at(o * 3 + 2) - m2};
        const float a[3] = {m.dm[i], m.dm[Vp + i], m.dm[2 * Vp + i]};
        sf::partsum_vertex(t, a, WEIGHTED ? vw[(size_t)b * V + o] : 1.f, WEIGHTED, acc);
      }
    }
#pragma unroll
    for (int k = 0; k < sf::kPsum; ++k) {
      const float r = wave_sum_last(acc[k]);
      if (lane == 63) ps[k] = r;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// sparse post-LBS joint regression (joints-omitted path): out[b][j] = sum_k reg_val * src[b][:, slot]
// reference: bodyfitter.py:1342-1344.  grid B, block 64.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_regress_joints(DevModel m, const float* __restrict__ src,
                                                       float* __restrict__ out) {
  const int b = blockIdx.x;
  const float* s = src + (size_t)b * 3 * m.Vp;
  for (int j = threadIdx.x; j < m.J; j += 64) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int k = m.reg_start[j]; k < m.reg_start[j + 1]; ++k) {
      const int i = m.reg_slot[k];
      const float r = m.reg_val[k];
      a0 += r * s[i];
      a1 += r * s[m.Vp + i];
      a2 += r * s[2 * m.Vp + i];
    }
    out[((size_t)b * m.J + j) * 3] = a0;
    out[((size_t)b * m.J + j) * 3 + 1] = a1;
    out[((size_t)b * m.J + j) * 3 + 2] = a2;
  }
}

// ------------------------------------------------------------------------------------------------
// K1: joint stage.  grid B, block 64.  dynamic LDS = joint_scratch_floats.
// ------------------------------------------------------------------------------------------------
struct JointStageArgs {
  const float* tj;       // (B,J,3) centred target joints (given or regressed)
  const float* rj;       // (B,J,3) reference joints, or (J,3) template when rj_shared
  int rj_shared;
  const float* Gprev;    // (B,J,9) or null
  const float* jw;       // (B,J) or null
  int fit_rotations, do_prologue, joint_block, joint_block_weighted, vertex_sa_closed_form;
};

__global__ __launch_bounds__(64) void k_joint_stage(DevModel m, JointStageArgs a, Workspace ws) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.x, J = m.J, S = m.S;
  DevCtx cx{(int)threadIdx.x, 64};
  sf::JointScratch sh = sf::carve_joint_scratch(smem, J, S, 0);
  const int NE1 = sf::ne_size(S) + 1;
  sf::joint_stage(cx, m.jt, sh, ws.psum + (size_t)b * J * sf::kPsum, a.tj + (size_t)b * J * 3,
                  a.rj_shared ? a.rj : a.rj + (size_t)b * J * 3,
                  a.Gprev ? a.Gprev + (size_t)b * J * 9 : nullptr,
                  a.jw ? a.jw + (size_t)b * J : nullptr, a.fit_rotations != 0, a.do_prologue != 0,
                  a.joint_block != 0,
                  a.joint_block_weighted != 0, a.vertex_sa_closed_form != 0, ws.G + (size_t)b * J * 9,
                  ws.rp + (size_t)b * m.Kp, ws.jd + (size_t)b * J * sf::jd_stride(S),
                  ws.pext + (size_t)b * J * 3 * (S + 1), ws.gramj + (size_t)b * NE1);
}

// ------------------------------------------------------------------------------------------------
// K2: v_posed[Mp][N] = bias[N] + A[Mp][Kp] . Bm[Kp][N]   (fp32 MFMA 32x32x2, exact f32)
// reference: bodyfitter.py:913-916 einsum('vcp,bp->bvc').  N = 3*Vp, column n = c*Vp + slot.
// 128x128 block tile, 4 waves x (64x64), K step 16, register-prefetched double-buffered LDS.
// grid = (N/128) * (Mp/128), n-tile major so the blocks of one posedirs column tile run together.
// ------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <bool TRANSPOSED>
__global__ __launch_bounds__(256) void k_posedirs_gemm(const float* __restrict__ A,
                                                       const float* __restrict__ Bm,
                                                       const float* __restrict__ bias,
                                                       float* __restrict__ C, int Mp, int N, int Kp) {
  __shared__ float As[2][128][17];
  __shared__ __attribute__((aligned(16))) float Bs[2][16][128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int mtiles = Mp / 128;
  const int ntile = blockIdx.x / mtiles, mtile = blockIdx.x % mtiles;
  const int m0 = mtile * 128, n0 = ntile * 128;
  const int wm = (wave & 1) * 64, wn = (wave >> 1) * 64;
  const int l31 = lane & 31, lk = lane >> 5;

  f32x16 acc[2][2];
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const float bv = bias[n0 + wn + ni * 32 + l31];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r)  // transposed: the tile's rows are the columns n
        acc[mi][ni][r] = TRANSPOSED ? bias[n0 + wn + ni * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk] : bv;
  }
  // global -> register staging: A 128x16 (2 float4 / thread), B 16x128 (2 float4 / thread)
  const int a_row = tid >> 2, a_kc = (tid & 3) * 4;
  const int b_row = tid >> 5, b_nc = (tid & 31) * 4;
  float4 ra[2], rb[2];
  auto gload = [&](int k0) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      ra[h] = *reinterpret_cast<const float4*>(A + (size_t)(m0 + a_row + 64 * h) * Kp + k0 + a_kc);
      rb[h] = *reinterpret_cast<const float4*>(Bm + (size_t)(k0 + b_row + 8 * h) * N + n0 + b_nc);
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float* ap = &As[buf][a_row + 64 * h][a_kc];
      ap[0] = ra[h].x; ap[1] = ra[h].y; ap[2] = ra[h].z; ap[3] = ra[h].w;
      *reinterpret_cast<float4*>(&Bs[buf][b_row + 8 * h][b_nc]) = rb[h];
    }
  };
  const int nk = Kp / 16;
  gload(0);
  sstore(0);
  __syncthreads();
  for (int it = 0; it < nk; ++it) {
    const int buf = it & 1;
    if (it + 1 < nk) gload((it + 1) * 16);
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const float a0 = As[buf][wm + l31][2 * kk + lk];
      const float a1 = As[buf][wm + 32 + l31][2 * kk + lk];
      const float b0 = Bs[buf][2 * kk + lk][wn + l31];
      const float b1 = Bs[buf][2 * kk + lk][wn + 32 + l31];
      if (TRANSPOSED) {  // operands swapped: instances become the columns of the result tile
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0, a0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b1, a0, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0, a1, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b1, a1, acc[1][1], 0, 0, 0);
      } else {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      }
    }
    if (it + 1 < nk) sstore(buf ^ 1);
    __syncthreads();
  }
  // C layout of 32x32: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (TRANSPOSED) {  // batch-major: [instance / 64][n][64]
          const int inst = m0 + wm + mi * 32 + l31;
          const int n = n0 + wn + ni * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
          C[((size_t)(inst >> 6) * N + n) * 64 + (inst & 63)] = acc[mi][ni][r];
        } else {
          const int row = m0 + wm + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
          const int col = n0 + wn + ni * 32 + l31;
          C[(size_t)row * N + col] = acc[mi][ni][r];
        }
      }
}

constexpr int kNW = 4;  // instances (= waves) per workgroup in the vertex kernels

// ------------------------------------------------------------------------------------------------
// K2, A-stationary form (Kp <= 256): each wave keeps its 32 instances' whole pose-feature rows in
// NK2 = Kp/2 VGPRs (lane = (instance, k parity), the 32x32x2 A operand) and streams 32-column tiles
// of posedirs through a double-buffered LDS tile shared by the 4 waves (128 instances) of the
// workgroup.  Per tile a wave issues NK2 back-to-back MFMAs on one accumulator (issue interval =
// dependent latency = 64 cycles), fed by 16-byte LDS reads: one barrier per 6656 MFMA cycles instead
// of one per 512 in the generic tiled kernel.  Bsw: (N/32, 32, Kp) pre-transposed tiles.
// grid = (nchunk, Mp/128); workgroup y handles instances [128y, 128y+128), chunk x a run of tiles.
// ------------------------------------------------------------------------------------------------
// TRANSPOSED: the MFMA operands swap roles (instances become the columns of the 32x32 result) and the
// output is written instance-innermost, C[n][Mp] — the layout of the batch-major vertex kernels.
template <int NK2, bool TRANSPOSED>
__global__ __launch_bounds__(256, 2) void k_posedirs_gemm_as(const float* __restrict__ A,
                                                          const float* __restrict__ Bsw,
                                                          const float* __restrict__ bias,
                                                          float* __restrict__ C, int N,
                                                          int tiles_per_chunk, int Mp) {
  constexpr int KP = 2 * NK2, RS = KP + 4;  // LDS row stride: +4 floats keeps 16 lanes on 16 slots
  constexpr int TILE_F4 = 32 * KP / 4;      // float4 per tile in global memory
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [2][32][RS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lk = lane >> 5;
  const int m0 = (blockIdx.y * 4 + wave) * 32;
  const int ntiles = N / 32;
  const int t_begin = blockIdx.x * tiles_per_chunk;
  const int t_end = min(t_begin + tiles_per_chunk, ntiles);
  // A fragment: this lane's instance row, its k parity (contiguous thanks to rp_pos)
  float a[NK2];
  {
    const float4* src = reinterpret_cast<const float4*>(A + (size_t)(m0 + l31) * KP + lk * NK2);
#pragma unroll
    for (int q = 0; q < NK2 / 4; ++q) {
      const float4 v = src[q];
      a[4 * q] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
    }
  }
  // global -> register -> LDS staging of the next tile, in named registers (an indexed array would
  // live in scratch); TILE_F4 <= 7 * 256
  static_assert(TILE_F4 <= 7 * 256, "tile too large for the staging registers");
  float4 s0, s1, s2, s3, s4, s5, s6;
#define SF_STAGE_ALL(OP) OP(0, s0) OP(1, s1) OP(2, s2) OP(3, s3) OP(4, s4) OP(5, s5) OP(6, s6)
  auto gload = [&](int tile) {
    const float4* src = reinterpret_cast<const float4*>(Bsw) + (size_t)tile * TILE_F4;
#define SF_LD(q, r) if (tid + 256 * q < TILE_F4) r = src[tid + 256 * q];
    SF_STAGE_ALL(SF_LD)
#undef SF_LD
  };
  auto sstore = [&](int buf) {
    float* base = smem + (size_t)buf * 32 * RS;
#define SF_ST(q, r)                                                                        \
  if (tid + 256 * q < TILE_F4) {                                                           \
    const int f = tid + 256 * q;                                                           \
    *reinterpret_cast<float4*>(base + (f / (KP / 4)) * RS + 4 * (f % (KP / 4))) = r;      \
  }
    SF_STAGE_ALL(SF_ST)
#undef SF_ST
  };
#undef SF_STAGE_ALL
  if (t_begin >= t_end) return;
  gload(t_begin);
  sstore(0);
  __syncthreads();
  for (int t = t_begin; t < t_end; ++t) {
    const int buf = (t - t_begin) & 1;
    if (t + 1 < t_end) gload(t + 1);
    f32x16 acc;
    // C layout of 32x32: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    if (TRANSPOSED) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = bias[t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk];
    } else {
      const float bv = bias[t * 32 + l31];
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = bv;
    }
    const float* brow = smem + (size_t)buf * 32 * RS + l31 * RS + lk * NK2;
#pragma unroll
    for (int q = 0; q < NK2 / 4; ++q) {
      const float4 bq = *reinterpret_cast<const float4*>(brow + 4 * q);
      if (TRANSPOSED) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bq.x, a[4 * q], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bq.y, a[4 * q + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bq.z, a[4 * q + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bq.w, a[4 * q + 3], acc, 0, 0, 0);
      } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * q], bq.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * q + 1], bq.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * q + 2], bq.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * q + 3], bq.w, acc, 0, 0, 0);
      }
    }
    if (TRANSPOSED) {  // batch-major: [m0 / 64][n][64]; 32 consecutive instances per half wave
      float* ccol = C + ((size_t)(m0 >> 6) * N + t * 32) * 64 + (m0 & 63) + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        ccol[(size_t)((r & 3) + 8 * (r >> 2) + 4 * lk) * 64] = acc[r];
    } else {
      float* crow = C + (size_t)m0 * N + t * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        crow[(size_t)((r & 3) + 8 * (r >> 2) + 4 * lk) * N] = acc[r];
    }
    if (t + 1 < t_end) sstore(buf ^ 1);
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// K3: vertex block of the normal equations.  grid ceil(B/4), block 256: wave w fits instance
// 4*blockIdx + w; the 4 waves walk the vertex tiles in lockstep so that the per-vertex constants
// (shapedirs, skinning pairs: 144 B/vertex) are fetched ONCE per workgroup and staged through a
// double-buffered LDS tile; per-instance streams (targets, v_posed) are register-prefetched one
// tile ahead.  98 fp32 accumulators per lane, DPP wave reduction, fp64 result.
// dynamic LDS: 4 joint blocks + 2 x (64 x cstride) constants.
// ------------------------------------------------------------------------------------------------
template <int S, int KW, bool WEIGHTED>
__global__ __launch_bounds__(256, 2) void k_shape_accum(DevModel m, Workspace ws, int B) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NE = sf::ne_size(S), STRIDE = sf::jd_stride(S);
  constexpr int CS = sf::cpack_stride(S, KW), TILE_F4 = 64 * CS / 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int J = m.J, Vp = m.Vp, ntiles = Vp / 64;
  const int b_raw = blockIdx.x * kNW + wave;
  const int b = b_raw < B ? b_raw : B - 1;
  float* jd = smem + wave * J * STRIDE;
  float* cst = smem + kNW * J * STRIDE;  // [2][64*CS]
  float* priv = cst + 2 * 64 * CS + tid * 12;  // lane-private 12 floats (48-B stride: conflict-free b128)
  {
    const float4* src = reinterpret_cast<const float4*>(ws.jd + (size_t)b * J * STRIDE);
    float4* dst = reinterpret_cast<float4*>(jd);
    for (int k = lane; k < J * STRIDE / 4; k += 64) dst[k] = src[k];
  }
  // cooperative staging of the constants tile: TILE_F4 float4 over 256 threads (<= 3 each), kept in
  // named registers (an indexed array here ends up in scratch)
  const float4* cg = reinterpret_cast<const float4*>(m.cpackA);
  static_assert(TILE_F4 <= 1024, "constants tile too large for 4 float4 per thread");
  float4 c0, c1, c2, c3;
  auto cload = [&](int tile) {
    const float4* src = cg + (size_t)tile * TILE_F4;
    c0 = src[tid];
    if (TILE_F4 > 256 && tid + 256 < TILE_F4) c1 = src[tid + 256];
    if (TILE_F4 > 512 && tid + 512 < TILE_F4) c2 = src[tid + 512];
    if (TILE_F4 > 768 && tid + 768 < TILE_F4) c3 = src[tid + 768];
  };
  auto cstore = [&](int buf) {
    float4* dst = reinterpret_cast<float4*>(cst + buf * 64 * CS);
    dst[tid] = c0;
    if (TILE_F4 > 256 && tid + 256 < TILE_F4) dst[tid + 256] = c1;
    if (TILE_F4 > 512 && tid + 512 < TILE_F4) dst[tid + 512] = c2;
    if (TILE_F4 > 768 && tid + 768 < TILE_F4) dst[tid + 768] = c3;
  };
  const float* tvs = ws.tvs + (size_t)b * 3 * Vp;
  const float* vps = ws.vposed + (size_t)b * 3 * Vp;
  const float* vws = ws.vws + (size_t)b * Vp;
  float nx[7];
  auto sload = [&](int tile) {
    const int i = tile * 64 + lane;
    nx[0] = vps[i]; nx[1] = vps[Vp + i]; nx[2] = vps[2 * Vp + i];
    nx[3] = tvs[i]; nx[4] = tvs[Vp + i]; nx[5] = tvs[2 * Vp + i];
    nx[6] = WEIGHTED ? vws[i] : 1.f;
  };
  cload(0);
  sload(0);
  cstore(0);
  __syncthreads();
  float acc[NE + 1];
#pragma unroll
  for (int k = 0; k <= NE; ++k) acc[k] = 0.f;
  for (int tile = 0; tile < ntiles; ++tile) {
    const float vp[3] = {nx[0], nx[1], nx[2]};
    const float tv[3] = {nx[3], nx[4], nx[5]};
    const float wv = nx[6];
    if (tile + 1 < ntiles) {
      cload(tile + 1);
      sload(tile + 1);
    }
    if (WEIGHTED) acc[NE] += wv;
    sf::shape_accum_vertex<S, KW, WEIGHTED>(jd, cst + (tile & 1) * 64 * CS + lane * CS, vp, tv, wv,
                                            priv, acc);
    if (tile + 1 < ntiles) cstore((tile + 1) & 1);
    __syncthreads();
  }
  double* out = ws.gramv + (size_t)b * (NE + 1);
  constexpr int NG = sf::ne_ng(S);
#pragma unroll
  for (int k = 0; k <= NE; ++k) {
    // unit weights: SA comes from the joint stage in closed form, W = V (bodyfitter.py:1038-1040)
    const bool dead = !WEIGHTED && ((k >= NG + S && k < NG + 4 * S) || k == NE);
    float r = 0.f;
    if (!dead) r = wave_sum_last(acc[k]);
    if (lane == 63 && b_raw < B) out[k] = dead ? (k == NE ? (double)m.V : 0.0) : (double)r;
  }
}

// ------------------------------------------------------------------------------------------------
// K3r: residual pass of the pair-Gram form (unit weights).  grid ceil(B/4), block 256, wave =
// instance, lockstep over the residual tiles (part-aligned, <= 16 distinct joints) whose constants
// blob [vertex records | MFMA A-operand weights | joint ids] is staged through double-buffered LDS.
// Per vertex ~115 FMAs: blended R/T0, residual b, u = Rt^T b, r1 += S_v^T u, Sb += b.  The per-joint
// residual moments mb_j = sum_v w_vj b_v (a scatter over the 4 skinning joints of every vertex) run on
// the otherwise idle matrix pipe: D(16 joint slots x 16) += W(16 x 4 vertices) . b(4 x 16) with
// v_mfma_f32_16x16x4_f32, 16 instructions per 64-vertex tile, then 3x4 lanes add D into the wave's LDS
// bins — deterministic, no atomics.
// dynamic LDS: 4 x [jd | bbuf 256 | bins (J+1)*4] + 2 blobs.
// ------------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int S, int KW>
__global__ __launch_bounds__(256) void k_residual(DevModel m, Workspace ws, int B) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NE = sf::ne_size(S), NG = sf::ne_ng(S), STRIDE = sf::jd_stride(S);
  constexpr int CS = sf::cpack_stride(S, KW), BLOB = 64 * CS + 16 * 64 + 16, BLOB_F4 = BLOB / 4;
  static_assert(BLOB % 4 == 0 && BLOB_F4 <= 1280, "blob staging");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int J = m.J, Vp = m.Vp;
  const int b_raw = blockIdx.x * kNW + wave;
  const int b = b_raw < B ? b_raw : B - 1;
  const int per_wave = (J * STRIDE + 256 + (J + 1) * 4 + 3) / 4 * 4;
  float* jd = smem + wave * per_wave;
  float* bbuf = jd + J * STRIDE;  // [64][4]
  float* bins = bbuf + 256;       // [(J+1)][4]
  float* blob = smem + kNW * per_wave;  // [2][BLOB]
  {
    const float4* src = reinterpret_cast<const float4*>(ws.jd + (size_t)b * J * STRIDE);
    float4* dst = reinterpret_cast<float4*>(jd);
    for (int k = lane; k < J * STRIDE / 4; k += 64) dst[k] = src[k];
    for (int k = lane; k < (J + 1) * 4; k += 64) bins[k] = 0.f;
  }
  const float4* cg = reinterpret_cast<const float4*>(m.gblob);
  const float* tvs = ws.tvs + (size_t)b * 3 * Vp;
  const float* vps = ws.vposed + (size_t)b * 3 * Vp;
  // Two register stages (A, B), each holding one tile's constants-blob slice + this lane's target /
  // v_posed values; a stage is (re)loaded TWO tiles ahead of its use so that HBM / L2 latency is
  // covered by two tiles of work (one tile ahead left the loop latency-bound).  Named registers and
  // macros: a struct passed to a lambda ends up in scratch memory.
  float4 cA0, cA1, cA2, cA3, cA4, cB0, cB1, cB2, cB3, cB4;
  float vA0, vA1, vA2, tA0, tA1, tA2, vB0, vB1, vB2, tB0, tB1, tB2;
#define SF_ISSUE(P, tile_)                                                                     \
  do {                                                                                         \
    const float4* src_ = cg + (size_t)(tile_) * BLOB_F4;                                       \
    c##P##0 = src_[tid];                                                                       \
    if (BLOB_F4 > 256 && tid + 256 < BLOB_F4) c##P##1 = src_[tid + 256];                       \
    if (BLOB_F4 > 512 && tid + 512 < BLOB_F4) c##P##2 = src_[tid + 512];                       \
    if (BLOB_F4 > 768 && tid + 768 < BLOB_F4) c##P##3 = src_[tid + 768];                       \
    if (BLOB_F4 > 1024 && tid + 1024 < BLOB_F4) c##P##4 = src_[tid + 1024];                    \
    const int start_ = m.gtiles[(tile_) * 3], count_ = m.gtiles[(tile_) * 3 + 1];              \
    const int i_ = start_ + (lane < count_ ? lane : 0);                                        \
    const float keep_ = lane < count_ ? 1.f : 0.f; /* padding lanes: zero residual */          \
    v##P##0 = vps[i_] * keep_; v##P##1 = vps[Vp + i_] * keep_; v##P##2 = vps[2 * Vp + i_] * keep_; \
    t##P##0 = tvs[i_] * keep_; t##P##1 = tvs[Vp + i_] * keep_; t##P##2 = tvs[2 * Vp + i_] * keep_; \
  } while (0)
#define SF_COMMIT(P, buf_)                                                                     \
  do {                                                                                         \
    float4* dst_ = reinterpret_cast<float4*>(blob + (buf_) * BLOB);                            \
    dst_[tid] = c##P##0;                                                                       \
    if (BLOB_F4 > 256 && tid + 256 < BLOB_F4) dst_[tid + 256] = c##P##1;                       \
    if (BLOB_F4 > 512 && tid + 512 < BLOB_F4) dst_[tid + 512] = c##P##2;                       \
    if (BLOB_F4 > 768 && tid + 768 < BLOB_F4) dst_[tid + 768] = c##P##3;                       \
    if (BLOB_F4 > 1024 && tid + 1024 < BLOB_F4) dst_[tid + 1024] = c##P##4;                    \
  } while (0)
  float acc[S + 3];
#pragma unroll
  for (int k = 0; k < S + 3; ++k) acc[k] = 0.f;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int bsel = l15 < 3 ? l15 : 3;  // B-operand column: coordinate, columns >= 3 read the zero pad
  auto compute = [&](int buf, const float* vp, const float* tv) {
    const float* cur = blob + buf * BLOB;
    float bo[3];
    sf::residual_vertex<S, KW>(jd, cur + lane * CS, vp, tv, acc, bo);
    *reinterpret_cast<float4*>(bbuf + lane * 4) = make_float4(bo[0], bo[1], bo[2], 0.f);
    // scatter on the matrix pipe: D[slot][coord] += sum_vertex W[slot][vertex] b[vertex][coord]
    const float* wA = cur + 64 * CS;
    f32x4 D0 = {0.f, 0.f, 0.f, 0.f}, D1 = {0.f, 0.f, 0.f, 0.f};  // two chains: 40-cycle dependent latency
#pragma unroll
    for (int t = 0; t < 16; t += 2) {
      D0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[t * 64 + lane], bbuf[(4 * t + l4) * 4 + bsel], D0, 0, 0, 0);
      D1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[(t + 1) * 64 + lane], bbuf[(4 * t + 4 + l4) * 4 + bsel], D1, 0, 0, 0);
    }
    if (l15 < 3) {  // C/D layout 16x16: col = lane & 15, row = (lane >> 4) * 4 + r
      const int* slots = reinterpret_cast<const int*>(cur + 64 * CS + 16 * 64);
#pragma unroll
      for (int r = 0; r < 4; ++r) bins[slots[l4 * 4 + r] * 4 + l15] += D0[r] + D1[r];
    }
  };
  const int ntiles = m.ngt;
  SF_ISSUE(A, 0);
  if (ntiles > 1) SF_ISSUE(B, 1);
  SF_COMMIT(A, 0);
  __syncthreads();
  for (int tile = 0; tile < ntiles; tile += 2) {
    {
      const float vp[3] = {vA0, vA1, vA2}, tv[3] = {tA0, tA1, tA2};
      if (tile + 2 < ntiles) SF_ISSUE(A, tile + 2);
      compute(0, vp, tv);
      if (tile + 1 < ntiles) SF_COMMIT(B, 1);
      __syncthreads();
    }
    if (tile + 1 >= ntiles) break;
    {
      const float vp[3] = {vB0, vB1, vB2}, tv[3] = {tB0, tB1, tB2};
      if (tile + 3 < ntiles) SF_ISSUE(B, tile + 3);
      compute(1, vp, tv);
      if (tile + 2 < ntiles) SF_COMMIT(A, 0);
      __syncthreads();
    }
  }
#undef SF_ISSUE
#undef SF_COMMIT
  double* out = ws.gramv + (size_t)b * (NE + 1);
#pragma unroll
  for (int k = 0; k < S + 3; ++k) {
    const float r = wave_sum_last(acc[k]);
    if (lane == 63 && b_raw < B) out[k < S ? NG + k : NG + 4 * S + (k - S)] = (double)r;
  }
  if (b_raw < B) {
    if (lane == 63) out[NE] = (double)m.V;  // w_sum = num_vertices (bodyfitter.py:1038-1040)
    for (int k = lane; k < 3 * S; k += 64) out[NG + S + k] = 0.0;  // SA: closed form in the joint stage
    for (int k = lane; k < J * 3; k += 64) ws.mbj[(size_t)b * J * 3 + k] = bins[(k / 3) * 4 + k % 3];
  }
}

// ------------------------------------------------------------------------------------------------
// K3g: pair-Gram — the S x S Gramian of the vertex block from the rotations alone (unit weights).
// grid B, block 64 (one wave per instance, lanes = upper-triangle entries).  Independent of the
// targets and of the GEMM: it can run on a second stream next to K2.
// dynamic LDS: joint block + np*9 + J*3*S floats.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_pair_gram(DevModel m, Workspace ws) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.x, J = m.J, S = m.S, stride = sf::jd_stride(S);
  DevCtx cx{(int)threadIdx.x, 64};
  float* jd = smem;
  float* scratch = smem + J * stride;
  for (int k = threadIdx.x; k < J * stride; k += 64) jd[k] = ws.jd[(size_t)b * J * stride + k];
  __syncthreads();
  sf::pair_gram_stage(cx, m.jt, scratch, jd, ws.gramv + (size_t)b * (sf::ne_size(S) + 1));
}

// ------------------------------------------------------------------------------------------------
// K4: solve.  grid B, block 64.
// ------------------------------------------------------------------------------------------------
// mode 0: per-instance solve; 1 / 2: the two halves of a share_beta solve (sf::solve_stage)
__global__ __launch_bounds__(64) void k_shape_solve(DevModel m, Workspace ws, float beta_reg,
                                                    float beta_reg2, float kid_reg, int pair_form,
                                                    int use_ref, int mode = 0, int B = 0) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.x, J = m.J, S = m.S;
  DevCtx cx{(int)threadIdx.x, 64};
  const int NE1 = sf::ne_size(S) + 1;
  sf::solve_stage(cx, m.jt, smem, ws.gramv + (size_t)b * NE1, ws.gramj + (size_t)b * NE1,
                  ws.pext + (size_t)b * J * 3 * (S + 1), ws.jd + (size_t)b * J * sf::jd_stride(S),
                  pair_form ? ws.mbj + (size_t)b * J * 3 : nullptr, beta_reg, beta_reg2, kid_reg,
                  ws.beta + (size_t)b * S, ws.trans + (size_t)b * 3, ws.rjoints + (size_t)b * J * 3,
                  ws.jb + (size_t)b * J * 4, use_ref ? ws.regref + (size_t)b * S : nullptr, mode,
                  mode == 1 ? ws.cen + (size_t)b * (S * S + S) : ws.cen + (size_t)B * (S * S + S));
}

// ------------------------------------------------------------------------------------------------
// Scaled last solve of fit(scale_target / scale_fit) — a niche option, kept simple: one wave per
// instance for the extra vertex sums (joint block in LDS, records and streams straight from L2/HBM).
// ------------------------------------------------------------------------------------------------
template <int S, int KW>
__global__ __launch_bounds__(64) void k_scale_extras(DevModel m, Workspace ws, int weighted) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int STRIDE = sf::jd_stride(S), CS = sf::cpack_stride(S, KW), NX = S + sf::kScaleExtras;
  const int b = blockIdx.x, lane = threadIdx.x, J = m.J, Vp = m.Vp;
  for (int k = lane; k < J * STRIDE; k += 64) smem[k] = ws.jd[(size_t)b * J * STRIDE + k];
  __syncthreads();
  const float* tvs = ws.tvs + (size_t)b * 3 * Vp;
  const float* vps = ws.vposed + (size_t)b * 3 * Vp;
  float acc[NX];
#pragma unroll
  for (int k = 0; k < NX; ++k) acc[k] = 0.f;
  for (int i = lane; i < m.V; i += 64) {  // slots [0, V) are the real vertices
    const float vp[3] = {vps[i], vps[Vp + i], vps[2 * Vp + i]};
    const float tv[3] = {tvs[i], tvs[Vp + i], tvs[2 * Vp + i]};
    const float wv = weighted ? ws.vws[(size_t)b * Vp + i] : 1.f;
    sf::scale_extras_vertex<S, KW>(smem, m.cpackA + (size_t)i * CS, vp, tv, wv, acc);
  }
#pragma unroll
  for (int k = 0; k < NX; ++k) {
    const float r = wave_sum(acc[k]);
    if (lane == 0) ws.vextra[(size_t)b * 32 + k] = r;
  }
}

struct ScaledSolveArgs {
  const float* tj;  // centred target joints (joint block) or null
  const float* jw;  // joint weights entering the solve or null
  int joint_block, mode, pair_form, use_ref;
  float beta_reg, beta_reg2, kid_reg, scale_reg;
};

__global__ __launch_bounds__(64) void k_shape_solve_scaled(DevModel m, Workspace ws, ScaledSolveArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.x, J = m.J, S = m.S;
  DevCtx cx{(int)threadIdx.x, 64};
  const int NE1 = sf::ne_size(S) + 1;
  sf::scaled_solve_stage(cx, m.jt, smem, ws.gramv + (size_t)b * NE1, ws.gramj + (size_t)b * NE1,
                         ws.vextra + (size_t)b * 32, ws.pext + (size_t)b * J * 3 * (S + 1),
                         ws.jd + (size_t)b * J * sf::jd_stride(S),
                         a.pair_form ? ws.mbj + (size_t)b * J * 3 : nullptr,
                         a.tj ? a.tj + (size_t)b * J * 3 : nullptr, a.jw ? a.jw + (size_t)b * J : nullptr,
                         a.joint_block != 0, a.mode, a.beta_reg, a.beta_reg2, a.kid_reg, a.scale_reg,
                         a.use_ref ? ws.regref + (size_t)b * S : nullptr, ws.beta_out + (size_t)b * S,
                         ws.beta + (size_t)b * S, ws.trans + (size_t)b * 3, ws.scale + b,
                         ws.rjoints + (size_t)b * J * 3, ws.jb + (size_t)b * J * 4);
}

// After the scaled solve (and the LBS pass when the refinement follows): what the refinement and the
// epilogue see (bodyfitter.py:462-519).
//   scale_target: targets x s -> part sums raw, s_t x s, target joints x s; mean x s
//   scale_fit:    reference <- s reference + (1 - s) trans -> raw, s_a, reference joints; mean / s
// and the undivided shape back into ws.beta.  grid B, block 64.
__global__ __launch_bounds__(64) void k_scale_refs(DevModel m, Workspace ws, const float* __restrict__ tj_in,
                                                   int mode, int have_psum, int regressed) {
  const int b = blockIdx.x, lane = threadIdx.x, J = m.J, S = m.S;
  const float s = ws.scale[b];
  const float tr[3] = {(1.f - s) * ws.trans[b * 3], (1.f - s) * ws.trans[b * 3 + 1],
                       (1.f - s) * ws.trans[b * 3 + 2]};
  if (have_psum) {
    for (int j = lane; j < J; j += 64) {
      float* ps = ws.psum + ((size_t)b * J + j) * sf::kPsum;
      if (mode == 1) {
        for (int k = 0; k < 12; ++k) ps[k] *= s;  // raw and s_t are linear in the targets
        for (int c = 0; c < 3; ++c) ws.tjs[((size_t)b * J + j) * 3 + c] = s * tj_in[((size_t)b * J + j) * 3 + c];
      } else {
        for (int r = 0; r < 3; ++r)
          for (int c = 0; c < 3; ++c) ps[r * 3 + c] = s * ps[r * 3 + c] + ps[9 + r] * tr[c];
        for (int c = 0; c < 3; ++c) {
          ps[12 + c] = s * ps[12 + c] + ps[15] * tr[c];
          float* rj = ws.rjoints + ((size_t)b * J + j) * 3 + c;
          *rj = s * *rj + tr[c];
          if (regressed) {
            float* rr = ws.rjreg + ((size_t)b * J + j) * 3 + c;
            *rr = s * *rr + m.reg_rowsum[j] * tr[c];
          }
        }
      }
    }
  }
  if (lane < S) ws.beta[(size_t)b * S + lane] = ws.beta_out[(size_t)b * S + lane];
  if (lane < 3) ws.mean[b * 3 + lane] = mode == 1 ? ws.mean[b * 3 + lane] * s : ws.mean[b * 3 + lane] / s;
}

// share_beta: sum of the per-instance systems over the batch, instances in order (deterministic);
// one workgroup, thread = entry of the (S*S + S) record; the sum lands in row B of ws.cen.
__global__ __launch_bounds__(512) void k_share_reduce(Workspace ws, int B, int NC) {
  const int e = threadIdx.x;
  if (e >= NC) return;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  int b = 0;
  for (; b + 3 < B; b += 4) {
    a0 += ws.cen[(size_t)b * NC + e];
    a1 += ws.cen[(size_t)(b + 1) * NC + e];
    a2 += ws.cen[(size_t)(b + 2) * NC + e];
    a3 += ws.cen[(size_t)(b + 3) * NC + e];
  }
  for (; b < B; ++b) a0 += ws.cen[(size_t)b * NC + e];
  ws.cen[(size_t)B * NC + e] = (a0 + a1) + (a2 + a3);
}

// ------------------------------------------------------------------------------------------------
// K5: vertices at the solved shape + part sums against the targets.  grid ceil(B/4), block 256,
// wave = instance, lockstep over the part-aligned segments with LDS-staged constants (as K3).
// MODE 0: part sums only (joints given).  MODE 1: also store the vertices (sorted SoA) for the
// joint regression of the joints-omitted path.  MODE 2: store vertices in ORIGINAL order to `out`
// (B,V,3) (shape-solve entry point / forward) over ALL slots — no part sums.
// dynamic LDS: 4 x (joint block rows R|T0 + jb) + 2 x (64 x cstride) constants + 4 x 36.
// ------------------------------------------------------------------------------------------------
template <int S, int KW, bool WEIGHTED, int MODE, bool SOLVE>
__global__ __launch_bounds__(256) void k_lbs_partsum(DevModel m, Workspace ws, int B, int nb,
                                                     const float* __restrict__ beta_in,
                                                     const float* __restrict__ trans_in,
                                                     const float* __restrict__ kid_in,
                                                     float* __restrict__ out, float beta_reg,
                                                     float beta_reg2) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int STRIDE = sf::jd_stride(S);
  constexpr int CS = sf::cpack_stride(S, KW), TILE_F4 = 64 * CS / 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int J = m.J, Vp = m.Vp, V = m.V;
  const int b_raw = blockIdx.x * kNW + wave;
  const int b = b_raw < B ? b_raw : B - 1;
  const bool live = b_raw < B;
  const int per_wave =
      (J * STRIDE + J * 4 + 36 + (SOLVE ? sf::solve_scratch_floats(S) : 0) + 3) / 4 * 4;  // 16-B multiple
  float* jd = smem + wave * per_wave;
  float* jb = jd + J * STRIDE;
  float* sbeta = jb + J * 4;   // [32]
  float* strans = sbeta + 32;  // [4]
  float* cst = smem + kNW * per_wave;
  {
    const float4* src = reinterpret_cast<const float4*>(ws.jd + (size_t)b * J * STRIDE);
    float4* dst = reinterpret_cast<float4*>(jd);
    for (int k = lane; k < J * STRIDE / 4; k += 64) dst[k] = src[k];
  }
  if (SOLVE) {
    // K4 fused: this wave solves its instance's normal equations (fp64 centring + Cholesky) and
    // leaves beta / trans / per-joint skinning translations in LDS for the vertex loop below.
    // All four waves run the stage in lockstep (uniform barriers).
    DevCtx cx{lane, 64};
    const int NE1 = sf::ne_size(S) + 1;
    if (lane < 32) sbeta[lane] = 0.f;
    sf::solve_stage(cx, m.jt, strans + 4, ws.gramv + (size_t)b * NE1, ws.gramj + (size_t)b * NE1,
                    ws.pext + (size_t)b * J * 3 * (S + 1), ws.jd + (size_t)b * J * STRIDE, nullptr,
                    beta_reg, beta_reg2, beta_reg, sbeta, strans, ws.rjoints + (size_t)b * J * 3, jb);
    __syncthreads();
    if (lane < S) ws.beta[(size_t)b * S + lane] = sbeta[lane];
    if (lane < 3) ws.trans[(size_t)b * 3 + lane] = strans[lane];
  } else {
    for (int k = lane; k < J * 4; k += 64) jb[k] = ws.jb[(size_t)b * J * 4 + k];
    if (lane < S) {
      float v = (beta_in && lane < nb) ? beta_in[(size_t)b * nb + lane] : 0.f;
      if (kid_in && m.jt.n_kid && lane == S - 1) v = kid_in[b];
      sbeta[lane] = v;
    }
    if (lane < 3) strans[lane] = trans_in ? trans_in[(size_t)b * 3 + lane] : 0.f;
  }
  // MODE 2 walks the dense tiles (every slot), the other modes the part-aligned segments
  const int nsteps = MODE == 2 ? Vp / 64 : m.nseg;
  // cooperative staging of the constants tile: TILE_F4 float4 over 256 threads (<= 3 each), kept in
  // named registers (an indexed array here ends up in scratch)
  const float4* cg = reinterpret_cast<const float4*>(MODE == 2 ? m.cpackA : m.cpackB);
  static_assert(TILE_F4 <= 1024, "constants tile too large for 4 float4 per thread");
  float4 c0, c1, c2, c3;
  auto cload = [&](int step) {
    const float4* src = cg + (size_t)step * TILE_F4;
    c0 = src[tid];
    if (TILE_F4 > 256 && tid + 256 < TILE_F4) c1 = src[tid + 256];
    if (TILE_F4 > 512 && tid + 512 < TILE_F4) c2 = src[tid + 512];
    if (TILE_F4 > 768 && tid + 768 < TILE_F4) c3 = src[tid + 768];
  };
  auto cstore = [&](int buf) {
    float4* dst = reinterpret_cast<float4*>(cst + buf * 64 * CS);
    dst[tid] = c0;
    if (TILE_F4 > 256 && tid + 256 < TILE_F4) dst[tid + 256] = c1;
    if (TILE_F4 > 512 && tid + 512 < TILE_F4) dst[tid + 512] = c2;
    if (TILE_F4 > 768 && tid + 768 < TILE_F4) dst[tid + 768] = c3;
  };
  const float* tvs = ws.tvs + (size_t)b * 3 * Vp;
  const float* vps = ws.vposed + (size_t)b * 3 * Vp;
  const float* vws = ws.vws + (size_t)b * Vp;
  float* rv = ws.rverts + (size_t)b * 3 * Vp;
  float nx[7];
  int n_start = 0, n_count = 0, n_part = 0;
  auto sload = [&](int step) {
    if (MODE == 2) {
      n_start = step * 64;
      n_count = 64;
      n_part = 0;
    } else {
      n_start = m.segments[step * 3];
      n_count = m.segments[step * 3 + 1];
      n_part = m.segments[step * 3 + 2];
    }
    const int i = n_start + (lane < n_count ? lane : 0);
    nx[0] = vps[i]; nx[1] = vps[Vp + i]; nx[2] = vps[2 * Vp + i];
    if (MODE != 2) {
      nx[3] = tvs[i]; nx[4] = tvs[Vp + i]; nx[5] = tvs[2 * Vp + i];
      nx[6] = WEIGHTED ? vws[i] : 1.f;
    }
  };
  cload(0);
  sload(0);
  cstore(0);
  __syncthreads();
  float beta[S];
#pragma unroll
  for (int s = 0; s < S; ++s) beta[s] = sbeta[s];
  const float trans[3] = {strans[0], strans[1], strans[2]};
  float acc[sf::kPsum];
#pragma unroll
  for (int k = 0; k < sf::kPsum; ++k) acc[k] = 0.f;
  for (int step = 0; step < nsteps; ++step) {
    const float vp[3] = {nx[0], nx[1], nx[2]};
    const float tv[3] = {nx[3], nx[4], nx[5]};
    const float wv = nx[6];
    const int start = n_start, count = n_count, part = n_part;
    if (step + 1 < nsteps) {
      cload(step + 1);
      sload(step + 1);
    }
    const bool flush = MODE != 2 && ((step + 1 == nsteps) || (n_part != part));
    if (lane < count) {
      float v[3];
      sf::lbs_vertex<S, KW>(jd, jb, cst + (step & 1) * 64 * CS + lane * CS, vp, beta, trans, v);
      const int i = start + lane;
      if (MODE == 2) {
        const int o = m.perm[i];
        if (o >= 0 && live) {
          out[((size_t)b * V + o) * 3] = v[0];
          out[((size_t)b * V + o) * 3 + 1] = v[1];
          out[((size_t)b * V + o) * 3 + 2] = v[2];
        }
      } else {
        if (MODE == 1 && live) {
          rv[i] = v[0];
          rv[Vp + i] = v[1];
          rv[2 * Vp + i] = v[2];
        }
        sf::partsum_vertex(tv, v, wv, WEIGHTED, acc);
      }
    }
    if (flush) {
      float* ps = ws.psum + ((size_t)b * J + part) * sf::kPsum;
#pragma unroll
      for (int k = 0; k < sf::kPsum; ++k) {
        const float r = wave_sum_last(acc[k]);
        if (lane == 63 && live) ps[k] = r;
        acc[k] = 0.f;
      }
    }
    if (step + 1 < nsteps) cstore((step + 1) & 1);
    __syncthreads();
  }
}

// the unused-part vertices of the joints-omitted path (MODE 1 stores only the used segments)
template <int S, int KW>
__global__ __launch_bounds__(256) void k_lbs_rest(DevModel m, Workspace ws) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int STRIDE = sf::jd_stride(S);
  const int b = blockIdx.x, tid = threadIdx.x;
  const int J = m.J, Vp = m.Vp;
  float* jd = smem;
  float* jb = jd + J * STRIDE;
  for (int k = tid; k < J * STRIDE; k += 256) jd[k] = ws.jd[(size_t)b * J * STRIDE + k];
  for (int k = tid; k < J * 4; k += 256) jb[k] = ws.jb[(size_t)b * J * 4 + k];
  __syncthreads();
  float beta[S];
#pragma unroll
  for (int s = 0; s < S; ++s) beta[s] = ws.beta[(size_t)b * S + s];
  const float trans[3] = {ws.trans[b * 3], ws.trans[b * 3 + 1], ws.trans[b * 3 + 2]};
  const float* vps = ws.vposed + (size_t)b * 3 * Vp;
  float* rv = ws.rverts + (size_t)b * 3 * Vp;
  for (int i = m.n_used + tid; i < Vp; i += 256) {
    float v[3] = {0.f, 0.f, 0.f};
    if (m.perm[i] >= 0) {
      const float vp[3] = {vps[i], vps[Vp + i], vps[2 * Vp + i]};
      sf::lbs_vertex<S, KW>(jd, jb, m.cpackA + (size_t)i * sf::cpack_stride(S, KW), vp, beta, trans, v);
    }
    rv[i] = v[0];
    rv[Vp + i] = v[1];
    rv[2 * Vp + i] = v[2];
  }
}

// ------------------------------------------------------------------------------------------------
// K6: dependent refinement + epilogue.  grid B, block 64.
// ------------------------------------------------------------------------------------------------
struct RefineArgs {
  const float* tj;        // (B,J,3) joints of the joint term (centred targets or regressed)
  const float* rj_term;   // (B,J,3) reference joints of the joint term
  const float* jw;        // (B,J) or null
  int final_adjust;
  float *pose, *betas, *trans, *kid, *orient, *rel;
  int scaled;  // known-shape fit with scale_fit: rest joints are scaled by ws.scale
};

__global__ __launch_bounds__(64) void k_refine_epilogue(DevModel m, RefineArgs a, Workspace ws) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.x, J = m.J, S = m.S;
  DevCtx cx{(int)threadIdx.x, 64};
  sf::JointScratch sh = sf::carve_joint_scratch(smem, J, S, 1);
  sf::refine_stage(cx, m.jt, sh, ws.psum + (size_t)b * J * sf::kPsum, a.tj + (size_t)b * J * 3,
                   a.rj_term + (size_t)b * J * 3, ws.rjoints + (size_t)b * J * 3,
                   a.jw ? a.jw + (size_t)b * J : nullptr, ws.G + (size_t)b * J * 9,
                   ws.beta + (size_t)b * S, ws.trans + (size_t)b * 3, ws.mean + (size_t)b * 3,
                   a.final_adjust != 0, a.pose + (size_t)b * J * 3,
                   a.betas ? a.betas + (size_t)b * (S - m.jt.n_kid) : nullptr, a.trans + (size_t)b * 3,
                   a.kid ? a.kid + b : nullptr, a.orient ? a.orient + (size_t)b * J * 9 : nullptr,
                   a.rel ? a.rel + (size_t)b * J * 9 : nullptr, a.scaled ? ws.scale + b : nullptr);
}

// ------------------------------------------------------------------------------------------------
// known-shape fit: the given shape (betas [+ kid]) to the workspace layout, zero translation
// ------------------------------------------------------------------------------------------------
__global__ void k_fill_shape(Workspace ws, int B, int S, int n_kid, const float* __restrict__ betas,
                             int nb, const float* __restrict__ kid) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  for (int s = 0; s < S - n_kid; ++s)
    ws.beta[(size_t)b * S + s] = (betas && s < nb) ? betas[(size_t)b * nb + s] : 0.f;
  if (n_kid) ws.beta[(size_t)b * S + S - 1] = kid ? kid[b] : 0.f;
  for (int s = 0; s < S; ++s) ws.regref[(size_t)b * S + s] = ws.beta[(size_t)b * S + s];
  for (int c = 0; c < 3; ++c) ws.trans[b * 3 + c] = 0.f;
}

// known-shape fit: scale + translation of the posed reference onto the target, folded into the part
// sums and joints the refinement reads (sf::scale_trans_stage).  grid B, block 256.
struct ScaleTransArgs {
  const float* tj;   // (B,J,3) centred target joints or null (vertices only)
  const float* jw;   // (B,J) or null
  int weighted_v, with_scale, regressed;
  float* scale_out;  // (B) or null
};

__global__ __launch_bounds__(256) void k_scale_trans(DevModel m, ScaleTransArgs a, Workspace ws) {
  __shared__ float red[256 * 8];
  const int b = blockIdx.x, J = m.J, Vp = m.Vp;
  DevCtx cx{(int)threadIdx.x, 256};
  sf::scale_trans_stage(cx, J, m.V, Vp, red, ws.tvs + (size_t)b * 3 * Vp, ws.rverts + (size_t)b * 3 * Vp,
                        a.weighted_v ? ws.vws + (size_t)b * Vp : nullptr,
                        a.tj ? a.tj + (size_t)b * J * 3 : nullptr, ws.rjoints + (size_t)b * J * 3,
                        a.jw ? a.jw + (size_t)b * J : nullptr, a.with_scale != 0,
                        ws.psum + (size_t)b * J * sf::kPsum,
                        a.regressed ? ws.rjreg + (size_t)b * J * 3 : nullptr, m.reg_rowsum,
                        ws.trans + (size_t)b * 3, ws.scale + b);
  if (a.scale_out && threadIdx.x == 0) a.scale_out[b] = ws.scale[b];  // written by this lane above
}

// ------------------------------------------------------------------------------------------------
// forward: joint prologue.  grid B, block 64.
// ------------------------------------------------------------------------------------------------
struct ForwardArgs {
  const float *pose, *glob, *betas, *trans, *kid;
  int nb;
  float *joints, *orient;
};

__global__ __launch_bounds__(64) void k_forward_joint(DevModel m, ForwardArgs a, Workspace ws) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.x, J = m.J, S = m.S;
  DevCtx cx{(int)threadIdx.x, 64};
  sf::JointScratch sh = sf::carve_joint_scratch(smem, J, S, 0);
  float* jd = ws.jd + (size_t)b * J * sf::jd_stride(S);
  sf::forward_joint_stage(cx, m.jt, sh, a.pose ? a.pose + (size_t)b * J * 3 : nullptr,
                          a.glob ? a.glob + (size_t)b * J * 9 : nullptr,
                          a.betas ? a.betas + (size_t)b * a.nb : nullptr, a.betas ? a.nb : 0,
                          a.kid ? a.kid + b : nullptr, a.trans ? a.trans + (size_t)b * 3 : nullptr,
                          ws.rp + (size_t)b * m.Kp, jd,
                          a.joints + (size_t)b * J * 3,
                          a.orient ? a.orient + (size_t)b * J * 9 : nullptr);
  __syncthreads();
  // skinning translations for the vertex kernel
  for (int k = threadIdx.x; k < J * 3; k += 64)
    ws.jb[(size_t)b * J * 4 + (k / 3) * 4 + k % 3] = jd[(k / 3) * sf::jd_stride(S) + 9 + k % 3];
}

// betas / kid / trans of the last solve to the caller's arrays (shape-solve entry point)
__global__ void k_emit_solution(Workspace ws, int B, int S, int n_kid, int add_mean,
                                float* __restrict__ betas, float* __restrict__ trans,
                                float* __restrict__ kid) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  for (int s = 0; s < S - n_kid; ++s) betas[(size_t)b * (S - n_kid) + s] = ws.beta[(size_t)b * S + s];
  if (n_kid && kid) kid[b] = ws.beta[(size_t)b * S + S - 1];
  for (int c = 0; c < 3; ++c)
    trans[b * 3 + c] = ws.trans[b * 3 + c] + (add_mean ? ws.mean[b * 3 + c] : 0.f);
}

// scatter G given by the caller into the workspace (shape-solve entry point)
__global__ void k_copy(const float* __restrict__ src, float* __restrict__ dst, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    dst[i] = src[i];
}

// ================================================================================================
// Batch-major vertex kernels: LANE = INSTANCE.  A workgroup owns 64 consecutive instances and one
// vertex group (HostTables::groups: a run of one part's sorted slots with <= 12 skinning joints);
// its 4 waves split the group's vertices.  Everything that depends on the vertex only (shapedirs,
// skinning weights, joint slots) is wave-uniform and arrives through scalar loads; everything that
// depends on the instance (rotations, translations, betas, accumulators) lives in the lane's
// registers or in a lane-private LDS column; the streams (v_posed, targets) are stored instance-
// innermost so that a wave reads 256 contiguous bytes per vertex coordinate.  No cross-lane
// reductions, no per-vertex constants in VGPRs / LDS.
// ================================================================================================
constexpr int kGQ = sf::kGroupJoints;

// Batch-major stream layout: [instance block of 64][n][64 instances], n = c * Vp + slot.  A workgroup
// (one instance block, one vertex group) then streams three contiguous runs per buffer.
// targets: sorted SoA per instance [B][N] -> batch-major (N = 3 Vp; 64x64 LDS tiles)
__global__ __launch_bounds__(256) void k_transpose_targets(const float* __restrict__ src,
                                                           float* __restrict__ dst, int B, int N,
                                                           int Mp) {
  __shared__ float tile[64][65];
  const int n0 = blockIdx.x * 64, b0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const int b = b0 + r;
    tile[r][tx] = b < B ? src[(size_t)b * N + n0 + tx] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) dst[((size_t)blockIdx.y * N + n0 + r) * 64 + tx] = tile[tx][r];
}

typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 mk2(float a, float b) { f2 r; r.x = a; r.y = b; return r; }

__host__ __device__ constexpr int brec_w(int S) { return (3 * S + 3) / 4 * 4; }
__host__ __device__ constexpr int brec_d(int S, int KW) { return brec_w(S) + (KW + KW / 4 + 3) / 4 * 4; }
__host__ __device__ constexpr int brec_stride(int S, int KW) { return brec_d(S, KW) + kGQ; }

// K5 (batch-major): vertices at the solved shape + part sums against the targets, used groups only.
// grid (ngroups_used, Mp/64), block 256.  LDS: [kGQ][12][64] joint data as 6 pairs per joint
//   (R0,R3) (R1,R4) (R2,R5) (R6,R7) (R8, jb2+trans2) (jb0+trans0, jb1+trans1)
// so that the blended quantities come out in the register pairs the packed-fp32 rotation wants;
// reused as [kBW][16][64] for the wave combine.  Output: ws.psumP[g][16][Mp].
// Per vertex: 24 ds_read2st64 + ~60 VALU (packed fp32); the vertex record is read with scalar loads
// one vertex ahead (weights / slots) resp. right after its last use (shapedirs).
// waves per workgroup of the batch-major kernels (they share the staged joints).  4, not 8: one such
// workgroup (1 wave per SIMD, ~120 VGPRs, 37 KB LDS) fits on a CU next to two GEMM workgroups of
// another chunk, so the MFMA-bound GEMM and these VALU / LDS / HBM-bound passes really overlap
constexpr int kBW = 4;

template <int S, int KW>
__global__ __launch_bounds__(64 * kBW) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_lbs_partsum_bm(DevModel m, Workspace ws, int B, int Mp) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int STRIDE = sf::jd_stride(S), BW = brec_w(S), BS = brec_stride(S, KW);
  static_assert(KW == 4, "batch-major kernels: 4 skinning pairs per vertex");
  const int tid = threadIdx.x, lane = tid & 63;
  // wave index as a SCALAR: everything derived from it (vertex range, record addresses) stays
  // wave-uniform for the compiler, so the per-vertex records are fetched with s_load
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = blockIdx.x, J = m.J, Vp = m.Vp;
  const int32_t* gr = m.groups + (size_t)g * kGroupRec;
  const int start = gr[0], count = gr[1], nq = gr[4];
  const int bcol = blockIdx.y * 64 + lane;           // column of the instance-innermost buffers
  const int b = bcol < B ? bcol : B - 1;             // row of the per-instance buffers
  const float tr0 = ws.trans[b * 3], tr1 = ws.trans[b * 3 + 1], tr2 = ws.trans[b * 3 + 2];
  for (int q = wave; q < nq; q += kBW) {             // stage the group's joints: lane = instance
    const int j = gr[5 + q];
    const float* row = ws.jd + ((size_t)b * J + j) * STRIDE;
    const float4 r0 = *reinterpret_cast<const float4*>(row);
    const float4 r1 = *reinterpret_cast<const float4*>(row + 4);
    const float r8 = row[8];
    const float4 tb = *reinterpret_cast<const float4*>(ws.jb + ((size_t)b * J + j) * 4);
    float* dst = smem + (size_t)q * 12 * 64 + lane;
    dst[0] = r0.x; dst[64] = r0.w;        // R0 R3
    dst[128] = r0.y; dst[192] = r1.x;     // R1 R4
    dst[256] = r0.z; dst[320] = r1.y;     // R2 R5
    dst[384] = r1.z; dst[448] = r1.w;     // R6 R7
    dst[512] = r8; dst[576] = tb.z + tr2; // R8 jb2
    dst[640] = tb.x + tr0; dst[704] = tb.y + tr1;
  }
  float beta[S];
#pragma unroll
  for (int s = 0; s < S; ++s) beta[s] = ws.beta[(size_t)b * S + s];
  __syncthreads();
  const int per = (count + kBW - 1) / kBW;
  const int v0 = start + wave * per, v1 = min(v0 + per, start + count);
  // accumulators: P0..P2 = raw rows (cols 0,1), P3 = (raw02, raw12), s22; P4 = (st0, st1), st2;
  // P5 = (sa0, sa1), sa2
  f2 P0 = mk2(0, 0), P1 = P0, P2 = P0, P3 = P0, P4 = P0, P5 = P0;
  float s22 = 0.f, st2 = 0.f, sa2 = 0.f;
  const float* vp = ws.vpT + (size_t)blockIdx.y * 3 * Vp * 64 + lane;
  const float* tp = ws.tT + (size_t)blockIdx.y * 3 * Vp * 64 + lane;
  const size_t cstr = (size_t)Vp * 64;  // coordinate stride inside the instance block
  // Software pipeline, 3 stages per iteration v:  the streams of vertex v+2 are requested (vector
  // loads, consumed one iteration later: a full iteration of latency cover per wave, x 6 waves/SIMD),
  // vertex v+1 is prepared from its streams + its record (scalar loads) down to 13 carried registers
  // (shaped rest position, target, LDS offsets of its 4 joints, weights), vertex v is blended from
  // LDS, posed and accumulated.
  struct Raw { float x0, x1, x2, t0, t1, t2; };
  auto fetch = [&](int v) {
    const size_t o = (size_t)v * 64;
    Raw r;
    r.x0 = vp[o]; r.x1 = vp[o + cstr]; r.x2 = vp[o + 2 * cstr];
    r.t0 = tp[o]; r.t1 = tp[o + cstr]; r.t2 = tp[o + 2 * cstr];
    return r;
  };
  auto prep = [&](int v, const Raw& r, f2& vs01, float& vs2, f2& t01, float& t2, int (&off)[4],
                  float (&wq)[4]) {
    t01 = mk2(r.t0, r.t1);
    t2 = r.t2;
    const float* rec = m.brec + (size_t)v * BS;  // wave-uniform -> scalar loads
    f2 vz = mk2(r.x2, 0.f);
    f2 vx = mk2(r.x0, 0.f), vy = mk2(r.x1, 0.f);
#pragma unroll
    for (int s2 = 0; s2 + 1 < S; s2 += 2) {  // beta pairs against the per-coordinate shapedirs rows
      const f2 bp = mk2(beta[s2], beta[s2 + 1]);
      vx += mk2(rec[s2], rec[s2 + 1]) * bp;
      vy += mk2(rec[S + s2], rec[S + s2 + 1]) * bp;
      vz += mk2(rec[2 * S + s2], rec[2 * S + s2 + 1]) * bp;
    }
    if (S & 1) {
      vx.x += rec[S - 1] * beta[S - 1];
      vy.x += rec[2 * S - 1] * beta[S - 1];
      vz.x += rec[3 * S - 1] * beta[S - 1];
    }
    vs01 = mk2(vx.x + vx.y, vy.x + vy.y);
    vs2 = vz.x + vz.y;
    const uint32_t slots = __float_as_uint(rec[BW + 4]);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      off[k] = lane + (int)((slots >> (8 * k)) & 0xffu) * (12 * 64);
      wq[k] = rec[BW + k];
    }
  };
  if (v0 < v1) {
    f2 vs01, t01;
    float vs2, t2, wq[4];
    int off[4];
    const int vl = v1 - 1;  // prefetches are clamped to the last vertex (harmless re-reads)
    prep(v0, fetch(v0), vs01, vs2, t01, t2, off, wq);
    // two raw-stream buffers used alternately (no register copies: a copy would make the compiler
    // wait for the loads at the end of the very iteration that issued them)
    Raw rA, rB = fetch(min(v0 + 1, vl));
    auto step = [&](int v, const Raw& use, Raw& fill) {
      fill = fetch(min(v + 2, vl));
      // this vertex's prepared values, then the preparation of the next one FIRST: its scalar loads
      // are in flight while the LDS reads of the blend below are issued
      const f2 cs01 = vs01, ct01 = t01;
      const float cs2 = vs2, ct2 = t2;
      const int o0 = off[0], o1 = off[1], o2 = off[2], o3 = off[3];
      const float w0 = wq[0], w1 = wq[1], w2 = wq[2], w3 = wq[3];
      prep(min(v + 1, vl), use, vs01, vs2, t01, t2, off, wq);
      // blend of the 4 joints: 6 register pairs
      f2 Q0 = mk2(0, 0), Q1 = Q0, Q2 = Q0, Q3 = Q0, Q4 = Q0, Q5 = Q0;
      const int co[4] = {o0, o1, o2, o3};
      const float cw[4] = {w0, w1, w2, w3};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float* src = smem + co[k];
        const float w = cw[k];
        Q0 += w * mk2(src[0], src[64]);
        Q1 += w * mk2(src[128], src[192]);
        Q2 += w * mk2(src[256], src[320]);
        Q3 += w * mk2(src[384], src[448]);
        Q4 += w * mk2(src[512], src[576]);
        Q5 += w * mk2(src[640], src[704]);
      }
      // posed vertex (translation already folded into the staged jb)
      const f2 a01 = (Q0 * cs01.x + Q1 * cs01.y) + (Q2 * cs2 + Q5);
      const float a2 = (Q3.x * cs01.x + Q3.y * cs01.y) + (Q4.x * cs2 + Q4.y);
      // part sums (_part_sums, bodyfitter.py:257-280), unit weights
      P0 += ct01.x * a01;
      P1 += ct01.y * a01;
      P2 += ct2 * a01;
      P3 += ct01 * a2;
      s22 += ct2 * a2;
      P4 += ct01;
      st2 += ct2;
      P5 += a01;
      sa2 += a2;
    };
    int v = v0;
    for (; v + 1 < v1; v += 2) {
      step(v, rB, rA);
      step(v + 1, rA, rB);
    }
    if (v < v1) step(v, rB, rA);
  }
  const float acc[sf::kPsum] = {P0.x, P0.y, P3.x, P1.x, P1.y, P3.y, P2.x, P2.y, s22,
                                P4.x, P4.y, st2, P5.x, P5.y, sa2, (float)max(v1 - v0, 0)};
  // combine the waves in a fixed order (deterministic), wave 0 writes
  __syncthreads();
#pragma unroll
  for (int k = 0; k < sf::kPsum; ++k) smem[(wave * sf::kPsum + k) * 64 + lane] = acc[k];
  __syncthreads();
  if (wave == 0) {
    float* out = ws.psumP + (size_t)g * sf::kPsum * Mp + bcol;
#pragma unroll
    for (int k = 0; k < sf::kPsum; ++k) {
      float sum = smem[k * 64 + lane];
#pragma unroll
      for (int w = 1; w < kBW; ++w) sum += smem[(w * sf::kPsum + k) * 64 + lane];
      out[(size_t)k * Mp] = sum;
    }
  }
}

// K3 (batch-major, unit weights): residual pass of the pair-Gram form over ALL vertex groups.
//   b_v = t_v - (Rt_v v_posed_v + T0_v);  Sb = sum b;  r1 = sum_v S_v^T (Rt_v^T b_v);
//   mb_q = sum_v w_vq b_v for the group's joint slots q (dense weights of the record).
// grid (ngroups, Mp/64), block 64*kBW.  LDS joint data as in k_lbs_partsum_bm with T0 in place of jb.
// Output: ws.resP[g][kResRec][Mp] = [r1 : S][Sb : 3][mb : 12 x 3].
constexpr int kResRec = 16 + 3 * kGQ;  // S <= 13 + 3 here (S = 10 / 11); padded record

template <int S>
__global__ __launch_bounds__(64 * kBW) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_residual_bm(
    DevModel m, Workspace ws, int B, int Mp) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int KW = 4, STRIDE = sf::jd_stride(S), BW = brec_w(S), BD = brec_d(S, KW),
                BS = brec_stride(S, KW);
  static_assert(S % 2 == 0 && S + 3 <= 16, "batch-major residual kernel: even S <= 12");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = blockIdx.x, J = m.J, Vp = m.Vp;
  const int32_t* gr = m.groups + (size_t)g * kGroupRec;
  const int start = gr[0], count = gr[1], nq = gr[4];
  const int bcol = blockIdx.y * 64 + lane;
  const int b = bcol < B ? bcol : B - 1;
  for (int q = wave; q < nq; q += kBW) {
    const int j = gr[5 + q];
    const float* row = ws.jd + ((size_t)b * J + j) * STRIDE;
    const float4 r0 = *reinterpret_cast<const float4*>(row);
    const float4 r1 = *reinterpret_cast<const float4*>(row + 4);
    const float4 r2 = *reinterpret_cast<const float4*>(row + 8);  // R8, T0
    float* dst = smem + (size_t)q * 12 * 64 + lane;
    dst[0] = r0.x; dst[64] = r0.w;        // R0 R3
    dst[128] = r0.y; dst[192] = r1.x;     // R1 R4
    dst[256] = r0.z; dst[320] = r1.y;     // R2 R5
    dst[384] = r1.z; dst[448] = r1.w;     // R6 R7
    dst[512] = r2.x; dst[576] = r2.w;     // R8 T0z
    dst[640] = r2.y; dst[704] = r2.z;     // T0x T0y
  }
  __syncthreads();
  const int per = (count + kBW - 1) / kBW;
  const int v0 = start + wave * per, v1 = min(v0 + per, start + count);
  f2 r1p[S / 2], sb01 = mk2(0, 0), m01[kGQ];
  float sb2 = 0.f, m2[kGQ];
#pragma unroll
  for (int k = 0; k < S / 2; ++k) r1p[k] = mk2(0, 0);
#pragma unroll
  for (int q = 0; q < kGQ; ++q) {
    m01[q] = mk2(0, 0);
    m2[q] = 0.f;
  }
  const float* vp = ws.vpT + (size_t)blockIdx.y * 3 * Vp * 64 + lane;
  const float* tp = ws.tT + (size_t)blockIdx.y * 3 * Vp * 64 + lane;
  const size_t cstr = (size_t)Vp * 64;
  struct Raw { float x0, x1, x2, t0, t1, t2; };
  auto fetch = [&](int v) {
    const size_t o = (size_t)v * 64;
    Raw r;
    r.x0 = vp[o]; r.x1 = vp[o + cstr]; r.x2 = vp[o + 2 * cstr];
    r.t0 = tp[o]; r.t1 = tp[o + cstr]; r.t2 = tp[o + 2 * cstr];
    return r;
  };
  auto slots_of = [&](int v, int (&off)[4], float (&wq)[4]) {
    const float* rec = m.brec + (size_t)v * BS;  // wave-uniform -> scalar loads
    const uint32_t slots = __float_as_uint(rec[BW + 4]);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      off[k] = lane + (int)((slots >> (8 * k)) & 0xffu) * (12 * 64);
      wq[k] = rec[BW + k];
    }
  };
  if (v0 < v1) {
    const int vl = v1 - 1;
    int off[4];
    float wq[4];
    slots_of(v0, off, wq);
    Raw rA = fetch(v0), rB = fetch(min(v0 + 1, vl));
    auto step = [&](int v, const Raw& cur, Raw& fill) {
      // joint slots of the NEXT vertex first: the scalar loads fly while the LDS reads below issue
      const int co[4] = {off[0], off[1], off[2], off[3]};
      const float cw[4] = {wq[0], wq[1], wq[2], wq[3]};
      slots_of(min(v + 1, vl), off, wq);
      // blend of the 4 joints: (R0,R3) (R1,R4) (R2,R5) (R6,R7) (R8,T0z) (T0x,T0y)
      f2 Q0 = mk2(0, 0), Q1 = Q0, Q2 = Q0, Q3 = Q0, Q4 = Q0, Q5 = Q0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float* src = smem + co[k];
        const float w = cw[k];
        Q0 += w * mk2(src[0], src[64]);
        Q1 += w * mk2(src[128], src[192]);
        Q2 += w * mk2(src[256], src[320]);
        Q3 += w * mk2(src[384], src[448]);
        Q4 += w * mk2(src[512], src[576]);
        Q5 += w * mk2(src[640], src[704]);
      }
      const float* rec = m.brec + (size_t)v * BS;  // shapedirs + dense weights of THIS vertex
      // residual
      const f2 pos01 = (Q0 * cur.x0 + Q1 * cur.x1) + (Q2 * cur.x2 + Q5);
      const float pos2 = (Q3.x * cur.x0 + Q3.y * cur.x1) + (Q4.x * cur.x2 + Q4.y);
      const f2 b01 = mk2(cur.t0, cur.t1) - pos01;
      const float b2 = cur.t2 - pos2;
      fill = fetch(min(v + 2, vl));
      sb01 += b01;
      sb2 += b2;
      // u = Rt^T b
      const float u0 = (Q0.x * b01.x + Q0.y * b01.y) + Q3.x * b2;
      const float u1 = (Q1.x * b01.x + Q1.y * b01.y) + Q3.y * b2;
      const float u2 = (Q2.x * b01.x + Q2.y * b01.y) + Q4.x * b2;
#pragma unroll
      for (int k = 0; k < S / 2; ++k)
        r1p[k] += (mk2(rec[2 * k], rec[2 * k + 1]) * u0 + mk2(rec[S + 2 * k], rec[S + 2 * k + 1]) * u1) +
                  mk2(rec[2 * S + 2 * k], rec[2 * S + 2 * k + 1]) * u2;
#pragma unroll
      for (int q = 0; q < kGQ; ++q) {
        const float wd = rec[BD + q];
        m01[q] += wd * b01;
        m2[q] += wd * b2;
      }
    };
    int v = v0;
    for (; v + 1 < v1; v += 2) {
      step(v, rA, rA);      // rA is consumed before it is refilled with vertex v + 2
      step(v + 1, rB, rB);
    }
    if (v < v1) step(v, rA, rA);
  }
  // combine the waves in a fixed order, 16 values per pass through the (reused) staging region
  float vals[kResRec];
#pragma unroll
  for (int k = 0; k < S / 2; ++k) {
    vals[2 * k] = r1p[k].x;
    vals[2 * k + 1] = r1p[k].y;
  }
  vals[S] = sb01.x; vals[S + 1] = sb01.y; vals[S + 2] = sb2;
#pragma unroll
  for (int k = S + 3; k < 16; ++k) vals[k] = 0.f;
#pragma unroll
  for (int q = 0; q < kGQ; ++q) {
    vals[16 + 3 * q] = m01[q].x;
    vals[16 + 3 * q + 1] = m01[q].y;
    vals[16 + 3 * q + 2] = m2[q];
  }
  float* out = ws.resP + (size_t)g * kResRec * Mp + bcol;
#pragma unroll
  for (int pass = 0; pass < (kResRec + 15) / 16; ++pass) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k)
      if (pass * 16 + k < kResRec) smem[(wave * 16 + k) * 64 + lane] = vals[pass * 16 + k];
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        if (pass * 16 + k >= kResRec) break;
        float sum = smem[k * 64 + lane];
#pragma unroll
        for (int w = 1; w < kBW; ++w) sum += smem[(w * 16 + k) * 64 + lane];
        out[(size_t)(pass * 16 + k) * Mp] = sum;
      }
    }
  }
}

// K3g (batch-major): pair-Gram of sf::pair_gram_stage with lane = instance.  Work units = the np joint
// pairs followed by the J diagonal joints; a wave takes a contiguous run of units for its 64
// instances and accumulates the NG upper-triangle entries in registers; all model constants
// (c1: 9 S^2 per pair) arrive through scalar loads.  grid (kGramChunks, Mp/64), block 64.
// Output: ws.gramP[chunk][NG][Mp].
constexpr int kGramChunks = 32;

template <int S>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_pair_gram_bm(DevModel m, Workspace ws, int B, int Mp) {
  constexpr int STRIDE = sf::jd_stride(S), ROW = sf::jd_row(S), NG = sf::ne_ng(S), NC1 = 9 * S * S;
  constexpr int NLD = (NC1 + 63) / 64;
  // the pair's 9 S^2 constants go through LDS (coalesced load, broadcast reads): as scalar loads the
  // compiler hoists all of them and spills SGPRs by the thousand
  __shared__ __attribute__((aligned(16))) float c1s[2][NLD * 64];
  const int lane = threadIdx.x, J = m.J, np = m.jt.np;
  const int bcol = blockIdx.y * 64 + lane;
  const int b = bcol < B ? bcol : B - 1;
  const int nunits = np + J;
  const int per = (nunits + kGramChunks - 1) / kGramChunks;
  const int u0 = blockIdx.x * per, u1 = min(u0 + per, nunits);
  float G[NG];
#pragma unroll
  for (int e = 0; e < NG; ++e) G[e] = 0.f;
  const float* jdb = ws.jd + (size_t)b * J * STRIDE;
  auto load_RD = [&](int j, float (&R)[9], float (&D)[3 * S]) {  // R_j and D_j = R_j^T T'_j
    const float* row = jdb + (size_t)j * STRIDE;
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = row[k];
    float T[3 * S];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int i = 0; i < S; ++i) T[c * S + i] = row[12 + c * ROW + i];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int i = 0; i < S; ++i)
        D[a * S + i] = (R[a] * T[i] + R[3 + a] * T[S + i]) + R[6 + a] * T[2 * S + i];
  };
  float nx[NLD];  // the next pair's constants on their way to LDS
  auto c1_fetch = [&](int u) {
    const float* src = m.pair_c1x + (size_t)(u < np ? u : 0) * NC1;
#pragma unroll
    for (int k = 0; k < NLD; ++k) nx[k] = (k * 64 + lane < NC1) ? src[k * 64 + lane] : 0.f;
  };
  auto c1_commit = [&](int buf) {
#pragma unroll
    for (int k = 0; k < NLD; ++k) c1s[buf][k * 64 + lane] = nx[k];
  };
  if (u0 < u1) {
    c1_fetch(u0);
    c1_commit(0);
  }
  for (int u = u0; u < u1; ++u) {
    const int buf = (u - u0) & 1;
    if (u + 1 < u1) c1_fetch(u + 1);
    if (u < np) {
      const int j1 = m.jt.pair_j[2 * u], j2 = m.jt.pair_j[2 * u + 1];
      float R1[9], R2[9], D1[3 * S], D2[3 * S];
      load_RD(j1, R1, D1);
      load_RD(j2, R2, D2);
      float Q[9];
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int a2 = 0; a2 < 3; ++a2)
          Q[a * 3 + a2] = (R1[a] * R2[a2] + R1[3 + a] * R2[3 + a2]) + R1[6 + a] * R2[6 + a2];
      const float* c1 = c1s[buf];                          // [x][a][a'][y], uniform addresses
      const float* c2 = m.jt.pair_c2 + (size_t)u * 3 * S;  // wave-uniform -> scalar loads
      const float c3 = m.jt.pair_c3[u];
      // U = Q D2, V = Q c2 + c3 U as register pairs over y (packed fp32)
      f2 U[3][S / 2], V[3][S / 2];
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int k = 0; k < S / 2; ++k) {
          const f2 uu = (Q[a * 3] * mk2(D2[2 * k], D2[2 * k + 1]) + Q[a * 3 + 1] * mk2(D2[S + 2 * k], D2[S + 2 * k + 1])) +
                        Q[a * 3 + 2] * mk2(D2[2 * S + 2 * k], D2[2 * S + 2 * k + 1]);
          U[a][k] = uu;
          V[a][k] = ((Q[a * 3] * mk2(c2[2 * k], c2[2 * k + 1]) + Q[a * 3 + 1] * mk2(c2[S + 2 * k], c2[S + 2 * k + 1])) +
                     Q[a * 3 + 2] * mk2(c2[2 * S + 2 * k], c2[2 * S + 2 * k + 1])) + c3 * uu;
        }
#pragma unroll
      for (int x = 0; x < S; ++x) {
        f2 f[S / 2];
#pragma unroll
        for (int k = 0; k < S / 2; ++k) f[k] = mk2(0.f, 0.f);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          const float* cx = c1 + (x * 3 + a) * 3 * S;  // 3 x S constants of this (x, a)
          const float c2x = c2[a * S + x], d1x = D1[a * S + x];
#pragma unroll
          for (int k = 0; k < S / 2; ++k)
            f[k] += ((Q[a * 3] * mk2(cx[2 * k], cx[2 * k + 1]) + Q[a * 3 + 1] * mk2(cx[S + 2 * k], cx[S + 2 * k + 1])) +
                     Q[a * 3 + 2] * mk2(cx[2 * S + 2 * k], cx[2 * S + 2 * k + 1])) +
                    (c2x * U[a][k] + d1x * V[a][k]);
        }
        // G[i][i2] (i <= i2) collects f[i][i2] + f[i2][i]
#pragma unroll
        for (int y = 0; y < S; ++y) {
          const float fy = (y & 1) ? f[y / 2].y : f[y / 2].x;
          G[sf::ne_g(S, x < y ? x : y, x < y ? y : x)] += (x == y) ? 2.f * fy : fy;
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the LDS reads of one x together (register pressure)
      }
    } else {
      const int j = u - np;
      float R[9], D[3 * S];
      load_RD(j, R, D);
      const float* c2 = m.jt.diag_c2 + (size_t)j * 3 * S;
      const float c3 = m.jt.diag_c3[j];
#pragma unroll
      for (int i = 0; i < S; ++i)
#pragma unroll
        for (int i2 = i; i2 < S; ++i2) {
          float acc = 0.f;
#pragma unroll
          for (int a = 0; a < 3; ++a)
            acc += c2[a * S + i] * D[a * S + i2] + D[a * S + i] * (c2[a * S + i2] + c3 * D[a * S + i2]);
          G[sf::ne_g(S, i, i2)] += acc;
        }
    }
    if (u + 1 < u1) c1_commit(buf ^ 1);
  }
  float* out = ws.gramP + (size_t)blockIdx.x * NG * Mp + bcol;
#pragma unroll
  for (int e = 0; e < NG; ++e) out[(size_t)e * Mp] = G[e];
}

// K3c: partial sums of the two kernels above -> the normal-equation record the solve stage reads
// (gramv: G | r1 | SA = 0 (closed form in the joint stage) | Sb | W = V) and the per-joint residual
// moments ws.mbj.  grid (ceil(B/256), S + 3 + 3 J + NG): blockIdx.y = output element.
template <int S>
__global__ __launch_bounds__(256) void k_gram_combine_bm(DevModel m, Workspace ws, int B, int Mp) {
  constexpr int NG = sf::ne_ng(S), NE = sf::ne_size(S);
  const int b = blockIdx.x * 256 + threadIdx.x, e = blockIdx.y, J = m.J;
  if (b >= B) return;
  double* out = ws.gramv + (size_t)b * (NE + 1);
  if (e < S + 3) {  // r1 / Sb: groups in table order
    float acc = 0.f;
    for (int g = 0; g < m.ngroups; ++g) acc += ws.resP[((size_t)g * kResRec + e) * Mp + b];
    out[e < S ? NG + e : NG + 4 * S + (e - S)] = (double)acc;
    if (e == 0) {
      for (int k = 0; k < 3 * S; ++k) out[NG + S + k] = 0.0;
      out[NE] = (double)m.V;  // w_sum = num_vertices (bodyfitter.py:1038-1040)
    }
  } else if (e < S + 3 + 3 * J) {  // residual moment of joint j, coordinate c
    const int j = (e - S - 3) / 3, c = (e - S - 3) % 3;
    float acc = 0.f;
    for (int k = m.mb_start[j]; k < m.mb_start[j + 1]; ++k)  // groups in table order
      acc += ws.resP[((size_t)m.mb_row[k] + c) * Mp + b];
    ws.mbj[((size_t)b * J + j) * 3 + c] = acc;
  } else {  // Gramian entry: instance-independent part + the chunks of the pair kernel
    const int k = e - (S + 3 + 3 * J);
    int i = 0, r = k;
    while (r >= S - i) {
      r -= S - i;
      ++i;
    }
    float acc = m.jt.diag_g0[i * S + i + r];
    for (int ch = 0; ch < kGramChunks; ++ch) acc += ws.gramP[((size_t)ch * NG + k) * Mp + b];
    out[k] = (double)acc;
  }
}

// part sums of the vertex groups -> ws.psum[b][part][16] (groups of one part summed in table order).
// grid (ceil(B/256), J): blockIdx.y = part.
__global__ __launch_bounds__(256) void k_psum_combine(DevModel m, Workspace ws, int B, int Mp) {
  const int b = blockIdx.x * 256 + threadIdx.x, part = blockIdx.y;
  if (b >= B) return;
  float acc[sf::kPsum];
#pragma unroll
  for (int k = 0; k < sf::kPsum; ++k) acc[k] = 0.f;
  bool any = false;
  for (int g = 0; g < m.ngroups_used; ++g) {
    if (m.groups[(size_t)g * kGroupRec + 2] != part) continue;
    any = true;
#pragma unroll
    for (int k = 0; k < sf::kPsum; ++k) acc[k] += ws.psumP[((size_t)g * sf::kPsum + k) * Mp + b];
  }
  if (!any) return;
  float4* dst = reinterpret_cast<float4*>(ws.psum + ((size_t)b * m.J + part) * sf::kPsum);
  dst[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
  dst[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
  dst[2] = make_float4(acc[8], acc[9], acc[10], acc[11]);
  dst[3] = make_float4(acc[12], acc[13], acc[14], acc[15]);
}

// ------------------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------------------
// The unit-weight vertex block has two implementations:
//   direct (default): k_shape_accum accumulates G, r, Sb per vertex (VALU-bound);
//   pair  (SMPLFIT_SHAPE_FORM=pair): k_residual + k_pair_gram — 4x fewer per-vertex FLOPs, parity-tested,
//          but not yet faster: the residual pass is bound by the memory system.
bool use_pair_form() {
  static const bool pair = [] {
    const char* e = getenv("SMPLFIT_SHAPE_FORM");
    return e && std::string(e) == "pair";
  }();
  return pair;
}

// Batch-major vertex kernels: the default whenever they apply (unit vertex weights, target joints
// given, 10 betas, 4 skinning pairs per vertex); SMPLFIT_BM=0 selects the
// wave-per-instance kernels everywhere (they also serve every other configuration).
bool use_bm() {
  const char* e = getenv("SMPLFIT_BM");
  return !(e && e[0] == '0');
}
// Small vertex subsets stay on the wave-per-instance kernels: the pair-Gram and combine passes cost the
// same per instance whatever V is (measured at V = 1024, B = 16384: 4.15 M fits/s batch-major vs 4.63 M).
bool bm_applies(const DevModel& d) {
  return use_bm() && d.KW == 4 && d.S == 10 && d.ngroups > 0 && d.V >= 2048;
}

template <int S, int KW>
void launch_lbs_bm(const DevModel& d, const Workspace& ws, int B, hipStream_t st) {
  const int Mp = (int)align_up((size_t)B, 128);
  const size_t lds = (size_t)kGQ * 12 * 64 * 4;
  if constexpr (KW == 4)
    hipLaunchKernelGGL((k_lbs_partsum_bm<S, 4>), dim3(d.ngroups_used, Mp / 64), dim3(64 * kBW), lds, st, d, ws, B, Mp);
  hipLaunchKernelGGL(k_psum_combine, dim3((B + 255) / 256, d.J), dim3(256), 0, st, d, ws, B, Mp);
}

template <int S, int KW>
int launch_shape_accum(const DevModel& d, const Workspace& ws, int B, bool weighted, hipStream_t st) {
  const dim3 grid((B + kNW - 1) / kNW);
  if (weighted || !use_pair_form()) {
    const size_t lds =
        ((size_t)kNW * d.J * sf::jd_stride(S) + 2 * 64 * sf::cpack_stride(S, KW) + 256 * 12) * 4;
    if (weighted)
      hipLaunchKernelGGL((k_shape_accum<S, KW, true>), grid, dim3(256), lds, st, d, ws, B);
    else
      hipLaunchKernelGGL((k_shape_accum<S, KW, false>), grid, dim3(256), lds, st, d, ws, B);
    return 0;
  }
  const size_t per_wave = ((size_t)d.J * sf::jd_stride(S) + 256 + (d.J + 1) * 4 + 3) / 4 * 4;
  const size_t blob = 64 * sf::cpack_stride(S, KW) + 16 * 64 + 16;
  hipLaunchKernelGGL((k_residual<S, KW>), grid, dim3(256), (kNW * per_wave + 2 * blob) * 4, st, d, ws, B);
  const size_t lds_g = ((size_t)d.J * sf::jd_stride(S) + (size_t)d.jt.np * 9 + (size_t)d.J * 3 * S) * 4;
  hipLaunchKernelGGL(k_pair_gram, dim3(B), dim3(64), lds_g, st, d, ws);
  return 0;
}

template <int S, int KW, int MODE, bool SOLVE>
void launch_lbs(const DevModel& d, const Workspace& ws, int B, bool weighted, int nb,
                const float* beta, const float* trans, float* out, float beta_reg, float beta_reg2,
                hipStream_t st, const float* kid = nullptr) {
  const size_t per_wave =
      ((size_t)d.J * sf::jd_stride(S) + d.J * 4 + 36 + (SOLVE ? sf::solve_scratch_floats(S) : 0) + 3) / 4 * 4;
  const size_t lds = (kNW * per_wave + 2 * 64 * sf::cpack_stride(S, KW)) * 4;
  const dim3 grid((B + kNW - 1) / kNW);
  if (weighted)
    hipLaunchKernelGGL((k_lbs_partsum<S, KW, true, MODE, SOLVE>), grid, dim3(256), lds, st, d, ws, B,
                       nb, beta, trans, kid, out, beta_reg, beta_reg2);
  else
    hipLaunchKernelGGL((k_lbs_partsum<S, KW, false, MODE, SOLVE>), grid, dim3(256), lds, st, d, ws, B,
                       nb, beta, trans, kid, out, beta_reg, beta_reg2);
  if (MODE == 1)
    hipLaunchKernelGGL((k_lbs_rest<S, KW>), dim3(B), dim3(256),
                       ((size_t)d.J * sf::jd_stride(S) + d.J * 4) * 4, st, d, ws);
}

// K0 dispatch: LDS-staged form when the (V,3) row fits in LDS, gather form otherwise.
void launch_center_sort(const DevModel& d, const float* tv, const float* tj, const float* vw,
                        const Workspace& ws, int B, hipStream_t st) {
  // vertices staged in LDS: the whole row, or — when that leaves room for one workgroup per CU only but
  // an 80 KB slice covers >= 15/16 of the row — the slice that lets two workgroups share the CU
  int VL = d.V;
  {
    const int cap = ((80 * 1024 - 64 * 4) / 12) & ~1;
    static const bool two = [] { const char* e = getenv("SMPLFIT_K0_TWO"); return !(e && e[0] == '0'); }();
    if (two && d.V > cap && d.V - cap <= d.V / 16) VL = cap;
  }
  const size_t lds_row = ((size_t)((3 * VL + 3) & ~3) + 64) * 4;
  if (lds_row <= 160 * 1024) {
    // dynamic LDS above 64 KB has to be opted into once per kernel (and per device: the attribute is
    // set again whenever the current device changes; idempotent, so racing threads are harmless)
    static std::once_flag once[16];
    int dev_id = 0;
    (void)hipGetDevice(&dev_id);
    std::call_once(once[dev_id & 15], [] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_center_sort_partsum_lds<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_center_sort_partsum_lds<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    if (vw)
      hipLaunchKernelGGL((k_center_sort_partsum_lds<true>), dim3(B), dim3(1024), lds_row, st, d, tv, tj, vw, ws, VL);
    else
      hipLaunchKernelGGL((k_center_sort_partsum_lds<false>), dim3(B), dim3(1024), lds_row, st, d, tv, tj, vw, ws, VL);
    return;
  }
  const size_t lds0 = ((size_t)4 * d.J * sf::kPsum + 20) * 4;
  if (vw)
    hipLaunchKernelGGL((k_center_sort_partsum<true>), dim3(B), dim3(256), lds0, st, d, tv, tj, vw, ws);
  else
    hipLaunchKernelGGL((k_center_sort_partsum<false>), dim3(B), dim3(256), lds0, st, d, tv, tj, vw, ws);
}

#define SF_DISPATCH_SKW(d, CALL)                                               \
  do {                                                                         \
    if ((d).S == 10 && (d).KW == 4) { CALL(10, 4); }                           \
    else if ((d).S == 10 && (d).KW == 8) { CALL(10, 8); }                      \
    else if ((d).S == 16 && (d).KW == 4) { CALL(16, 4); }                      \
    else if ((d).S == 16 && (d).KW == 8) { CALL(16, 8); }                      \
    else if ((d).S == 11 && (d).KW == 4) { CALL(11, 4); }                      \
    else if ((d).S == 11 && (d).KW == 8) { CALL(11, 8); }                      \
    else if ((d).S == 17 && (d).KW == 4) { CALL(17, 4); }                      \
    else return fail(SMPLFIT_ERR_UNSUPPORTED,                                  \
                     "num_betas must be 10 or 16 (+1 with the kid blend shape)"); \
  } while (0)

size_t chunked_workspace_bytes(const sf::HostTables& t, int batch);

int check_common(const smplfit_handle* h, int batch, void* workspace, size_t workspace_bytes) {
  if (!h) return fail(SMPLFIT_ERR_BAD_ARG, "null handle");
  if (!h->has_device) return fail(SMPLFIT_ERR_HIP, "handle was created host-only (no device)");
  if (batch <= 0) return fail(SMPLFIT_ERR_BAD_ARG, "batch must be positive");
  if (!workspace || ((uintptr_t)workspace & 255))
    return fail(SMPLFIT_ERR_WORKSPACE, "workspace must be a 256-byte aligned device pointer");
  if (workspace_bytes < chunked_workspace_bytes(h->t, batch))
    return fail(SMPLFIT_ERR_WORKSPACE, "workspace too small (see smplfit_workspace_bytes)");
  return 0;
}

int launch_gemm(const DevModel& d, const Workspace& ws, int B, hipStream_t st, bool transposed = false) {
  const int Mp = (int)align_up((size_t)B, 128), N = 3 * d.Vp;
  if (d.Kp == 208) {  // SMPL (J = 24): A-stationary kernel, 104 A registers per lane
    constexpr int NK2 = 104;
    const int ntiles = N / 32;
    // ~768 workgroups (measured faster than exactly one 512-workgroup residency wave)
    int nchunk = std::max(1, (3 * 256 + Mp / 128 - 1) / (Mp / 128));
    nchunk = std::min(nchunk, ntiles);
    const int per = (ntiles + nchunk - 1) / nchunk;
    nchunk = (ntiles + per - 1) / per;
    const size_t lds = (size_t)2 * 32 * (2 * NK2 + 4) * 4;
    if (transposed)
      hipLaunchKernelGGL((k_posedirs_gemm_as<NK2, true>), dim3(nchunk, Mp / 128), dim3(256), lds, st,
                         ws.rp, d.pdSw, d.vtN, ws.vpT, N, per, Mp);
    else
      hipLaunchKernelGGL((k_posedirs_gemm_as<NK2, false>), dim3(nchunk, Mp / 128), dim3(256), lds, st,
                         ws.rp, d.pdSw, d.vtN, ws.vposed, N, per, Mp);
    return 0;
  }
  if (transposed)
    hipLaunchKernelGGL(k_posedirs_gemm<true>, dim3((N / 128) * (Mp / 128)), dim3(256), 0, st, ws.rp, d.pdT,
                       d.vtN, ws.vpT, Mp, N, d.Kp);
  else
    hipLaunchKernelGGL(k_posedirs_gemm<false>, dim3((N / 128) * (Mp / 128)), dim3(256), 0, st, ws.rp, d.pdT,
                       d.vtN, ws.vposed, Mp, N, d.Kp);
  return 0;
}

size_t joint_lds(const DevModel& d, int kind = 0) {
  return (size_t)sf::joint_scratch_floats(d.J, d.S, kind) * 4;
}
size_t solve_lds(const DevModel& d) { return (size_t)sf::solve_scratch_floats(d.S) * 4; }

int post_launch_check() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(SMPLFIT_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
  return 0;
}

// Shared driver of fit / part_rotations / shape_solve.
struct FitOptions {
  int num_iter;
  float beta_reg, beta_reg2, kid_reg;
  int final_adjust;
  int rotations_only;  // stop after the first rotation pass, write G to `orient`
  // warm start (bodyfitter.py:363-382): the first rotation pass runs against the model posed with
  // these values instead of the template, and the ridge pulls towards init_betas / init_kid
  const float* init_pose = nullptr;   // (B,3J) or null (rest pose)
  const float* init_betas = nullptr;  // (B,init_nb) or null
  int init_nb = 0;
  const float* init_kid = nullptr;    // (B) or null
  int share_beta = 0;                 // one shape for the whole batch (pt/lstsq.py:24-26)
  int scale_mode = 0;                 // 1 scale_target, 2 scale_fit: the last solve has a scale unknown
  float scale_reg = 0.f;
  float* scale_out = nullptr;         // (B) scale_corr
};

int run_fit(const smplfit_handle* h, const float* tv, const float* tj, const float* vw,
            const float* jw, int B, const FitOptions& o, float* pose, float* betas, float* trans,
            float* kid, float* orient, float* rel, const Workspace& ws, hipStream_t st) {
  const DevModel& d = h->d;
  const bool joints = tj != nullptr;
  if (!joints && !h->t.has_regressor)
    return fail(SMPLFIT_ERR_BAD_ARG,
                "target_joints omitted but the model has no J_regressor_post_lbs over its vertices");
  const bool vweighted = vw != nullptr;
  // weights enter the shape solve only if both are given (with joints) or vertex weights without
  // joints (bodyfitter.py:1018-1028)
  const bool eff_v = joints ? (vw && jw) : (vw != nullptr);
  const bool eff_j = joints && vw && jw;
  launch_center_sort(d, tv, tj, vw, ws, B, st);
  const bool bm = bm_applies(d) && joints && !vw && !o.rotations_only && !o.scale_mode;
  if (bm) {
    const int Mp = (int)align_up((size_t)B, 128), N = 3 * d.Vp;
    hipLaunchKernelGGL(k_transpose_targets, dim3(N / 64, Mp / 64), dim3(256), 0, st, ws.tvs, ws.tT, B, N, Mp);
  }
  const float* tj_rot = ws.tjc;
  if (!joints) {  // regressed target joints from the centred vertices (bodyfitter.py:1342-1344)
    hipLaunchKernelGGL(k_regress_joints, dim3(B), dim3(64), 0, st, d, ws.tvs, ws.tjreg);
    tj_rot = ws.tjreg;
  }
  JointStageArgs ja{};
  ja.tj = tj_rot;
  ja.jw = jw;
  ja.joint_block = joints ? 1 : 0;
  ja.joint_block_weighted = eff_j ? 1 : 0;
  ja.vertex_sa_closed_form = eff_v ? 0 : 1;
  ja.do_prologue = o.rotations_only ? 0 : 1;
  ja.fit_rotations = 1;
  ja.Gprev = nullptr;
  const bool warm = o.init_pose || o.init_betas;
  const int use_ref = (warm && (o.init_betas || o.init_kid)) ? 1 : 0;
  if (warm) {
    hipLaunchKernelGGL(k_fill_shape, dim3((B + 255) / 256), dim3(256), 0, st, ws, B, d.S, d.jt.n_kid,
                       o.init_betas, std::min(o.init_nb, d.S - d.jt.n_kid), o.init_kid);
    ForwardArgs fa{};
    fa.pose = o.init_pose;
    fa.betas = ws.beta;  // (B,S) incl. the kid column
    fa.nb = d.S;
    fa.joints = ws.rjoints;
    fa.orient = ws.G;
    hipLaunchKernelGGL(k_forward_joint, dim3(B), dim3(64), joint_lds(d), st, d, fa, ws);
    launch_gemm(d, ws, B, st);
#define SF_CALL_LBS(S_, KW_) \
  launch_lbs<S_, KW_, 1, false>(d, ws, B, vweighted, d.S, ws.beta, ws.trans, nullptr, 0.f, 0.f, st)
    SF_DISPATCH_SKW(d, SF_CALL_LBS);
#undef SF_CALL_LBS
    if (!joints)
      hipLaunchKernelGGL(k_regress_joints, dim3(B), dim3(64), 0, st, d, ws.rverts, ws.rjreg);
    ja.rj = joints ? ws.rjoints : ws.rjreg;
    ja.rj_shared = 0;
    ja.Gprev = ws.G;  // compose with the initial orientations
  } else if (joints) {
    ja.rj = d.j_template;
    ja.rj_shared = 1;
  } else {  // template joints regressed from the default mesh: same regressor on a (1,3,Vp) source
    hipLaunchKernelGGL(k_regress_joints, dim3(1), dim3(64), 0, st, d, d.dm, ws.rjreg);
    ja.rj = ws.rjreg;
    ja.rj_shared = 1;
  }
  hipLaunchKernelGGL(k_joint_stage, dim3(B), dim3(64), joint_lds(d), st, d, ja, ws);
  if (o.rotations_only) {
    hipLaunchKernelGGL(k_copy, dim3(256), dim3(256), 0, st, ws.G, orient, (size_t)B * d.J * 9);
    return post_launch_check();
  }
  for (int it = 0; it < o.num_iter; ++it) {
    if (bm) {
      // batch-major vertex block: one transposed GEMM feeds the residual pass and, after the solve,
      // the LBS / part-sum pass of this iteration
      const int Mp = (int)align_up((size_t)B, 128);
      launch_gemm(d, ws, B, st, true);
      const size_t lds = (size_t)kGQ * 12 * 64 * 4;
      hipLaunchKernelGGL((k_residual_bm<10>), dim3(d.ngroups, Mp / 64), dim3(64 * kBW), lds, st, d, ws, B, Mp);
      hipLaunchKernelGGL((k_pair_gram_bm<10>), dim3(kGramChunks, Mp / 64), dim3(64), 0, st, d, ws, B, Mp);
      hipLaunchKernelGGL((k_gram_combine_bm<10>), dim3((B + 255) / 256, 10 + 3 + 3 * d.J + sf::ne_ng(10)),
                         dim3(256), 0, st, d, ws, B, Mp);
    } else {
      launch_gemm(d, ws, B, st);
#define SF_CALL_ACCUM(S_, KW_) launch_shape_accum<S_, KW_>(d, ws, B, eff_v, st)
      SF_DISPATCH_SKW(d, SF_CALL_ACCUM);
#undef SF_CALL_ACCUM
    }
    // K4 stays its own launch: fused into the prologue of the LBS kernel (template flag SOLVE) its
    // ~40 serial barriers stall all four waves of the workgroup and the kernel ran 230 us longer
    const int pair_in = (bm || (!eff_v && use_pair_form())) ? 1 : 0;
    const bool scaled_now = o.scale_mode && it + 1 == o.num_iter;  // only the last solve (:434-455)
    if (scaled_now) {
#define SF_CALL_EXTRAS(S_, KW_)                                                                       \
  hipLaunchKernelGGL((k_scale_extras<S_, KW_>), dim3(B), dim3(64),                                    \
                     (size_t)d.J * sf::jd_stride(S_) * 4, st, d, ws, eff_v ? 1 : 0)
      SF_DISPATCH_SKW(d, SF_CALL_EXTRAS);
#undef SF_CALL_EXTRAS
      ScaledSolveArgs sa{};
      sa.tj = joints ? ws.tjc : nullptr;
      sa.jw = eff_j ? jw : nullptr;
      sa.joint_block = joints ? 1 : 0;
      sa.mode = o.scale_mode;
      sa.pair_form = pair_in;
      sa.use_ref = use_ref;
      sa.beta_reg = o.beta_reg; sa.beta_reg2 = o.beta_reg2; sa.kid_reg = o.kid_reg; sa.scale_reg = o.scale_reg;
      hipLaunchKernelGGL(k_shape_solve_scaled, dim3(B), dim3(64),
                         (size_t)sf::scaled_solve_scratch_floats(d.S) * 4, st, d, ws, sa);
    } else if (o.share_beta) {  // assemble per instance, sum over the batch, solve the sum + own translation
      const int NC = d.S * d.S + d.S;
      hipLaunchKernelGGL(k_shape_solve, dim3(B), dim3(64), solve_lds(d), st, d, ws, o.beta_reg,
                         o.beta_reg2, o.kid_reg, pair_in, use_ref, 1, B);
      hipLaunchKernelGGL(k_share_reduce, dim3(1), dim3(512), 0, st, ws, B, NC);
      hipLaunchKernelGGL(k_shape_solve, dim3(B), dim3(64), solve_lds(d), st, d, ws, o.beta_reg,
                         o.beta_reg2, o.kid_reg, pair_in, use_ref, 2, B);
    } else {
      hipLaunchKernelGGL(k_shape_solve, dim3(B), dim3(64), solve_lds(d), st, d, ws, o.beta_reg,
                         o.beta_reg2, o.kid_reg, pair_in, use_ref);
    }
    const bool last = it + 1 == o.num_iter;
    if (last && !o.final_adjust) break;  // nothing consumes the re-evaluated mesh
    if (bm) {
#define SF_CALL_LBS(S_, KW_) launch_lbs_bm<S_, KW_>(d, ws, B, st)
      SF_DISPATCH_SKW(d, SF_CALL_LBS);
#undef SF_CALL_LBS
    } else if (joints) {
#define SF_CALL_LBS(S_, KW_) \
  launch_lbs<S_, KW_, 0, false>(d, ws, B, vweighted, d.S, ws.beta, ws.trans, nullptr, 0.f, 0.f, st)
      SF_DISPATCH_SKW(d, SF_CALL_LBS);
#undef SF_CALL_LBS
    } else {
#define SF_CALL_LBS(S_, KW_) \
  launch_lbs<S_, KW_, 1, false>(d, ws, B, vweighted, d.S, ws.beta, ws.trans, nullptr, 0.f, 0.f, st)
      SF_DISPATCH_SKW(d, SF_CALL_LBS);
#undef SF_CALL_LBS
      hipLaunchKernelGGL(k_regress_joints, dim3(B), dim3(64), 0, st, d, ws.rverts, ws.rjreg);
    }
    if (last) break;
    ja.rj = joints ? ws.rjoints : ws.rjreg;
    ja.rj_shared = 0;
    ja.Gprev = ws.G;
    hipLaunchKernelGGL(k_joint_stage, dim3(B), dim3(64), joint_lds(d), st, d, ja, ws);
  }
  RefineArgs ra{};
  ra.tj = tj_rot;
  ra.rj_term = joints ? ws.rjoints : ws.rjreg;
  ra.jw = jw;
  ra.final_adjust = o.final_adjust;
  ra.pose = pose;
  ra.betas = betas;
  ra.trans = trans;
  ra.kid = kid;
  ra.orient = orient;
  ra.rel = rel;
  if (o.scale_mode) {
    hipLaunchKernelGGL(k_scale_refs, dim3(B), dim3(64), 0, st, d, ws, tj_rot, o.scale_mode,
                       o.final_adjust ? 1 : 0, joints ? 0 : 1);
    if (o.scale_mode == 1 && o.final_adjust) ra.tj = ws.tjs;  // target joints times the scale
    ra.scaled = o.scale_mode == 2 ? 1 : 0;                      // rest joints times the scale (:1449-1450)
    if (o.scale_out)
      hipLaunchKernelGGL(k_copy, dim3(16), dim3(256), 0, st, ws.scale, o.scale_out, (size_t)B);
  }
  hipLaunchKernelGGL(k_refine_epilogue, dim3(B), dim3(64), joint_lds(d, 1), st, d, ra, ws);
  return post_launch_check();
}

// fit_with_known_shape (bodyfitter.py:655-838): pose and translation (optionally a scale) for given
// shape parameters.  Alternates the LBS forward at the current rotations (forward joint stage, GEMM,
// K5 with the part sums against the target) with the part-rotation stage; then the alignment stage
// and the dependent refinement.  num_iter rotation passes, num_iter + 1 forward passes.
struct KnownShapeOptions {
  int num_iter, final_adjust, scale_fit;
};

int run_fit_known_shape(const smplfit_handle* h, const float* betas, int nb, const float* kid,
                        const float* init_pose, const float* tv, const float* tj, const float* vw,
                        const float* jw, int B, const KnownShapeOptions& o, float* pose, float* trans,
                        float* scale_out, float* orient, float* rel, const Workspace& ws, hipStream_t st) {
  const DevModel& d = h->d;
  const bool joints = tj != nullptr;
  if (!joints && !h->t.has_regressor)
    return fail(SMPLFIT_ERR_BAD_ARG,
                "target_joints omitted but the model has no J_regressor_post_lbs over its vertices");
  const bool vweighted = vw != nullptr;
  launch_center_sort(d, tv, tj, vw, ws, B, st);
  const float* tj_rot = ws.tjc;
  if (!joints) {
    hipLaunchKernelGGL(k_regress_joints, dim3(B), dim3(64), 0, st, d, ws.tvs, ws.tjreg);
    tj_rot = ws.tjreg;
  }
  hipLaunchKernelGGL(k_fill_shape, dim3((B + 255) / 256), dim3(256), 0, st, ws, B, d.S, d.jt.n_kid, betas,
                     std::min(nb, d.S - d.jt.n_kid), kid);
  ForwardArgs fa{};
  fa.pose = init_pose;  // null -> rest pose
  fa.betas = ws.beta;   // (B,S) incl. the kid column: j_ext's last column is kid_J_shapedir
  fa.nb = d.S;
  fa.joints = ws.rjoints;
  fa.orient = ws.G;
  JointStageArgs ja{};
  ja.tj = tj_rot;
  ja.jw = jw;
  ja.fit_rotations = 1;
  ja.do_prologue = 0;
  ja.rj_shared = 0;
  ja.Gprev = ws.G;
  for (int it = 0; it <= o.num_iter; ++it) {
    hipLaunchKernelGGL(k_forward_joint, dim3(B), dim3(64), joint_lds(d), st, d, fa, ws);
    launch_gemm(d, ws, B, st);
    // MODE 1: the posed mesh is kept (regressed joints, alignment sums) next to the part sums
#define SF_CALL_LBS(S_, KW_) \
  launch_lbs<S_, KW_, 1, false>(d, ws, B, vweighted, d.S, ws.beta, ws.trans, nullptr, 0.f, 0.f, st)
    SF_DISPATCH_SKW(d, SF_CALL_LBS);
#undef SF_CALL_LBS
    if (!joints)
      hipLaunchKernelGGL(k_regress_joints, dim3(B), dim3(64), 0, st, d, ws.rverts, ws.rjreg);
    if (it == o.num_iter) break;
    ja.rj = joints ? ws.rjoints : ws.rjreg;
    hipLaunchKernelGGL(k_joint_stage, dim3(B), dim3(64), joint_lds(d), st, d, ja, ws);
    fa.pose = nullptr;
    fa.glob = ws.G;
  }
  ScaleTransArgs sa{};
  sa.tj = joints ? ws.tjc : nullptr;
  // weights enter only if both are given (with joints) / vertex weights alone without joints (:1640-1661)
  sa.weighted_v = joints ? (vw && jw) : (vw != nullptr);
  sa.jw = (joints && vw && jw) ? jw : nullptr;
  sa.with_scale = o.scale_fit;
  sa.regressed = joints ? 0 : 1;
  sa.scale_out = o.scale_fit ? scale_out : nullptr;
  hipLaunchKernelGGL(k_scale_trans, dim3(B), dim3(256), 0, st, d, sa, ws);
  RefineArgs ra{};
  ra.tj = tj_rot;
  ra.rj_term = joints ? ws.rjoints : ws.rjreg;
  ra.jw = jw;
  ra.final_adjust = o.final_adjust;
  ra.pose = pose;
  ra.betas = nullptr;
  ra.trans = trans;
  ra.kid = nullptr;
  ra.orient = orient;
  ra.rel = rel;
  ra.scaled = o.scale_fit;
  hipLaunchKernelGGL(k_refine_epilogue, dim3(B), dim3(64), joint_lds(d, 1), st, d, ra, ws);
  return post_launch_check();
}

template <typename T>
int upload(smplfit_handle* h, const std::vector<T>& src, const T** dst) {
  void* p = nullptr;
  const size_t bytes = std::max<size_t>(src.size() * sizeof(T), 16);
  SF_HIP_TRY(hipMalloc(&p, bytes));
  h->allocs.push_back(p);
  if (!src.empty()) SF_HIP_TRY(hipMemcpy(p, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
  *dst = (const T*)p;
  return 0;
}

// Chunk plan of one fit call: large batches are split into chunks (default 3, SMPLFIT_CHUNKS=1..4) that run concurrently on
// the caller's stream and the handle's side streams, so that the MFMA-bound posedirs GEMM of one
// chunk overlaps the VALU / HBM-bound vertex passes of the others (measured +5 % at B = 4096).
// Chunk sizes are multiples of 128 (the GEMM's instance tile).
int chunk_plan(int batch, int* sizes) {
  static const int want = [] {
    const char* e = getenv("SMPLFIT_CHUNKS");
    int v = e ? atoi(e) : 3;  // measured at B = 4096 (batch-major kernels): 1 chunk 1.23 M fits/s, 2: 1.30 M, 3: 1.31 M, 4: 1.31 M
    return v < 1 ? 1 : (v > kMaxChunks ? kMaxChunks : v);
  }();
  int n = want;
  while (n > 1 && batch < n * 512) --n;  // keep every chunk >= 512 instances
  const int per = ((batch + n - 1) / n + 127) / 128 * 128;
  int left = batch, k = 0;
  while (left > 0 && k < kMaxChunks) {
    sizes[k] = left < per ? left : per;
    left -= sizes[k];
    ++k;
  }
  return k;
}

size_t chunked_workspace_bytes(const sf::HostTables& t, int batch) {
  int sizes[kMaxChunks];
  const int n = chunk_plan(batch, sizes);
  size_t total = 0;
  for (int i = 0; i < n; ++i) total += carve(t, sizes[i], nullptr, nullptr);
  return std::max(total, carve(t, batch, nullptr, nullptr));
}

}  // namespace

// ================================================================================================
// C-ABI
// ================================================================================================
extern "C" {

const char* smplfit_last_error(void) { return g_last_error.c_str(); }
const char* smplfit_version(void) { return "smplfit-hip 0.1 (gfx950)"; }

int smplfit_create(const smplfit_model_desc* desc, int flags, smplfit_handle** out) {
  if (!desc || !out) return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_create: null argument");
  *out = nullptr;
  smplfit_handle* h = new smplfit_handle();
  bool unsupported = false;
  std::string err = sf::build_tables(*desc, h->t, &unsupported);
  if (!err.empty()) {
    delete h;
    return fail(unsupported ? SMPLFIT_ERR_UNSUPPORTED : SMPLFIT_ERR_BAD_ARG, err);
  }
  if (h->t.S + 3 > 3 * h->t.J) {
    delete h;
    return fail(SMPLFIT_ERR_UNSUPPORTED, "smplfit_create: num_betas too large for this joint count");
  }
  if (flags & SMPLFIT_CREATE_HOST_ONLY) {
    *out = h;
    return SMPLFIT_OK;
  }
  const sf::HostTables& t = h->t;
  DevModel& d = h->d;
  d.V = t.V; d.J = t.J; d.S = t.S; d.P = t.P; d.Vp = t.Vp; d.Kp = t.Kp; d.KW = t.KW;
  d.n_used = t.n_used;
  d.nseg = (int)t.segments.size();
  std::vector<int32_t> seg;
  for (auto& s : t.segments) {
    seg.push_back(s.start);
    seg.push_back(s.count);
    seg.push_back(s.part);
  }
  std::vector<float> jtemplate((size_t)t.J * 3);
  for (int k = 0; k < t.J * 3; ++k) jtemplate[k] = t.j_ext[(size_t)k * (t.S + 1)];
  int rc = 0;
  auto up = [&](auto& vec, auto** dst) {
    if (rc == 0) rc = upload(h, vec, dst);
  };
  up(t.perm, &d.perm);
  up(seg, &d.segments);
  std::vector<int32_t> pss(t.J + 1, 0);
  {
    size_t k = 0;
    for (int p = 0; p < t.J; ++p) {
      // segments are ordered by part id over the used parts
      while (k < t.segments.size() && t.segments[k].part < p) ++k;
      pss[p] = (int32_t)k;
      while (k < t.segments.size() && t.segments[k].part == p) ++k;
      pss[p + 1] = (int32_t)k;
    }
  }
  up(pss, &d.part_seg_start);
  up(t.vt, &d.vt);
  up(t.dm, &d.dm);
  up(t.sd, &d.sd);
  up(t.wval, &d.wval);
  up(t.widx, &d.widx);
  up(t.pdT, &d.pdT);
  up(t.pdSw, &d.pdSw);
  up(t.vtN, &d.vtN);
  up(t.cpackA, &d.cpackA);
  up(t.cpackB, &d.cpackB);
  up(t.gblob, &d.gblob);
  std::vector<int32_t> gt;
  for (auto& g : t.gtiles) {
    gt.push_back(g.start);
    gt.push_back(g.count);
    gt.push_back(g.part);
  }
  up(gt, &d.gtiles);
  d.ngt = (int)t.gtiles.size();
  up(jtemplate, &d.j_template);
  up(t.reg_start, &d.reg_start);
  up(t.reg_slot, &d.reg_slot);
  up(t.reg_val, &d.reg_val);
  up(t.reg_rowsum, &d.reg_rowsum);
  {
    std::vector<int32_t> gr;
    d.ngroups = (int)t.groups.size();
    d.ngroups_used = 0;
    for (const auto& g : t.groups) {
      gr.push_back(g.start); gr.push_back(g.count); gr.push_back(g.part); gr.push_back(g.used);
      gr.push_back(g.nq);
      for (int q = 0; q < sf::kGroupJoints; ++q) gr.push_back(g.joints[q]);
      if (g.used) ++d.ngroups_used;  // used parts come first in slot order
    }
    up(gr, &d.groups);
    std::vector<int32_t> mb_start(t.J + 1, 0), mb_row;
    for (int j = 0; j < t.J; ++j) {
      for (size_t g = 0; g < t.groups.size(); ++g)
        for (int q = 0; q < t.groups[g].nq; ++q)
          if (t.groups[g].joints[q] == j) mb_row.push_back((int32_t)(g * (16 + 3 * sf::kGroupJoints) + 16 + 3 * q));
      mb_start[j + 1] = (int32_t)mb_row.size();
    }
    up(mb_start, &d.mb_start);
    up(mb_row, &d.mb_row);
    up(t.brec, &d.brec);
    up(t.pair_c1x, &d.pair_c1x);
  }
  sf::JointTabs& jt = d.jt;
  jt.J = t.J; jt.S = t.S; jt.num_levels = t.num_levels(); jt.adj_last_level = t.adj_last_level;
  jt.P = t.P; jt.Kp = t.Kp;
  jt.n_kid = t.n_kid;
  up(t.parents, &jt.parents);
  up(t.fk_js, &jt.fk_js);
  up(t.fk_level_start, &jt.fk_level_start);
  up(t.cas_start, &jt.cas_start);
  up(t.cas_flat, &jt.cas_flat);
  up(t.part_type, &jt.part_type);
  up(t.toe_src, &jt.toe_src);
  up(t.adj_level_start, &jt.adj_level_start);
  up(t.adj_parts, &jt.adj_parts);
  up(t.j_ext, &jt.j_ext);
  up(t.bone_ext, &jt.bone_ext);
  up(t.cs_joint, &jt.cs_joint);
  up(t.cw_joint, &jt.cw_joint);
  jt.np = (int)t.pair_c3.size();
  up(t.pair_j, &jt.pair_j);
  up(t.pair_c1, &jt.pair_c1);
  up(t.pair_c2, &jt.pair_c2);
  up(t.pair_c3, &jt.pair_c3);
  up(t.diag_g0, &jt.diag_g0);
  up(t.diag_c2, &jt.diag_c2);
  up(t.diag_c3, &jt.diag_c3);
  if (rc != 0) {
    smplfit_destroy(h);
    return rc;
  }
  // side streams + events of the chunked fit
  for (int i = 0; i < kMaxChunks - 1; ++i) {
    if (hipStreamCreateWithFlags(&h->side[i], hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_join[i], hipEventDisableTiming) != hipSuccess) {
      smplfit_destroy(h);
      return fail(SMPLFIT_ERR_HIP, "smplfit_create: could not create side streams");
    }
  }
  if (hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess) {
    smplfit_destroy(h);
    return fail(SMPLFIT_ERR_HIP, "smplfit_create: could not create events");
  }
  h->have_streams = true;
  h->has_device = true;
  *out = h;
  return SMPLFIT_OK;
}

void smplfit_destroy(smplfit_handle* h) {
  if (!h) return;
  for (int i = 0; i < kMaxChunks - 1; ++i) {
    if (h->side[i]) {
      (void)hipStreamSynchronize(h->side[i]);
      (void)hipStreamDestroy(h->side[i]);
    }
    if (h->ev_join[i]) (void)hipEventDestroy(h->ev_join[i]);
  }
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  for (void* p : h->allocs) (void)hipFree(p);
  delete h;
}

int smplfit_get_info(const smplfit_handle* h, smplfit_info* info) {
  if (!h || !info) return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_get_info: null argument");
  const sf::HostTables& t = h->t;
  info->num_vertices = t.V;
  info->num_joints = t.J;
  info->num_betas = t.S - t.n_kid;
  info->has_kid = t.n_kid;
  info->padded_vertices = t.Vp;
  info->num_used_vertices = t.n_used;
  info->skin_width = t.KW;
  info->num_segments = (int)t.segments.size();
  info->num_fk_levels = t.num_levels();
  info->adj_last_level = t.adj_last_level;
  info->has_device = h->has_device ? 1 : 0;
  return SMPLFIT_OK;
}

int smplfit_get_table(const smplfit_handle* h, int table_id, int32_t* dst, size_t cap, size_t* n) {
  if (!h || !n) return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_get_table: null argument");
  const sf::HostTables& t = h->t;
  std::vector<int32_t> tmp;
  const std::vector<int32_t>* src = nullptr;
  switch (table_id) {
    case SMPLFIT_TAB_PART_ASSIGNMENT: src = &t.part_assignment; break;
    case SMPLFIT_TAB_SORT_PERM: src = &t.perm; break;
    case SMPLFIT_TAB_PART_TYPE: src = &t.part_type; break;
    case SMPLFIT_TAB_FK_ORDER: src = &t.fk_js; break;
    case SMPLFIT_TAB_FK_LEVEL_START: src = &t.fk_level_start; break;
    case SMPLFIT_TAB_ADJ_FLAG: src = &t.adj_flag; break;
    case SMPLFIT_TAB_USED_PART: src = &t.used_part; break;
    case SMPLFIT_TAB_SEGMENTS:
      for (auto& s : t.segments) {
        tmp.push_back(s.start);
        tmp.push_back(s.count);
        tmp.push_back(s.part);
      }
      src = &tmp;
      break;
    default: return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_get_table: unknown table id");
  }
  *n = src->size();
  if (dst) std::memcpy(dst, src->data(), std::min(cap, src->size()) * sizeof(int32_t));
  return SMPLFIT_OK;
}

size_t smplfit_workspace_bytes(const smplfit_handle* h, int batch) {
  if (!h || batch <= 0) return 0;
  return chunked_workspace_bytes(h->t, batch);
}

int smplfit_fit_f32(const smplfit_handle* h, const float* target_vertices,
                    const float* target_joints, const float* vertex_weights,
                    const float* joint_weights, int batch, int num_iter, float beta_regularizer,
                    float beta_regularizer2, float kid_regularizer, int final_adjust_rots,
                    float* pose_rotvecs, float* shape_betas, float* trans, float* kid_factor,
                    float* orientations, float* relative_orientations, void* workspace,
                    size_t workspace_bytes, void* hip_stream) {
  return smplfit_fit_warm_f32(h, target_vertices, target_joints, vertex_weights, joint_weights, batch,
                              num_iter, beta_regularizer, beta_regularizer2, kid_regularizer,
                              final_adjust_rots, nullptr, nullptr, 0, nullptr, pose_rotvecs, shape_betas,
                              trans, kid_factor, orientations, relative_orientations, workspace,
                              workspace_bytes, hip_stream);
}

int smplfit_fit_warm_f32(const smplfit_handle* h, const float* target_vertices,
                         const float* target_joints, const float* vertex_weights,
                         const float* joint_weights, int batch, int num_iter, float beta_regularizer,
                         float beta_regularizer2, float kid_regularizer, int final_adjust_rots,
                         const float* initial_pose_rotvecs, const float* initial_shape_betas,
                         int num_initial_betas, const float* initial_kid_factor, float* pose_rotvecs,
                         float* shape_betas, float* trans, float* kid_factor, float* orientations,
                         float* relative_orientations, void* workspace, size_t workspace_bytes,
                         void* hip_stream) {
  smplfit_fit_args a{};
  a.target_vertices = target_vertices; a.target_joints = target_joints;
  a.vertex_weights = vertex_weights; a.joint_weights = joint_weights;
  a.batch = batch; a.num_iter = num_iter;
  a.beta_regularizer = beta_regularizer; a.beta_regularizer2 = beta_regularizer2;
  a.kid_regularizer = kid_regularizer; a.final_adjust_rots = final_adjust_rots;
  a.initial_pose_rotvecs = initial_pose_rotvecs; a.initial_shape_betas = initial_shape_betas;
  a.num_initial_betas = num_initial_betas; a.initial_kid_factor = initial_kid_factor;
  a.pose_rotvecs = pose_rotvecs; a.shape_betas = shape_betas; a.trans = trans; a.kid_factor = kid_factor;
  a.orientations = orientations; a.relative_orientations = relative_orientations;
  a.workspace = workspace; a.workspace_bytes = workspace_bytes; a.hip_stream = hip_stream;
  return smplfit_fit_ex_f32(h, &a);
}

int smplfit_fit_ex_f32(const smplfit_handle* h, const smplfit_fit_args* args) {
  if (!args) return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_fit_ex_f32: null arguments");
  const float *target_vertices = args->target_vertices, *target_joints = args->target_joints,
              *vertex_weights = args->vertex_weights, *joint_weights = args->joint_weights,
              *initial_pose_rotvecs = args->initial_pose_rotvecs,
              *initial_shape_betas = args->initial_shape_betas, *initial_kid_factor = args->initial_kid_factor;
  const int batch = args->batch, num_iter = args->num_iter, final_adjust_rots = args->final_adjust_rots,
            num_initial_betas = args->num_initial_betas;
  const float beta_regularizer = args->beta_regularizer, beta_regularizer2 = args->beta_regularizer2,
              kid_regularizer = args->kid_regularizer;
  float *pose_rotvecs = args->pose_rotvecs, *shape_betas = args->shape_betas, *trans = args->trans,
        *kid_factor = args->kid_factor, *orientations = args->orientations,
        *relative_orientations = args->relative_orientations;
  void *workspace = args->workspace, *hip_stream = args->hip_stream;
  const size_t workspace_bytes = args->workspace_bytes;
  int rc = check_common(h, batch, workspace, workspace_bytes);
  if (rc) return rc;
  if (!target_vertices || !pose_rotvecs || !shape_betas || !trans)
    return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_fit_f32: null input/output pointer");
  if (num_iter < 1) return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_fit_f32: num_iter must be >= 1");
  if (initial_kid_factor && !h->t.n_kid)
    return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_fit_warm_f32: initial_kid_factor given to a handle without kid");
  if (initial_shape_betas && num_initial_betas < 0)
    return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_fit_warm_f32: negative num_initial_betas");
  const int inb = initial_shape_betas ? num_initial_betas : 0;
  FitOptions o{num_iter, beta_regularizer, beta_regularizer2, kid_regularizer,
               final_adjust_rots ? 1 : 0, 0};
  o.share_beta = args->share_beta ? 1 : 0;
  if (args->scale_mode < 0 || args->scale_mode > 2)
    return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_fit_ex_f32: scale_mode must be 0, 1 (scale_target) or 2 (scale_fit)");
  if (args->scale_mode && !args->scale_corr)
    return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_fit_ex_f32: a scale option needs the scale_corr output");
  if (args->scale_mode && o.share_beta)
    return fail(SMPLFIT_ERR_UNSUPPORTED,
                "smplfit_fit_ex_f32: share_beta together with a scale unknown (partially shared solve, "
                "pt/lstsq.py:32-90) is not implemented");
  o.scale_mode = args->scale_mode;
  o.scale_reg = args->scale_regularizer;
  hipStream_t st = (hipStream_t)hip_stream;
  int sizes[kMaxChunks];
  // share_beta couples all instances in every shape solve: one chunk
  const int nchunk = (h->have_streams && !o.share_beta) ? chunk_plan(batch, sizes) : 1;
  const int J = h->t.J, V = h->t.V, Sb = h->t.S - h->t.n_kid;
  auto run_chunk = [&](int b0, int nb, char* wsbase, hipStream_t cs) -> int {
    Workspace ws;
    carve(h->t, nb, wsbase, &ws);
    FitOptions oc = o;
    oc.init_pose = initial_pose_rotvecs ? initial_pose_rotvecs + (size_t)b0 * J * 3 : nullptr;
    oc.init_betas = initial_shape_betas ? initial_shape_betas + (size_t)b0 * inb : nullptr;
    oc.init_nb = inb;
    oc.init_kid = initial_kid_factor ? initial_kid_factor + b0 : nullptr;
    oc.scale_out = args->scale_corr ? args->scale_corr + b0 : nullptr;
    return run_fit(h, target_vertices + (size_t)b0 * V * 3,
                   target_joints ? target_joints + (size_t)b0 * J * 3 : nullptr,
                   vertex_weights ? vertex_weights + (size_t)b0 * V : nullptr,
                   joint_weights ? joint_weights + (size_t)b0 * J : nullptr, nb, oc,
                   pose_rotvecs + (size_t)b0 * J * 3, shape_betas + (size_t)b0 * Sb,
                   trans + (size_t)b0 * 3, kid_factor ? kid_factor + b0 : nullptr,
                   orientations ? orientations + (size_t)b0 * J * 9 : nullptr,
                   relative_orientations ? relative_orientations + (size_t)b0 * J * 9 : nullptr, ws, cs);
  };
  if (nchunk <= 1) return run_chunk(0, batch, (char*)workspace, st);
  // fork: every chunk is an independent fit with its own workspace slice; chunk 0 stays on the
  // caller's stream, the others go to the handle's side streams and are joined back by events
  // (stream-ordered with respect to the caller, hipGraph-capturable).  The handle's streams and
  // events are shared state: concurrent fit calls on one handle serialise their ENQUEUE here.
  std::lock_guard<std::mutex> lock(h->mu);
  SF_HIP_TRY(hipEventRecord(h->ev_fork, st));
  char* wsp = (char*)workspace;
  int b0 = 0;
  for (int c = 0; c < nchunk; ++c) {
    hipStream_t cs = c == 0 ? st : h->side[c - 1];
    if (c > 0) SF_HIP_TRY(hipStreamWaitEvent(cs, h->ev_fork, 0));
    rc = run_chunk(b0, sizes[c], wsp, cs);
    if (rc) return rc;
    if (c > 0) {
      SF_HIP_TRY(hipEventRecord(h->ev_join[c - 1], cs));
      SF_HIP_TRY(hipStreamWaitEvent(st, h->ev_join[c - 1], 0));
    }
    wsp += carve(h->t, sizes[c], nullptr, nullptr);
    b0 += sizes[c];
  }
  return SMPLFIT_OK;
}

int smplfit_fit_known_shape_f32(const smplfit_handle* h, const float* shape_betas,
                                int num_betas_given, const float* kid_factor,
                                const float* initial_pose_rotvecs, const float* target_vertices,
                                const float* target_joints, const float* vertex_weights,
                                const float* joint_weights, int batch, int num_iter,
                                int final_adjust_rots, int scale_fit, float* pose_rotvecs, float* trans,
                                float* scale_corr, float* orientations, float* relative_orientations,
                                void* workspace, size_t workspace_bytes, void* hip_stream) {
  int rc = check_common(h, batch, workspace, workspace_bytes);
  if (rc) return rc;
  if (!shape_betas || !target_vertices || !pose_rotvecs || !trans)
    return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_fit_known_shape_f32: null input/output pointer");
  if (num_iter < 1) return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_fit_known_shape_f32: num_iter must be >= 1");
  if (scale_fit && !scale_corr)
    return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_fit_known_shape_f32: scale_fit needs the scale_corr output");
  const sf::HostTables& t = h->t;
  if (kid_factor && !t.n_kid)
    return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_fit_known_shape_f32: kid_factor given to a handle without kid");
  if (num_betas_given < 0 || num_betas_given > t.S - t.n_kid)
    return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_fit_known_shape_f32: more betas than the model holds; slice first");
  KnownShapeOptions o{num_iter, final_adjust_rots ? 1 : 0, scale_fit ? 1 : 0};
  hipStream_t st = (hipStream_t)hip_stream;
  Workspace ws;
  carve(t, batch, (char*)workspace, &ws);
  return run_fit_known_shape(h, shape_betas, num_betas_given, kid_factor, initial_pose_rotvecs,
                             target_vertices, target_joints, vertex_weights, joint_weights, batch, o,
                             pose_rotvecs, trans, scale_corr, orientations, relative_orientations, ws, st);
}

int smplfit_part_rotations_f32(const smplfit_handle* h, const float* target_vertices,
                               const float* target_joints, const float* vertex_weights,
                               const float* joint_weights, int batch, float* glob_rotmats,
                               void* workspace, size_t workspace_bytes, void* hip_stream) {
  int rc = check_common(h, batch, workspace, workspace_bytes);
  if (rc) return rc;
  if (!target_vertices || !glob_rotmats)
    return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_part_rotations_f32: null pointer");
  Workspace ws;
  carve(h->t, batch, (char*)workspace, &ws);
  FitOptions o{1, 0.f, 0.f, 0.f, 0, 1};
  return run_fit(h, target_vertices, target_joints, vertex_weights, joint_weights, batch, o, nullptr,
                 nullptr, nullptr, nullptr, glob_rotmats, nullptr, ws, (hipStream_t)hip_stream);
}

int smplfit_forward_f32(const smplfit_handle* h, const float* pose_rotvecs,
                        const float* glob_rotmats, const float* shape_betas, int num_betas_given,
                        const float* trans, const float* kid_factor, int batch, float* vertices,
                        float* joints, float* orientations, void* workspace, size_t workspace_bytes,
                        void* hip_stream) {
  int rc = check_common(h, batch, workspace, workspace_bytes);
  if (rc) return rc;
  if (pose_rotvecs && glob_rotmats)
    return fail(SMPLFIT_ERR_BAD_ARG, "Only one rotation input may be provided");
  if (!joints) return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_forward_f32: joints output is required");
  const DevModel& d = h->d;
  hipStream_t st = (hipStream_t)hip_stream;
  Workspace ws;
  carve(h->t, batch, (char*)workspace, &ws);
  ForwardArgs fa{};
  fa.pose = pose_rotvecs;
  fa.glob = glob_rotmats;
  fa.betas = shape_betas;
  fa.nb = shape_betas ? std::min(num_betas_given, d.S - d.jt.n_kid) : 0;
  if (kid_factor && !d.jt.n_kid)
    return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_forward_f32: kid_factor given to a handle without kid");
  fa.kid = kid_factor;
  if (shape_betas && num_betas_given > d.S - d.jt.n_kid)
    return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_forward_f32: more betas than the model holds; slice first");
  fa.trans = trans;
  fa.joints = joints;
  fa.orient = orientations;
  hipLaunchKernelGGL(k_forward_joint, dim3(batch), dim3(64), joint_lds(d), st, d, fa, ws);
  if (vertices) {
    launch_gemm(d, ws, batch, st);
#define SF_CALL_LBS(S_, KW_) \
  launch_lbs<S_, KW_, 2, false>(d, ws, batch, false, fa.nb, shape_betas, trans, vertices, 0.f, 0.f, st, kid_factor)
    SF_DISPATCH_SKW(d, SF_CALL_LBS);
#undef SF_CALL_LBS
  }
  return post_launch_check();
}

int smplfit_shape_solve_f32(const smplfit_handle* h, const float* glob_rotmats,
                            const float* target_vertices, const float* target_joints,
                            const float* vertex_weights, const float* joint_weights, int batch,
                            float beta_regularizer, float beta_regularizer2, float kid_regularizer,
                            int add_mean, float* shape_betas, float* trans, float* kid_factor,
                            float* vertices_out, float* joints_out, void* workspace,
                            size_t workspace_bytes, void* hip_stream) {
  int rc = check_common(h, batch, workspace, workspace_bytes);
  if (rc) return rc;
  if (!glob_rotmats || !target_vertices || !shape_betas || !trans)
    return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_shape_solve_f32: null pointer");
  const DevModel& d = h->d;
  hipStream_t st = (hipStream_t)hip_stream;
  Workspace ws;
  carve(h->t, batch, (char*)workspace, &ws);
  const bool joints = target_joints != nullptr;
  const bool eff_v = joints ? (vertex_weights && joint_weights) : (vertex_weights != nullptr);
  const bool eff_j = joints && vertex_weights && joint_weights;
  launch_center_sort(d, target_vertices, target_joints, vertex_weights, ws, batch, st);
  JointStageArgs ja{};
  ja.tj = joints ? ws.tjc : ws.tjreg;  // unused without the joint block
  ja.rj = nullptr;
  ja.rj_shared = 1;
  ja.Gprev = glob_rotmats;
  ja.jw = joint_weights;
  ja.fit_rotations = 0;
  ja.do_prologue = 1;
  ja.joint_block = joints ? 1 : 0;
  ja.joint_block_weighted = eff_j ? 1 : 0;
  ja.vertex_sa_closed_form = eff_v ? 0 : 1;
  if (!joints) hipMemsetAsync(ws.tjreg, 0, (size_t)batch * d.J * 3 * 4, st);
  hipLaunchKernelGGL(k_joint_stage, dim3(batch), dim3(64), joint_lds(d), st, d, ja, ws);
  launch_gemm(d, ws, batch, st);
#define SF_CALL_ACCUM(S_, KW_) launch_shape_accum<S_, KW_>(d, ws, batch, eff_v, st)
  SF_DISPATCH_SKW(d, SF_CALL_ACCUM);
#undef SF_CALL_ACCUM
  hipLaunchKernelGGL(k_shape_solve, dim3(batch), dim3(64), solve_lds(d), st, d, ws, beta_regularizer,
                     beta_regularizer2, kid_regularizer, (!eff_v && use_pair_form()) ? 1 : 0, 0);
  hipLaunchKernelGGL(k_emit_solution, dim3((batch + 255) / 256), dim3(256), 0, st, ws, batch, d.S,
                     d.jt.n_kid, add_mean, shape_betas, trans, kid_factor);
  if (joints_out)
    hipLaunchKernelGGL(k_copy, dim3(64), dim3(256), 0, st, ws.rjoints, joints_out,
                       (size_t)batch * d.J * 3);
  if (vertices_out) {
#define SF_CALL_LBS(S_, KW_) \
  launch_lbs<S_, KW_, 2, false>(d, ws, batch, false, d.S, ws.beta, ws.trans, vertices_out, 0.f, 0.f, st)
    SF_DISPATCH_SKW(d, SF_CALL_LBS);
#undef SF_CALL_LBS
  }
  return post_launch_check();
}

int smplfit_time_kernel_f32(const smplfit_handle* h, int kernel_id, int batch, int reps,
                            void* workspace, size_t workspace_bytes, void* hip_stream,
                            float* avg_ms) {
  int rc = check_common(h, batch, workspace, workspace_bytes);
  if (rc) return rc;
  if (!avg_ms || reps < 1) return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_time_kernel_f32: bad argument");
  const DevModel& d = h->d;
  hipStream_t st = (hipStream_t)hip_stream;
  Workspace ws;
  carve(h->t, batch, (char*)workspace, &ws);
  hipEvent_t e0, e1;
  SF_HIP_TRY(hipEventCreate(&e0));
  SF_HIP_TRY(hipEventCreate(&e1));
  const bool bm = bm_applies(d);  // time the kernels the default fit runs
  const int Mp = (int)align_up((size_t)batch, 128);
  auto once = [&]() -> int {
    switch (kernel_id) {
      case SMPLFIT_KERNEL_POSEDIRS_GEMM: return launch_gemm(d, ws, batch, st, bm);
      case SMPLFIT_KERNEL_PAIR_GRAM:
        if (!bm) return fail(SMPLFIT_ERR_UNSUPPORTED, "pair-Gram kernel: batch-major path not active");
        hipLaunchKernelGGL((k_pair_gram_bm<10>), dim3(kGramChunks, Mp / 64), dim3(64), 0, st, d, ws, batch, Mp);
        return 0;
      case SMPLFIT_KERNEL_TRANSPOSE:
        if (!bm) return fail(SMPLFIT_ERR_UNSUPPORTED, "transpose kernel: batch-major path not active");
        hipLaunchKernelGGL(k_transpose_targets, dim3(3 * d.Vp / 64, Mp / 64), dim3(256), 0, st, ws.tvs, ws.tT,
                           batch, 3 * d.Vp, Mp);
        return 0;
      case SMPLFIT_KERNEL_SHAPE_ACCUM: {
        if (bm) {
          hipLaunchKernelGGL((k_residual_bm<10>), dim3(d.ngroups, Mp / 64), dim3(64 * kBW),
                             (size_t)kGQ * 12 * 64 * 4, st, d, ws, batch, Mp);
          return 0;
        }
#define SF_CALL_ACCUM(S_, KW_) launch_shape_accum<S_, KW_>(d, ws, batch, false, st)
        SF_DISPATCH_SKW(d, SF_CALL_ACCUM);
#undef SF_CALL_ACCUM
        return 0;
      }
      case SMPLFIT_KERNEL_SHAPE_SOLVE:
        hipLaunchKernelGGL(k_shape_solve, dim3(batch), dim3(64), solve_lds(d), st, d, ws, 1.0f, 0.0f, 1.0f,
                           use_pair_form() ? 1 : 0, 0);
        return 0;
      case SMPLFIT_KERNEL_LBS_PARTSUM: {
        if (bm) {
          launch_lbs_bm<10, 4>(d, ws, batch, st);
          return 0;
        }
#define SF_CALL_LBS(S_, KW_) \
  launch_lbs<S_, KW_, 0, false>(d, ws, batch, false, d.S, ws.beta, ws.trans, nullptr, 0.f, 0.f, st)
        SF_DISPATCH_SKW(d, SF_CALL_LBS);
#undef SF_CALL_LBS
        return 0;
      }
      default: return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_time_kernel_f32: unknown kernel id");
    }
  };
  // Each timed launch runs right after the kernel that precedes it inside a fit (K2 before K3, K3
  // before K5), so caches are in the state the kernel sees in situ; only the target kernel is
  // bracketed by the two events.
  auto pre = [&]() {
    if (kernel_id == SMPLFIT_KERNEL_SHAPE_ACCUM) launch_gemm(d, ws, batch, st, bm);
    if (kernel_id == SMPLFIT_KERNEL_LBS_PARTSUM && bm)
      hipLaunchKernelGGL(k_shape_solve, dim3(batch), dim3(64), solve_lds(d), st, d, ws, 1.0f, 0.0f, 1.0f, 1, 0);
    if (kernel_id == SMPLFIT_KERNEL_LBS_PARTSUM && !bm) {
#define SF_CALL_ACCUM(S_, KW_) launch_shape_accum<S_, KW_>(d, ws, batch, false, st)
      SF_DISPATCH_SKW(d, SF_CALL_ACCUM);
#undef SF_CALL_ACCUM
    }
    return 0;
  };
  rc = pre();
  if (rc) return rc;
  rc = once();  // warm-up
  if (rc) return rc;
  float total = 0.f;
  for (int r = 0; r < reps; ++r) {
    pre();
    SF_HIP_TRY(hipEventRecord(e0, st));
    once();
    SF_HIP_TRY(hipEventRecord(e1, st));
    SF_HIP_TRY(hipEventSynchronize(e1));
    float ms1 = 0.f;
    SF_HIP_TRY(hipEventElapsedTime(&ms1, e0, e1));
    total += ms1;
  }
  const float ms = total;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *avg_ms = ms / (float)reps;
  return post_launch_check();
}

}  // extern "C"
