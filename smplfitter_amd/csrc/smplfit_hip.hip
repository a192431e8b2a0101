// libsmplfit_hip.so — HIP kernels (gfx950 / CDNA4) and the C-ABI of include/smplfit.h.
//
// Kernel inventory.  Wave-per-instance kernels (every configuration; grid = instances unless noted):
//   k_center_sort_partsum(_lds)  K0  mean-centre targets, re-order vertices by body part (SoA), part sums against
//                                    the template mesh
//   k_joint_stage          K1  part rotations (SO(3) projections, swing-twist), shape prologue (FK + beta-Jacobian,
//                              pose feature, joint normal equations)
//   k_posedirs_gemm*       K2  v_posed = v_template + pose_feature . posedirs: split-bf16 on the matrix cores
//                              (k_posedirs_gemm_bf16x3, _tiled for K > 208) or fp32 MFMA (_as, generic)
//   k_shape_accum          K3  vertex block of the normal equations (weighted / non-batch-major configurations)
//   k_shape_solve          K4  fp64 centring + Cholesky + translation (share_beta: + k_share_reduce; scale options:
//                              k_scale_extras + k_shape_solve_scaled + k_scale_refs)
//   k_lbs_partsum          K5  vertices at the solved shape fused with the part sums of the next rotation pass
//   k_refine_epilogue      K6  dependent rotation refinement + relative rotations + log map
//   k_forward_joint, k_lbs_partsum<MODE 2>   BodyModel.forward;  k_scale_trans  known-shape alignment
// Batch-major kernels (LANE = INSTANCE; the default vertex block where they apply, see bm_applies): k_layout_targets,
//   k_mean_finish, k_template_partsum_bm, k_residual_bm, k_pair_gram_bm, k_gram_combine_bm, k_lbs_partsum_bm,
//   k_psum_combine, k_regress_joints_bm, k_transpose_targets (joint rows) — grid = (vertex group | unit chunk) x
//   instance blocks of 64; k_transfer_bm / k_transfer_rows: topology transfer (BodyConverter).
// Everything is enqueued on the caller's stream; no host synchronisation, no allocation.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>
#include <type_traits>

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
  const int32_t* inv_slot;  // (V) sorted slot of every original vertex
  const int32_t* segments;  // (nseg,3)
  const int32_t* part_seg_start;  // (J+1) first segment of each part (empty range: unused part)
  const float *vt, *dm, *sd, *wval, *pdSw, *j_template, *cpackA, *cpackB, *gblob;
  const uint16_t* pdB;  // split-bf16 tile images of posedirs (Kp == 208), see HostTables::pdB
  const uint16_t* pdB2; // split-bf16 stage images for the tiled GEMM (Kp != 208), HostTables::pdB2
  int kc32;
  int gemm_exclusive;  // the split-bf16 GEMM kernels own whole CUs (smplfit_handle::gemm_vgprs >= 256)
  const int32_t* gtiles;  // (ngt,3) start, count, part
  int ngt;
  const uint32_t* widx;
  const int32_t *reg_start, *reg_slot;
  const float* reg_val;
  const float* reg_rowsum;
  // batch-major vertex kernels (HostTables::vpieces / brec; the share tables travel as ShareView arguments)
  int bm_tables;              // the model has batch-major tables (<= 4 skinning weights per vertex)
  const float* brec;          // (Vp, brec_stride) per-slot records: shapedirs + 4 weights in the piece's joint order
  const float *pair_E, *pair_c2e, *diag_c2e;  // constants of k_pair_gram_bm (HostTables)
  const int32_t *jn_start, *jn;               // neighbours of every joint (HostTables::jn)
  // GENERAL path (more than 16 betas / more than 8 skinning weights per vertex, kernels_gen.inc)
  int general;
  const float* sdg;        // (Vp, 3, S4) shapedirs, vertex-major, rows padded to a multiple of four
  const int32_t* segall;   // (nsegall, 3) part-aligned tiles over every slot
  int nsegall;
  // k_prologue_bm: the ancestors of every joint from the root down (CSR; the joint itself excluded)
  const int32_t *anc_start, *anc;
};

// One cell table on the device (sf::ShareTable) as the kernels take it, with the multiplier a launch picked.
struct ShareView {
  const int32_t* piece_start;  // (ncells + 1)
  const int32_t* pieces;       // (npieces + 1, kPieceRec)
  // LBS tables: rows of every part, CSR (part_row_start (J + 1), part_rows); residual tables: the resP rows holding a
  // joint's residual moments, CSR (mb_start (J + 1), mb_row)
  const int32_t *aux_start, *aux_rows;
  int ncells, nrows;
  int rec;   // ints per piece record: 12, or 24 for pieces of up to eight joints (sf::HostTables::piece_rec)
  int mult;  // cells per wave: share s walks the cells [s * mult, (s + 1) * mult)
  int fine;  // the fine table of its kind (small batches): the combine kernels split the rows over waves
  int max_aux;  // longest CSR run of aux_rows (a joint's moment rows / a part's rows)
  // residual tables: the CSR as rows of aux_pitch (max_aux rounded up to 16) entries per joint, -1 behind a joint's last
  // row (k_solve_bm)
  const int32_t* aux_pad;
  int aux_pitch;
};

}  // namespace

constexpr int kMaxChunks = 4;  // batch chunks of one fit call run on the caller's stream + 3 side streams

// k_refine_bm: the wave that owns an adjustable part (parts are grouped by their top-most adjustable ancestor, refine_bm_groups)
constexpr int kRefMaxAdj = 16;
struct RefGroups { int8_t wave[kRefMaxAdj]; };

struct smplfit_handle {
  sf::HostTables t;
  DevModel d{};
  // the cell tables of the batch-major vertex kernels on the device: views[kind]
  std::vector<ShareView> views;
  std::vector<void*> allocs;
  bool has_device = false;
  int refine_group_max = 0;  // most adjustable parts one wave of k_refine_bm gets (groups under one top-most adjustable ancestor)
  RefGroups refine_groups{};
  int8_t rot_slots[64];      // k_rotations_bm: a joint's slot among the joints a toe copies, or -1
  int rot_nslots = 0, rot_toes_per_wave = 0;
  // registers per lane the split-bf16 GEMM kernels were built with (hipFuncGetAttributes at create).  They must own
  // the whole register file of a CU (256 x 8 waves, see k_posedirs_gemm_bf16x3 "exclusive CU"); if a toolchain ever
  // allocates fewer, the fp32-MFMA GEMM is used instead
  int gemm_vgprs = 0;
  // fork/join resources of the chunked fit (see smplfit_fit_f32); guarded by `mu`
  hipStream_t side[kMaxChunks - 1] = {};
  hipEvent_t ev_fork = nullptr, ev_join[kMaxChunks - 1] = {};
  bool have_streams = false;
  mutable std::mutex mu;
};

// Topology-transfer matrix (BodyConverter's vertex_converter_csr, pt/bodyconverter.py:31-47): host + device CSR.
struct smplfit_transfer {
  int v_in = 0, v_out = 0;
  std::vector<int32_t> indptr, indices;
  std::vector<float> values;
  int32_t *d_indptr = nullptr, *d_indices = nullptr;
  float* d_values = nullptr;
};

// Fused conversion plan (smplfit_convert_f32): the transfer matrix re-indexed to the sorted slots of the two models.
struct smplfit_convert_plan {
  const smplfit_handle *in = nullptr, *out = nullptr;
  int nslab = 0;
  int32_t *d_oslot = nullptr, *d_start = nullptr, *d_islot = nullptr;
  float* d_w = nullptr;
};

namespace {

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

// the per-instance stages of sf_stages.h run with one wave per instance: lane / n / sync / sum_to_last
struct DevCtx {
  int lane, n;
  __device__ __forceinline__ void sync() const { __syncthreads(); }
  __device__ __forceinline__ float sum_to_last(float v) const { return wave_sum_last(v); }  // (n == 64: one wave)
#ifdef SMPLFIT_STAGE_STAMPS
#ifndef SMPLFIT_STAMP_B
#define SMPLFIT_STAMP_B 1000  // the instance whose stamps are printed
#endif
  long long t[12] = {};
  __device__ __forceinline__ void stamp(int k) { t[k] = __builtin_readcyclecounter(); }
#endif
};

// the same over each half of the wave: the half totals land in lanes 31 and 63
__device__ __forceinline__ float half_sum_last(float v) {
  v = dpp_step<0x111, 0xf>(v);
  v = dpp_step<0x112, 0xf>(v);
  v = dpp_step<0x114, 0xf>(v);
  v = dpp_step<0x118, 0xf>(v);
  v = dpp_step<0x142, 0xa>(v);
  return v;
}

// Small-model form of the per-instance stages (J <= 32): TWO instances per wave, 32 lanes each.  Those stages are
// VALU-issue bound and most of their loops run over the joints, so a 24-joint model leaves 40 of the 64 lanes idle;
// with two instances per wave the same instructions serve both.  The barrier is the workgroup's (both halves run the
// same uniform control flow), the scratch of half h starts at h * <stage scratch>.
struct DevCtxHalf {
  int lane, n;
  __device__ __forceinline__ void sync() const { __syncthreads(); }
  __device__ __forceinline__ float sum_to_last(float v) const { return half_sum_last(v); }
#ifdef SMPLFIT_STAGE_STAMPS
  long long t[12] = {};
  __device__ __forceinline__ void stamp(int k) { t[k] = __builtin_readcyclecounter(); }
#endif
};
template <int N> struct StageCtx { using type = DevCtx; };
template <> struct StageCtx<32> { using type = DevCtxHalf; };

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
  float* jdT;      // (Mp/64, J*jd_stride padded to 64, 64) the same, instance-innermost (pair-Gram kernel)
  float* pext;     // (B,J,3,S+1)
  float* gramj;    // (B,NE+1)
  double* gramv;   // (B,NE+1)
  float* beta;     // (B,S)
  float* trans;    // (B,3)
  float* jb;       // (B,J,4)
  float* jbT;      // (Mp/64, J*4, 64) the same, instance-innermost (batch-major LBS kernel)
  float* rjoints;  // (B,J,3)
  float* rverts;   // (B,3,Vp) re-evaluated vertices (joints-omitted path only)
  float* tjreg;    // (B,J,3) regressed target joints (joints-omitted path)
  float* rjreg;    // (B,J,3) regressed reference joints
  float* mbj;      // (B,J,3) per-joint residual moments (pair-Gram form)
  float* scale;    // (B) scale_corr of the known-shape fit
  float* regref;   // (B,S) ridge reference of the warm-started fit
  double* cen;     // (B, S*S+S) centred regularised systems of a share_beta fit (general path: one chunk of share_chunk() rows)
  double* cenP;    // (ceil(B/64), S*S+S) partial sums of the rows above (k_share_partial)
  double* censum;  // (S*S+S) their sum (row B of cen; general path: a row of its own)
  double* gvex;    // (B, S+6) general path: extra sums of the scaled solve (target column of k_gen_accum_mfma)
  float* vextra;   // (B,32) extra vertex sums of the scaled solve (scale_extras_vertex)
  float* beta_out; // (B,S) undivided shape of the scaled solve (ws.beta holds the evaluated one)
  float* tjs;      // (B,J,3) target joints times the scale (scale_target refinement)
  // batch-major path: streams with the instance index innermost (lane = instance reads coalesce)
  float* vpT;      // (Mp/64, 3*Vp, 64) v_posed, written by the GEMM
  float* tT;       // (Mp/64, 3*Vp, 64) targets AS GIVEN at their sorted slots (padding slots: the mean); the
                   // consumers subtract ws.mean (k_layout_targets / k_mean_finish)
  float* psumP;    // (rows, 16, Mp) part sums per row of the LBS share table
  float* resP;     // ([share][16] + [segment row][3 kGQ], Mp) residual-pass sums (k_residual_bm); scratch of the layout pass
  float* gramP;    // (workgroups of k_pair_gram_bm, NG, Mp) pair-Gram partial sums
  float* wT;       // (Mp/64, Vp, 64) vertex weights at the sorted slots (padding slots: 0), k_layout_weights
  float* accP;     // (cells, NE+1, Mp) cell records of the weighted accumulate (k_accum_w_bm)
  // general path: the S-sized scratch of the per-instance stages lives here instead of LDS
  float* gT;       // (B,J,3,S+1) T = P - G J_ext of the joint stage (its P is ws.pext)
  float* gsolve;   // (B, gen_solve_scratch_floats(S)) the scratch of stage S / S' (the S x S system)
  // batch-major prologue (k_prologue_bm, round 6)
  float* GT;       // (J*9, Mp) global rotations, instance-innermost (written by k_joint_stage beside ws.G)
  float* pextT;    // (J*3*(S+1), Mp) FK positions with their beta-Jacobian, instance-innermost
  float* gramjP;   // (prologue workgroups per instance block, NE+1, Mp) partial sums of the joint block
};

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Kernels that may ask for more than 64 KB of dynamic LDS: the attribute is set once per (device, kernel) — a list
// under a mutex, so that every device id has its own entry (round 5 kept 16 flags indexed by id & 15: the ids from 16
// up shared a flag with a lower id and never got the attribute)
void ensure_max_lds(const void* fn) {
  static std::mutex mu;
  static std::vector<std::pair<int, const void*>> done;
  int dev_id = 0;
  (void)hipGetDevice(&dev_id);
  std::lock_guard<std::mutex> lock(mu);
  for (const auto& e : done)
    if (e.first == dev_id && e.second == fn) return;
  (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  done.emplace_back(dev_id, fn);
}

// Work units of k_pair_gram_bm (kernels_bm.inc): every joint twice + chunks of kPgPairs joint pairs; a workgroup of
// kPgWaves waves takes kPgWaves units and writes ONE upper triangle to ws.gramP.  Shared by the workspace carve, the
// launch and the combine kernels.
// (3 pairs per chunk: 77 units = 10 workgroups per instance block for the SMPL-shaped model — 640 workgroups at
// B = 4096, one round of the chip at 2.5 resident workgroups per CU; with 2 pairs per chunk (rounds 4 - 5) 832: a second,
// mostly empty round: 27.3 -> 24.7 us, SMPL-X 57 -> 54)
#ifndef SMPLFIT_PG_PAIRS
#define SMPLFIT_PG_PAIRS 3
#endif
constexpr int kPgWaves = 8, kPgPairs = SMPLFIT_PG_PAIRS;
constexpr int pair_gram_units(int J, int npairs) { return 2 * J + (npairs + kPgPairs - 1) / kPgPairs; }
constexpr int pair_gram_workgroups(int J, int npairs) { return (pair_gram_units(J, npairs) + kPgWaves - 1) / kPgWaves; }
constexpr int kProWaves = 8;  // joints (= waves) per workgroup of k_prologue_bm
constexpr int kRefWaves = 8;  // waves per workgroup of k_refine_bm (64 instances; the waves take joints)
constexpr int prologue_splits(int J) { return (J + kProWaves - 1) / kProWaves; }
constexpr int kAccExtrasHost = 16;  // (= kAccExtras of kernels_bm.inc: the extras of the scaled solve behind a cell record)

#ifndef SMPLFIT_SLAB
#define SMPLFIT_SLAB 32  // original vertices per workgroup of k_layout_targets (kSlabV)
#endif

// fwd_only: the slice a batch-major forward of this model needs (the input side of a fused conversion, see
// smplfit_convert_f32): pose features, joint rows, shape / translation and the instance-innermost v_posed buffer.
// general path: instances per chunk of a share_beta solve (the workspace reserves their S^2 + S doubles)
inline int share_chunk(int B) { return std::min((B + 63) / 64 * 64, 256); }

size_t carve(const sf::HostTables& t, int B, char* base, Workspace* w, bool fwd_only = false) {
  const size_t Mp = align_up((size_t)B, 128);
  const size_t Vp = t.Vp, J = t.J, S = t.S, NE1 = t.ne() + 1;
  size_t off = 0;
  auto take = [&](size_t bytes, bool fwd = false) {  // fwd: also part of the forward-only slice
    if (fwd_only && !fwd) return (char*)nullptr;
    size_t o = off;
    off = align_up(off + bytes, 256);
    return base ? base + o : nullptr;
  };
  Workspace ws;
  ws.tvs = (float*)take((size_t)B * 3 * Vp * 4);
  ws.vws = (float*)take((size_t)B * Vp * 4);
  // instance-major GEMM output; on the batch-major path of a model with Kp != 208 it holds the split feature
  // images of the tiled GEMM instead (k_split_features: 32 KB per 256-instance tile and 32-k stage)
  ws.vposed = (float*)take(fwd_only ? (t.Kp != 208 ? (Mp + 255) / 256 * (size_t)((t.Kp + 31) / 32) * ((sf::kGemm3 ? 2 : 3) * 256 * 64) : 0)
                                    : Mp * 3 * Vp * 4, true);
  ws.rp = (float*)take(Mp * t.Kp * 4, true);
  ws.mean = (float*)take((size_t)B * 3 * 4);
  ws.tjc = (float*)take((size_t)B * J * 3 * 4);
  ws.psum = (float*)take((size_t)B * J * sf::kPsum * 4);
  ws.G = (float*)take((size_t)B * J * 9 * 4);
  ws.jd = (float*)take((size_t)B * J * sf::jd_stride(S) * 4, true);
  ws.pext = (float*)take((size_t)B * J * 3 * (S + 1) * 4);
  ws.gramj = (float*)take((size_t)B * NE1 * 4);
  ws.gramv = (double*)take((size_t)B * NE1 * 8);
  ws.beta = (float*)take((size_t)B * S * 4, true);
  ws.trans = (float*)take((size_t)B * 3 * 4, true);
  ws.jb = (float*)take((size_t)B * J * 4 * 4, true);
  ws.jbT = (float*)take(Mp * J * 4 * 4, true);
  ws.rjoints = (float*)take((size_t)B * J * 3 * 4, true);
  ws.rverts = (float*)take((size_t)B * 3 * Vp * 4);
  ws.tjreg = (float*)take((size_t)B * J * 3 * 4);
  ws.rjreg = (float*)take((size_t)B * J * 3 * 4);
  ws.mbj = (float*)take((size_t)B * J * 3 * 4);
  ws.scale = (float*)take((size_t)B * 4);
  ws.regref = (float*)take((size_t)B * S * 4, true);
  // (general path: (S^2 + S) doubles per instance — 0.7 MB at S = 300 — are reserved for ONE chunk of the batch; a
  // share_beta solve writes and sums the instances' systems chunk by chunk, in the order of the unchunked sum)
  ws.cen = (double*)take((t.general ? (size_t)share_chunk(B) : (size_t)B + 1) * (S * S + S) * 8);
  ws.cenP = (double*)take(((size_t)B + 63) / 64 * (S * S + S) * 8);
  ws.censum = t.general ? (double*)take((S * S + S) * 8) : (ws.cen ? ws.cen + (size_t)B * (S * S + S) : nullptr);
  ws.gvex = (double*)take(t.general ? (size_t)B * (S + sf::kScaleExtras) * 8 : 0);
  ws.vextra = (float*)take((size_t)B * 32 * 4);
  ws.beta_out = (float*)take((size_t)B * S * 4);
  ws.tjs = (float*)take((size_t)B * J * 3 * 4);
  // (the batch-major streams: never read on the general path)
  ws.vpT = (float*)take(t.general ? 0 : Mp * 3 * Vp * 4, true);
  ws.tT = (float*)take(t.general ? 0 : Mp * 3 * Vp * 4);
  {
    // rows of partial sums: the coarse tables', and the fine tables' as well for a batch that may take them (whatever
    // SMPLFIT_FINE_B says at the time of the call)
    size_t lbs_rows = 0, res_rows = 0;
    for (size_t k = 0; k < t.shares.size(); ++k) {
      if ((int)k >= sf::kShareFine && B > sf::kFineMaxBatch) break;
      if ((int)k % sf::kShareKinds == sf::kShareResidual)
        res_rows = std::max(res_rows, (size_t)t.shares[k].ncells * ((S + 3 + 3) / 4 * 4) + (size_t)t.shares[k].nrows * 3 * sf::kGroupJoints);
      else lbs_rows = std::max(lbs_rows, (size_t)t.shares[k].nrows);
    }
    // (the layout kernel's slab sums use ws.resP as scratch: 3 rows per slab)
    const size_t nslab = ((size_t)t.V + SMPLFIT_SLAB - 1) / SMPLFIT_SLAB;
    ws.psumP = (float*)take(lbs_rows * 16 * Mp * 4);
    ws.resP = (float*)take(std::max(res_rows, 3 * nslab) * Mp * 4);
    // weighted fits on the batch-major path: the weight stream and the cell records of k_accum_w_bm
    size_t acc_cells = 0;
    for (size_t k = sf::kShareResidual; k < t.shares.size(); k += sf::kShareKinds)
      if ((int)k < sf::kShareFine || B <= sf::kFineMaxBatch) acc_cells = std::max(acc_cells, (size_t)t.shares[k].ncells);
    ws.wT = (float*)take(t.shares.empty() ? 0 : Mp * Vp * 4);
    ws.accP = (float*)take(acc_cells * (NE1 + kAccExtrasHost) * Mp * 4);
  }
  {  // k_pair_gram_bm: one upper triangle (NG rows) per workgroup of its launch (pair_gram_workgroups: the same
     // constants as the launch and the combine kernels)
    ws.gramP = (float*)take((size_t)pair_gram_workgroups((int)J, (int)t.pair_c3.size()) * sf::ne_ng((int)S) * Mp * 4);
  }
  ws.jdT = (float*)take(t.general ? 0 : Mp * align_up((size_t)J * sf::jd_stride(S), 64) * 4, true);
  ws.gT = (float*)take(t.general ? (size_t)B * J * 3 * (S + 1) * 4 : 0, true);
  ws.gsolve = (float*)take(t.general ? (size_t)B * align_up((size_t)sf::gen_solve_scratch_floats((int)S), 4) * 4 : 0);
  const bool pro = !t.general && !t.shares.empty();  // (the models the batch-major path can serve)
  ws.GT = (float*)take(pro ? Mp * J * 9 * 4 : 0);
  ws.pextT = (float*)take(pro ? Mp * J * 3 * (S + 1) * 4 : 0);
  ws.gramjP = (float*)take(pro ? (size_t)prologue_splits((int)J) * NE1 * Mp * 4 : 0);
  if (w) *w = ws;
  return off;
}

// Cache policy of the streams.  The per-iteration streams of a fit (v_posed, the targets: 0.34 GB each at B = 4096)
// are written once and read once per pass, and together they are several times the 256 MB Infinity Cache and the
// 32 MB of L2: left to the default policy, the producer's output lingers as dirty lines that the NEXT kernel's reads
// push out — its write-back then competes with those reads (measured, B = 4096: the residual pass takes 178 us behind
// the GEMM, 145 us on its own) — and the streams evict the tables every wave re-reads.  Marked non-temporal, stream
// reads and writes pass through: residual 178 -> 126 us, LBS 142 -> 120, GEMM 106 -> 97, layout 150 -> 126,
// 2.01 -> 2.4 M fits/s, bit-identical results.  SMPLFIT_NT selects which accesses carry the hint (A/B builds):
// 1 stream loads of the batch-major vertex passes, 2 GEMM output stores, 4 target loads of the layout pass (slower:
// 139 us — off), 8 its stores (and the topology transfer's), 16 the streams of the wave-per-instance kernels (K0's
// sorted copy, the reads of K3 / K5).
#ifndef SMPLFIT_NT
#define SMPLFIT_NT 27
#endif
template <int BIT>
__device__ __forceinline__ float ld_stream(const float* p) {
  if constexpr ((SMPLFIT_NT & BIT) != 0) return __builtin_nontemporal_load(p);
  else return *p;
}
template <int BIT>
__device__ __forceinline__ void st_stream(float* p, float v) {
  if constexpr ((SMPLFIT_NT & BIT) != 0) __builtin_nontemporal_store(v, p);
  else *p = v;
}

// the kernels (same anonymous namespace, same translation unit)
#include "kernels_wave.inc"
#include "kernels_bm.inc"
#include "kernels_gen.inc"

// ------------------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// Tuning switches.  Every SMPLFIT_* environment variable the launch code honours is read ONCE (first use) into
// this struct — no getenv on the launch path; smplfit_reload_options() re-reads them (tests and the A/B tools
// switch paths inside one process).  INTEGRATION.md documents each of them.
// ------------------------------------------------------------------------------------------------
struct Tuning {
  bool bm = true;          // SMPLFIT_BM=0: wave-per-instance vertex kernels everywhere
  bool gemm_f32 = false;   // SMPLFIT_GEMM=f32: fp32-MFMA posedirs GEMM instead of the split-bf16 one
  bool pair_form = false;  // SMPLFIT_SHAPE_FORM=pair: pair-Gram form on the wave-per-instance path
  bool k0_two = true;      // SMPLFIT_K0_TWO=0: K0 stages the whole row (one workgroup per CU)
  int chunks = 0;          // SMPLFIT_CHUNKS=1..4: concurrent batch chunks of one fit call (0: by model, see chunk_plan)
  int gemm_nchunk = 0;     // SMPLFIT_GEMM_NCHUNK: column-tile chunks of the split-bf16 GEMM (0 = automatic)
  int gemm_lds_kb = 0;     // SMPLFIT_GEMM_LDS_KB: LDS request of the fp32 A-stationary GEMM (occupancy experiments)
  bool lbs_all_last = false;  // SMPLFIT_LBS_LAST=all: the last part sums of a fit over every used part (A/B of the adjustable-parts pass)
  int stage_half_b = 2048; // SMPLFIT_STAGE_HALF_B: smallest batch whose per-instance stages run two instances per wave (J <= 32)
  int fine_b = sf::kFineMaxBatch;  // SMPLFIT_FINE_B: largest batch that takes the fine cell tables (0: none; at most sf::kFineMaxBatch)
  bool bm_known_pose = true;  // SMPLFIT_BM_KNOWN_POSE=0: smplfit_shape_solve_ex_f32 on the wave-per-instance kernels (A/B)
  bool bm_forward = true;  // SMPLFIT_BM_FORWARD=0: BodyModel.forward on the wave-per-instance LBS kernel (A/B)
  bool bm_scale = true;    // SMPLFIT_BM_SCALE=0: fit(scale_target / scale_fit) on the wave-per-instance kernels (A/B)
  bool bm_known_shape = true;  // SMPLFIT_BM_KNOWN_SHAPE=0: fit_with_known_shape on the wave-per-instance kernels (A/B)
  bool bm_weighted = true; // SMPLFIT_BM_WEIGHTED=0: fits with vertex weights on the wave-per-instance kernels (A/B)
  int bm_slots = 4096;     // SMPLFIT_BM_SLOTS: resident waves a batch-major vertex pass is dealt for (share-count choice)
  int gen_flush = 0;       // SMPLFIT_GEN_FLUSH: vertices between two fp64 additions of the general accumulate kernel's fp32 sums (0: 2048; a scaled iteration: every blend pass)
  bool gen_mfma = true;    // SMPLFIT_GEN_MFMA=0: the general path's vertex block on the vector ALUs (k_gen_accum) instead of the matrix cores (A/B)
  bool rot_bm = true;      // SMPLFIT_ROT_BM=0: the part rotations as the wave-per-instance k_joint_stage behind a part-sum combine instead of k_rotations_bm (A/B)
  bool refine_bm = true;   // SMPLFIT_REFINE_BM=0: the refinement + epilogue as the wave-per-instance k_refine_epilogue behind a part-sum combine instead of k_refine_bm (A/B)
  bool prologue_bm = true; // SMPLFIT_PROLOGUE_BM=0: the shape prologue inside k_joint_stage + the joint-row transpose instead of k_prologue_bm (A/B)
  bool solve_bm = true;    // SMPLFIT_SOLVE_BM=0: normal-equation combine + wave-per-instance solve as two launches instead of k_solve_bm (A/B)
  int bm_lds_kb = 0;       // SMPLFIT_BM_LDS_KB: LDS request of the two batch-major vertex passes padded to this (54: three
                           // workgroups per CU instead of four, which leaves registers / LDS for another chunk's small kernels)
};
// The current options: an immutable snapshot behind an atomic pointer.  smplfit_reload_options() publishes a new
// snapshot; a launch that is reading the old one on another thread keeps a valid object (snapshots are never freed:
// a few hundred bytes per reload).
std::atomic<const Tuning*> g_tune{nullptr};
std::mutex g_tune_mu;
Tuning read_tuning() {
  Tuning t;
  auto env = [](const char* n) { return getenv(n); };
  if (const char* e = env("SMPLFIT_BM")) t.bm = e[0] != '0';
  if (const char* e = env("SMPLFIT_GEMM")) t.gemm_f32 = e[0] == 'f';
  if (const char* e = env("SMPLFIT_SHAPE_FORM")) t.pair_form = std::string(e) == "pair";
  if (const char* e = env("SMPLFIT_K0_TWO")) t.k0_two = e[0] != '0';
  if (const char* e = env("SMPLFIT_CHUNKS")) t.chunks = std::min(std::max(atoi(e), 1), 4);
  if (const char* e = env("SMPLFIT_GEMM_NCHUNK")) t.gemm_nchunk = std::max(1, atoi(e));
  if (const char* e = env("SMPLFIT_GEMM_LDS_KB")) t.gemm_lds_kb = atoi(e);
  if (const char* e = env("SMPLFIT_LBS_LAST")) t.lbs_all_last = e[0] == 'a';
  if (const char* e = env("SMPLFIT_BM_KNOWN_POSE")) t.bm_known_pose = e[0] != '0';
  if (const char* e = env("SMPLFIT_BM_FORWARD")) t.bm_forward = e[0] != '0';
  if (const char* e = env("SMPLFIT_BM_SCALE")) t.bm_scale = e[0] != '0';
  if (const char* e = env("SMPLFIT_BM_KNOWN_SHAPE")) t.bm_known_shape = e[0] != '0';
  if (const char* e = env("SMPLFIT_BM_WEIGHTED")) t.bm_weighted = e[0] != '0';
  if (const char* e = env("SMPLFIT_FINE_B")) t.fine_b = std::min(std::max(atoi(e), 0), sf::kFineMaxBatch);
  if (const char* e = env("SMPLFIT_STAGE_HALF_B")) t.stage_half_b = std::max(atoi(e), 1);
  if (const char* e = env("SMPLFIT_BM_SLOTS")) t.bm_slots = std::min(std::max(atoi(e), 256), 16384);
  if (const char* e = env("SMPLFIT_GEN_MFMA")) t.gen_mfma = e[0] != '0';
  if (const char* e = env("SMPLFIT_GEN_FLUSH")) t.gen_flush = std::max(atoi(e), 0);
  if (const char* e = env("SMPLFIT_SOLVE_BM")) t.solve_bm = e[0] != '0';
  if (const char* e = env("SMPLFIT_PROLOGUE_BM")) t.prologue_bm = e[0] != '0';
  if (const char* e = env("SMPLFIT_REFINE_BM")) t.refine_bm = e[0] != '0';
  if (const char* e = env("SMPLFIT_ROT_BM")) t.rot_bm = e[0] != '0';
  if (const char* e = env("SMPLFIT_BM_LDS_KB")) t.bm_lds_kb = std::min(std::max(atoi(e), 0), 64);
  return t;
}
void load_tuning() {
  std::lock_guard<std::mutex> lock(g_tune_mu);
  g_tune.store(new Tuning(read_tuning()), std::memory_order_release);
}
const Tuning& tune() {
  const Tuning* t = g_tune.load(std::memory_order_acquire);
  if (!t) {
    std::lock_guard<std::mutex> lock(g_tune_mu);
    t = g_tune.load(std::memory_order_acquire);
    if (!t) {
      t = new Tuning(read_tuning());
      g_tune.store(t, std::memory_order_release);
    }
  }
  return *t;
}

// The unit-weight vertex block has two implementations:
//   direct (default): k_shape_accum accumulates G, r, Sb per vertex (VALU-bound);
//   pair  (SMPLFIT_SHAPE_FORM=pair): k_residual + k_pair_gram — 4x fewer per-vertex FLOPs, parity-tested,
//          but not faster on wave-per-instance kernels (the batch-major path always uses the pair form).
bool use_pair_form() { return tune().pair_form; }

// Batch-major vertex kernels: the default whenever they apply (bm_applies); SMPLFIT_BM=0 selects the
// wave-per-instance kernels everywhere (they also serve every other configuration).
bool use_bm() { return tune().bm; }
// Small vertex subsets stay on the wave-per-instance kernels: the pair-Gram and combine passes cost the
// same per instance whatever V is (measured at V = 1024, B = 16384: 4.15 M fits/s batch-major vs 4.63 M).
bool bm_applies(const DevModel& d) {
  // (normalised skinning weights are checked where the handle is at hand: bm_applies(const smplfit_handle*))
  // Vp > V: the batch-major loops run their out-of-range steps on the first padding slot
  // (below ~1000 vertices the prologue of a wave outweighs its vertex work)
  // (KW == 8 since round 5: pieces of up to eight joints, two waves per SIMD — 5-8 skinning weights per vertex)
  // (16 betas ± the kid unknown since round 5 as well: the same kernels at two waves per SIMD)
  return use_bm() && (d.KW == 4 || d.KW == 8) && sf::bm_shape_count(d.S) && d.bm_tables && d.V >= 1024 && d.Vp > d.V;
}

// The batch-major residual kernel derives sum_v b_v from the per-joint moments: exact only when every vertex's
// skinning weights sum to one (sf::HostTables::wsum_dev) — other models stay on the wave-per-instance kernels.
bool bm_applies(const smplfit_handle* h) { return bm_applies(h->d) && h->t.wsum_dev <= 1e-5f; }

// joint rows of the current rotations, instance-innermost, for k_pair_gram_bm
void launch_jd_transpose(const DevModel& d, const Workspace& ws, int B, hipStream_t st) {
  const int Mp = (int)align_up((size_t)B, 128), Ns = d.J * sf::jd_stride(d.S), Np = (int)align_up((size_t)Ns, 64);
  hipLaunchKernelGGL(k_transpose_targets, dim3(Np / 64, Mp / 64), dim3(256), 0, st, ws.jd, ws.jdT, B, Np, Mp, Ns);
}

// The cell table a batch-major vertex pass of `kind` over B instances runs on, with its multiplier (sf::pick_share_mult).
// the table a launch over B instances walks: the fine one up to SMPLFIT_FINE_B instances (sf_tables.h)
int share_index(int kind, int B) { return kind + (B <= tune().fine_b ? sf::kShareFine : 0); }
ShareView share_view(const smplfit_handle* h, int kind, int B) {
  const int nblocks = (int)align_up((size_t)B, 128) / 64, idx = share_index(kind, B);
  ShareView sv = h->views[idx];
  sv.fine = idx >= sf::kShareFine;
  sv.mult = sf::pick_share_mult(h->t, idx, nblocks, tune().bm_slots);
  return sv;
}
dim3 share_grid(const ShareView& sv, int Mp) { return dim3(Mp / 64, sv.ncells / sv.mult / kBW); }
void launch_psum_combine(const DevModel& d, const ShareView& sv, const Workspace& ws, int B, int Mp, hipStream_t st) {
  if (sv.fine) hipLaunchKernelGGL(k_psum_combine_split<8>, dim3(Mp / 64, d.J), dim3(64 * 8), 0, st, d, sv, ws, B, Mp);
  else hipLaunchKernelGGL(k_psum_combine, dim3((B + 255) / 256, d.J), dim3(256), 0, st, d, sv, ws, B, Mp);
}

// part sums of the centred targets against the template + their combine (the first rotation estimate)
void launch_template_partsum_bm(const smplfit_handle* h, const Workspace& ws, int B, hipStream_t st, bool weighted = false,
                                bool combine = true) {
  const DevModel& d = h->d;
  const int Mp = (int)align_up((size_t)B, 128);
  const ShareView sv = share_view(h, sf::kShareLbsUsed, B);
  if (weighted) hipLaunchKernelGGL(k_template_partsum_bm<true>, share_grid(sv, Mp), dim3(64 * kBW), 0, st, d, sv, ws, B, Mp);
  else hipLaunchKernelGGL(k_template_partsum_bm<false>, share_grid(sv, Mp), dim3(64 * kBW), 0, st, d, sv, ws, B, Mp);
  if (combine) launch_psum_combine(d, sv, ws, B, Mp, st);  // (k_rotations_bm adds the rows itself)
}

// One-pass target layout of the batch-major path (k_layout_targets, k_mean_finish, k_template_partsum_bm):
// ws.tT, ws.mean, ws.tjc and the template part sums ws.psum.  ws.resP serves as the slab-sum scratch.
void launch_layout_bm(const smplfit_handle* h, const float* tv, const float* tj, const Workspace& ws, int B, hipStream_t st,
                      const float* vw = nullptr, bool template_sums = true, bool combine = true) {
  const DevModel& d = h->d;
  if (vw)  // vertex weights: their stream first (the template part sums below read it)
    hipLaunchKernelGGL(k_layout_weights, dim3((d.V + 63) / 64 + 1, (int)align_up((size_t)B, 128) / 64), dim3(256), 0, st, d, vw,
                       ws.wT, B);
  const int Mp = (int)align_up((size_t)B, 128), nslab = (d.V + kSlabV - 1) / kSlabV;
  hipLaunchKernelGGL(k_layout_targets, dim3(nslab, Mp / 64), dim3(256), (size_t)64 * kSlabRow * 4, st, d, tv, ws.tT,
                     ws.resP, B, Mp);
  hipLaunchKernelGGL(k_mean_finish, dim3(Mp / 64), dim3(64 * kMeanWaves), 0, st, d, tj, ws.resP, ws, B, Mp, nslab);
  // (a warm-started fit takes its first part sums against the posed initial model instead)
  if (template_sums) launch_template_partsum_bm(h, ws, B, st, vw != nullptr, combine);
}

// K3' + K3g + K3c of the batch-major path for 10 betas (S = 10) and 10 betas + the kid unknown (S = 11)
template <int S>
void launch_residual_bm_s(const smplfit_handle* h, const Workspace& ws, int B, hipStream_t st, int which = 7) {
  const DevModel& d = h->d;
  const int Mp = (int)align_up((size_t)B, 128);
  const ShareView sv = share_view(h, sf::kShareResidual, B);
  if ((which & 1) && d.KW == 8)
    hipLaunchKernelGGL((k_residual_bm<S, 8>), share_grid(sv, Mp), dim3(64 * kBW),
                       std::max(kResidualLds, (size_t)tune().bm_lds_kb * 1024), st, d, sv, ws, B, Mp);
  else if (which & 1)
    hipLaunchKernelGGL((k_residual_bm<S>), share_grid(sv, Mp), dim3(64 * kBW),
                       std::max(kResidualLds, (size_t)tune().bm_lds_kb * 1024), st, d, sv, ws, B, Mp);
  if (which & 2) {
    hipLaunchKernelGGL((k_pair_gram_bm<S>), dim3(pair_gram_workgroups(d.J, d.jt.np), Mp / 64), dim3(64 * kPgWaves), 0, st, d, ws, B, Mp);
  }
  if ((which & 4) && sv.fine)
    hipLaunchKernelGGL((k_gram_combine_split<S, 16>), dim3(Mp / 64, S + 3 + 3 * d.J + sf::ne_ng(S)), dim3(64 * 16), 0,
                       st, d, sv, ws, B, Mp);
  else if (which & 4)
    hipLaunchKernelGGL((k_gram_combine_bm<S>), dim3((B + 255) / 256, S + 3 + 3 * d.J + sf::ne_ng(S)), dim3(256), 0,
                       st, d, sv, ws, B, Mp);
}
void launch_residual_bm(const smplfit_handle* h, const Workspace& ws, int B, hipStream_t st, int which = 7) {
  switch (h->d.S) {
    case 11: launch_residual_bm_s<11>(h, ws, B, st, which); break;
    case 16: launch_residual_bm_s<16>(h, ws, B, st, which); break;
    case 17: launch_residual_bm_s<17>(h, ws, B, st, which); break;
    default: launch_residual_bm_s<10>(h, ws, B, st, which);
  }
}

// the vertex block of the normal equations accumulated per vertex on the batch-major path (S = 10): cell records +
// their combine.  weighted: the vertex weights enter (else unit weights); extras: also the sums of the scaled solve
// (the last iteration of a scale_target / scale_fit fit)
void launch_accum_w_bm(const smplfit_handle* h, const Workspace& ws, int B, hipStream_t st, bool weighted = true,
                       bool extras = false) {
  static_assert(kAccExtrasHost == kAccExtras, "record layout");
  const DevModel& d = h->d;
  const int Mp = (int)align_up((size_t)B, 128);
  ShareView sv = share_view(h, sf::kShareResidual, B);
  // four workgroups per CU (their LDS): the multiplier for rounds of 1024 workgroups
  sv.mult = sf::pick_share_mult(h->t, share_index(sf::kShareResidual, B), Mp / 64, 1024, 1);
  const dim3 grid(Mp / 64, sv.ncells / sv.mult);
  const size_t lds = accum_w_lds<10>();
  if (!extras)
    hipLaunchKernelGGL((k_accum_w_bm<10, kAccWaves>), grid, dim3(64 * kAccWaves), lds, st, d, sv, ws, B, Mp);
  else if (weighted)
    hipLaunchKernelGGL((k_accum_w_bm<10, 1, true, true>), grid, dim3(64), lds, st, d, sv, ws, B, Mp);
  else
    hipLaunchKernelGGL((k_accum_w_bm<10, 1, false, true>), grid, dim3(64), lds, st, d, sv, ws, B, Mp);
  constexpr int NE1 = sf::ne_size(10) + 1;
  const int nent = NE1 + (extras ? 10 + 6 : 0), unit = weighted ? 0 : 1;
  if (sv.fine) hipLaunchKernelGGL((k_accum_combine<10, 16>), dim3(Mp / 64, nent), dim3(64 * 16), 0, st, d, sv, ws, B, Mp, unit);
  else hipLaunchKernelGGL((k_accum_combine<10, 4>), dim3(Mp / 64, nent), dim3(64 * 4), 0, st, d, sv, ws, B, Mp, unit);
}

// write_v (joints-omitted fits): every slot, the vertices at the solution written over ws.vpT, then the reference
// joints of the next rotation pass regressed from them into ws.rjreg.  adj_only (the last pass of a fit with target
// joints): the part sums feed the dependent refinement alone, which reads them at the adjustable parts
// (bodyfitter.py:1505-1517) — only those parts' slots are visited, the other rows of ws.psum become zero.
template <int S, int KW>
void launch_lbs_bm(const smplfit_handle* h, const Workspace& ws, int B, hipStream_t st, bool write_v = false,
                   bool adj_only = false, bool weighted = false, bool write_all = false, int regress = -1, bool combine = true) {
  const DevModel& d = h->d;
  const int Mp = (int)align_up((size_t)B, 128);
  const size_t lds = (size_t)tune().bm_lds_kb * 1024;
  const ShareView sv = share_view(h, write_v ? sf::kShareLbsAll : adj_only ? sf::kShareLbsAdj : sf::kShareLbsUsed, B);
  if constexpr ((KW == 4 || KW == 8) && sf::bm_shape_count(S)) {  // what bm_applies admits (10 / 16 betas with or without the kid unknown)
    if (write_v) {
      // (write_all: every posed vertex — the alignment sums of a known-shape fit; else the slots the regressor reads)
      const int wa = write_all ? 1 : 0;
      if (weighted)
        hipLaunchKernelGGL((k_lbs_partsum_bm<S, KW, true, false, true>), share_grid(sv, Mp), dim3(64 * kBW), lds, st, d, sv, ws, B, Mp, wa);
      else
        hipLaunchKernelGGL((k_lbs_partsum_bm<S, KW, true>), share_grid(sv, Mp), dim3(64 * kBW), lds, st, d, sv, ws, B, Mp, wa);
      // regress: the regressed reference joints are consumed (joints-omitted fits).  A known-shape fit WITH target
      // joints keeps the posed mesh for its alignment sums only; and a model without a regressor has none to apply
      const bool do_regress = (regress < 0 ? true : regress != 0) && h->t.has_regressor;
      if (do_regress)
        hipLaunchKernelGGL(k_regress_joints_bm<false>, dim3(Mp / 64, d.J), dim3(64), 0, st, d, ws.vpT, nullptr, ws.rjreg, B);
    } else if (weighted) {
      hipLaunchKernelGGL((k_lbs_partsum_bm<S, KW, false, false, true>), share_grid(sv, Mp), dim3(64 * kBW), lds, st, d, sv, ws, B, Mp);
    } else {
      hipLaunchKernelGGL((k_lbs_partsum_bm<S, KW>), share_grid(sv, Mp), dim3(64 * kBW), lds, st, d, sv, ws, B, Mp);
    }
  }
  if (combine) launch_psum_combine(d, sv, ws, B, Mp, st);  // (k_refine_bm adds the rows of the last pass itself)
}

// the forward-only variant of the batch-major LBS pass (posed vertices left in ws.vpT): input side of a fused
// conversion, BodyModel.forward, the mesh of a shape solve
int launch_lbs_fwd_bm(const DevModel& d, const ShareView& sv, const Workspace& ws, int B, int Mp, hipStream_t st) {
  const dim3 grid = share_grid(sv, Mp);
#define SF_FWD(S_, KW_) hipLaunchKernelGGL((k_lbs_partsum_bm<S_, KW_, true, true>), grid, dim3(64 * kBW), 0, st, d, sv, ws, B, Mp)
  if (d.KW == 8) {
    switch (d.S) {
      case 11: SF_FWD(11, 8); break;
      case 16: SF_FWD(16, 8); break;
      case 10: SF_FWD(10, 8); break;
      default: return fail(SMPLFIT_ERR_UNSUPPORTED, "forward LBS pass: no batch-major kernel for this (betas, skinning width)");
    }
  } else {
    switch (d.S) {
      case 11: SF_FWD(11, 4); break;
      case 16: SF_FWD(16, 4); break;
      case 17: SF_FWD(17, 4); break;
      case 10: SF_FWD(10, 4); break;
      default: return fail(SMPLFIT_ERR_UNSUPPORTED, "forward LBS pass: no batch-major kernel for this (betas, skinning width)");
    }
  }
#undef SF_FWD
  return 0;
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
    if (tune().k0_two && d.V > cap && d.V - cap <= d.V / 16) VL = cap;
  }
  const size_t lds_row = ((size_t)((3 * VL + 3) & ~3) + 64) * 4;
  if (lds_row <= 160 * 1024) {
    // dynamic LDS above 64 KB has to be opted into once per kernel (and per device: the attribute is
    // set again whenever the current device changes; idempotent, so racing threads are harmless)
    ensure_max_lds(reinterpret_cast<const void*>(&k_center_sort_partsum_lds<true>));
    ensure_max_lds(reinterpret_cast<const void*>(&k_center_sort_partsum_lds<false>));
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
                     "unsupported (shape unknowns, skinning width) combination"); \
  } while (0)

// GENERAL path (kernels_gen.inc): the vertex block of the normal equations and the LBS / part-sum pass with run-time
// loops over the unknowns and the skinning weights
// the matrix-core form (k_gen_accum_mfma): (weighted, waves, blocks per wave, staged joint rows) -> instantiation
int launch_gen_accum_mfma(const DevModel& d, const Workspace& ws, int B, bool weighted, double* vextra, const float* tj,
                          const float* jw, hipStream_t st) {
  const size_t lds = gen2_lds(d.J, d.S, d.KW);
  if (lds > 160 * 1024) return fail(SMPLFIT_ERR_UNSUPPORTED, "general path: too many shape unknowns for the accumulate kernel's LDS tile");
  const bool stage = gen2_stage_joints(d.J, d.S, d.KW);
  const int nw = gen2_nw(d.S), nbw = gen2_nbw(d.S);
  const dim3 grid(B, gen2_groups(d.S));
  // blend passes (64 / 128 vertices) between two additions of the fp32 accumulators to the fp64 record: a scaled
  // iteration after every pass, the others after 2048 vertices (k_gen_accum_mfma; +2.5 % of its time, 1024: +5 %)
  const int flush_every = vextra ? 1 : std::max(1, (tune().gen_flush ? tune().gen_flush : 2048) / gen2_sv(d.S));
#define SF_GEN2(W_, NW_, NBW_, ST_)                                                                                   \
  do {                                                                                                                \
    ensure_max_lds(reinterpret_cast<const void*>(&k_gen_accum_mfma<W_, NW_, NBW_, ST_>)); \
    hipLaunchKernelGGL((k_gen_accum_mfma<W_, NW_, NBW_, ST_>), grid, dim3(64 * NW_), lds, st, d, ws, B, vextra, flush_every, tj, jw); \
  } while (0)
#define SF_GEN2_W(NW_, NBW_, ST_)               \
  do {                                          \
    if (weighted) SF_GEN2(true, NW_, NBW_, ST_); \
    else SF_GEN2(false, NW_, NBW_, ST_);        \
  } while (0)
#define SF_GEN2_S(NW_, NBW_)              \
  do {                                    \
    if (stage) SF_GEN2_W(NW_, NBW_, true); \
    else SF_GEN2_W(NW_, NBW_, false);     \
  } while (0)
  if (nw == 4) SF_GEN2_S(4, 1);
  else if (nbw == 1) SF_GEN2_S(16, 1);
  else SF_GEN2_S(8, 7);
#undef SF_GEN2_S
#undef SF_GEN2_W
#undef SF_GEN2
  return 0;
}
// (general path, matrix-core accumulate: the target joints enter as rows of the vertex block's kernel instead of the
// joint block of k_joint_stage)
bool gen_joint_rows(const DevModel& d) { return d.general && tune().gen_mfma; }
int launch_gen_accum(const DevModel& d, const Workspace& ws, int B, bool weighted, hipStream_t st, const float* jrows_tj,
                     const float* jrows_jw, bool extras) {
  if (tune().gen_mfma) return launch_gen_accum_mfma(d, ws, B, weighted, extras ? ws.gvex : nullptr, jrows_tj, jrows_jw, st);
  const size_t lds = gen_accum_lds(d.J, d.S, d.KW);
  if (lds > 160 * 1024) return fail(SMPLFIT_ERR_UNSUPPORTED, "general path: too many shape unknowns for the accumulate kernel's LDS tile");
  const bool stage = gen_accum_stage_joints(d.J, d.S, d.KW);
  const int tv = gen_tile_vertices(d.S), nt = gen_accum_threads(d.S);
  // (weighted, vertices per tile, threads, staged joint rows) -> instantiation
#define SF_GEN(W_, TV_, NT_, ST_)                                                                              \
  do {                                                                                                         \
    ensure_max_lds(reinterpret_cast<const void*>(&k_gen_accum<W_, TV_, NT_, ST_>)); \
    hipLaunchKernelGGL((k_gen_accum<W_, TV_, NT_, ST_>), dim3(B), dim3(NT_), lds, st, d, ws, B);               \
  } while (0)
#define SF_GEN_W(TV_, NT_, ST_)              \
  do {                                       \
    if (weighted) SF_GEN(true, TV_, NT_, ST_); \
    else SF_GEN(false, TV_, NT_, ST_);       \
  } while (0)
#define SF_GEN_S(TV_, NT_)            \
  do {                                \
    if (stage) SF_GEN_W(TV_, NT_, true); \
    else SF_GEN_W(TV_, NT_, false);   \
  } while (0)
  if (nt == 64) SF_GEN_S(32, 64);
  else if (tv == 32) SF_GEN_S(32, 256);
  else SF_GEN_S(8, 256);
#undef SF_GEN_S
#undef SF_GEN_W
#undef SF_GEN
  return 0;
}
template <int MODE>
void launch_gen_lbs(const DevModel& d, const Workspace& ws, int B, bool weighted, int nb, const float* beta,
                    const float* trans, float* out, hipStream_t st, const float* kid = nullptr) {
  const size_t lds = gen_lbs_lds(d.J, d.S, B);
  const int ni = gen_lbs_ni(d.S, B);
#define SF_GLBS(M_, W_)                                                                                                   \
  do {                                                                                                                    \
    if (ni == 4) {                                                                                                        \
      ensure_max_lds(reinterpret_cast<const void*>(&k_gen_lbs<M_, W_, 4>));   \
      hipLaunchKernelGGL((k_gen_lbs<M_, W_, 4>), dim3((B + 3) / 4), dim3(256), lds, st, d, ws, B, nb, beta, trans, kid, out); \
    } else {                                                                                                              \
      hipLaunchKernelGGL((k_gen_lbs<M_, W_, 1>), dim3(B), dim3(256), lds, st, d, ws, B, nb, beta, trans, kid, out);       \
    }                                                                                                                     \
  } while (0)
  if (weighted && MODE != 2) SF_GLBS((MODE == 2 ? 0 : MODE), true);
  else SF_GLBS(MODE, false);
#undef SF_GLBS
}
// the vertex block / the LBS pass of the wave-per-instance path OR the general one, by model
// jrows_tj / jrows_jw: the centred target joints (and their weights) when gen_joint_rows() moved the joint block here
// extras: also the extra sums of a scale unknown (general path: ws.gvex; the other paths run k_scale_extras)
int launch_accum_any(const DevModel& d, const Workspace& ws, int B, bool weighted, hipStream_t st,
                     const float* jrows_tj = nullptr, const float* jrows_jw = nullptr, bool extras = false) {
  if (d.general) return launch_gen_accum(d, ws, B, weighted, st, jrows_tj, jrows_jw, extras);
#define SF_CALL_ACCUM(S_, KW_) launch_shape_accum<S_, KW_>(d, ws, B, weighted, st)
  SF_DISPATCH_SKW(d, SF_CALL_ACCUM);
#undef SF_CALL_ACCUM
  return 0;
}
template <int MODE>
int launch_lbs_any(const DevModel& d, const Workspace& ws, int B, bool weighted, int nb, const float* beta,
                   const float* trans, float* out, hipStream_t st, const float* kid = nullptr) {
  if (d.general) {
    launch_gen_lbs<MODE>(d, ws, B, weighted, nb, beta, trans, out, st, kid);
    return 0;
  }
#define SF_CALL_LBS(S_, KW_) launch_lbs<S_, KW_, MODE, false>(d, ws, B, weighted, nb, beta, trans, out, 0.f, 0.f, st, kid)
  SF_DISPATCH_SKW(d, SF_CALL_LBS);
#undef SF_CALL_LBS
  return 0;
}

size_t chunked_workspace_bytes(const sf::HostTables& t, int batch, const sf::HostTables* tin = nullptr);

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

// Arithmetic of the posedirs contraction: "bf16x3" (default) runs it on the bf16 matrix cores with every fp32
// operand split error-free into bf16 terms (3 products per k-step + the bias row's third term, fp32 accumulate:
// fp32-class accuracy at the vertex level, see k_posedirs_gemm_bf16x3); "f32" (SMPLFIT_GEMM=f32) uses the fp32 MFMA,
// which on gfx950 shares the vector ALUs with ordinary VALU work.
bool gemm_bf16x3() { return !tune().gemm_f32; }

int launch_gemm(const DevModel& d, const Workspace& ws, int B, hipStream_t st, bool transposed = false) {
  const int Mp = (int)align_up((size_t)B, 128), N = 3 * d.Vp;
  if (d.Kp == 208 && gemm_bf16x3() && d.gemm_exclusive) {
    // Workgroup = 8 waves x 32 instances = one whole CU (see k_posedirs_gemm_bf16x3).  XCD-aware tiling: block
    // id = y * nchunk + x runs on XCD id % 8, so with nchunk a multiple of 8 a tile chunk x lives on ONE XCD,
    // and with nchunk * ny ~ 512 (two residency rounds) the instance blocks y of a chunk walk its tiles
    // together: the 39 KB tile images come out of that XCD's L2 and posedirs is fetched from HBM about twice
    // per launch instead of once per instance block (FETCH_SIZE: 444 -> 72 MB per launch at B = 4096).
    const int ntiles = N / 32, ny = (Mp + 32 * kGemmWaves - 1) / (32 * kGemmWaves);
    int nchunk = std::max(8, (2 * 256 / ny + 4) / 8 * 8);
    if (tune().gemm_nchunk > 0) nchunk = tune().gemm_nchunk;
    nchunk = std::min(nchunk, ntiles);
    const int per = (ntiles + nchunk - 1) / nchunk;  // trailing chunks may be empty (they return at once)
    const size_t lds = (size_t)kGemmRing * kGemmTileBytes;
    ensure_max_lds(reinterpret_cast<const void*>(&k_posedirs_gemm_bf16x3<true>));
    ensure_max_lds(reinterpret_cast<const void*>(&k_posedirs_gemm_bf16x3<false>));
    if (transposed)
      hipLaunchKernelGGL((k_posedirs_gemm_bf16x3<true>), dim3(nchunk, ny), dim3(64 * kGemmWaves), lds, st, ws.rp,
                         d.pdB, ws.vpT, N, per, Mp);
    else
      hipLaunchKernelGGL((k_posedirs_gemm_bf16x3<false>), dim3(nchunk, ny), dim3(64 * kGemmWaves), lds, st, ws.rp,
                         d.pdB, ws.vposed, N, per, Mp);
    return 0;
  }
  if (d.Kp != 208 && transposed && d.kc32 > 0 && gemm_bf16x3() && d.gemm_exclusive) {
    // tiled split-bf16 GEMM (SMPL-X): the feature images go to ws.vposed, which the batch-major path does not use
    const int mt = (Mp + 255) / 256, nt256 = N / 256;
    uint16_t* aimg = reinterpret_cast<uint16_t*>(ws.vposed);
    hipLaunchKernelGGL(k_split_features, dim3(mt, d.kc32), dim3(256), 0, st, ws.rp, aimg, Mp, d.Kp, d.kc32);
    ensure_max_lds(reinterpret_cast<const void*>(&k_posedirs_gemm_bf16x3_tiled));
    hipLaunchKernelGGL(k_posedirs_gemm_bf16x3_tiled, dim3(8 * ((mt + 7) / 8) * nt256), dim3(512), (size_t)2 * kTg2Stage, st,
                       aimg, d.pdB2, ws.vpT, N, Mp, mt, d.kc32, sf::rp_pos(d.P, d.Kp) / 16);
    return 0;
  }
  if (d.Kp == 208) {  // SMPL (J = 24): A-stationary kernel, 104 A registers per lane
    constexpr int NK2 = 104;
    const int ntiles = N / 32;
    // ~768 workgroups (measured faster than exactly one 512-workgroup residency wave)
    int nchunk = std::max(1, (3 * 256 + Mp / 128 - 1) / (Mp / 128));
    nchunk = std::min(nchunk, ntiles);
    const int per = (ntiles + nchunk - 1) / nchunk;
    nchunk = (ntiles + per - 1) / per;
    size_t lds = (size_t)2 * 32 * (2 * NK2 + 4) * 4;
    {
      // SMPLFIT_GEMM_LDS_KB: pad the workgroup's LDS request (84 = one GEMM workgroup per CU, which leaves
      // LDS and registers for two batch-major workgroups of another chunk beside it)
      const int pad_kb = tune().gemm_lds_kb;
      if (pad_kb > 0) {
        lds = std::max(lds, (size_t)pad_kb * 1024);
        ensure_max_lds(reinterpret_cast<const void*>(&k_posedirs_gemm_as<NK2, true>));
        ensure_max_lds(reinterpret_cast<const void*>(&k_posedirs_gemm_as<NK2, false>));
      }
    }
    if (transposed)
      hipLaunchKernelGGL((k_posedirs_gemm_as<NK2, true>), dim3(nchunk, Mp / 128), dim3(256), lds, st,
                         ws.rp, d.pdSw, ws.vpT, N, per, Mp);
    else
      hipLaunchKernelGGL((k_posedirs_gemm_as<NK2, false>), dim3(nchunk, Mp / 128), dim3(256), lds, st,
                         ws.rp, d.pdSw, ws.vposed, N, per, Mp);
    return 0;
  }
  if (transposed)
    hipLaunchKernelGGL(k_posedirs_gemm<true>, dim3((N / 128) * (Mp / 128)), dim3(256), 0, st, ws.rp, d.pdSw,
                       ws.vpT, Mp, N, d.Kp);
  else
    hipLaunchKernelGGL(k_posedirs_gemm<false>, dim3((N / 128) * (Mp / 128)), dim3(256), 0, st, ws.rp, d.pdSw,
                       ws.vposed, Mp, N, d.Kp);
  return 0;
}

// (general path: the S-sized parts of the scratch — P, T, the S x S system — live in the workspace, see gen_joint_scratch)
size_t joint_lds(const DevModel& d, int kind = 0) {
  return (size_t)sf::joint_scratch_floats(d.J, d.general ? 0 : d.S, kind) * 4;
}
// (general path: the stage's scratch is in global memory, LDS holds the panel of the blocked factorisation)
size_t solve_lds(const DevModel& d) {
  return d.general ? (size_t)(sf::kSolvePanel * d.S + sf::kSolvePanel) * 8 : (size_t)sf::solve_scratch_floats(d.S) * 4;
}

// The per-instance stages run two instances per wave (DevCtxHalf) when the model's joints fit 32 lanes and the batch
// fills the chip either way (below ~2 workgroups per CU the one-instance form is the faster one: B = 256 0.416 vs
// 0.403 M fits/s; B = 4096 2.55 -> 2.59, 32768 2.68 -> 2.70).  A half-wave sum adds lanes 0-31 in the order the
// wave sum does; loops longer than 32 split differently (tests/test_gpu_parity.py::test_stage_half runs both forms).
// SMPLFIT_STAGE_HALF (compile time) masks it per stage: 1 joint stage, 2 solve, 4 refinement.
#ifndef SMPLFIT_STAGE_HALF
#define SMPLFIT_STAGE_HALF 7
#endif
inline bool stage_half(const DevModel& d, int bit, int B) {
  return (SMPLFIT_STAGE_HALF & bit) && d.J <= 32 && B >= tune().stage_half_b && !d.general;
}

void launch_forward_joint(const DevModel& d, const ForwardArgs& fa, const Workspace& ws, int B, hipStream_t st) {
  if (d.general) hipLaunchKernelGGL(k_forward_joint<true>, dim3(B), dim3(64), joint_lds(d), st, d, fa, ws);
  else hipLaunchKernelGGL(k_forward_joint<false>, dim3(B), dim3(64), joint_lds(d), st, d, fa, ws);
}
void launch_joint_stage(const DevModel& d, JointStageArgs ja, const Workspace& ws, int B, hipStream_t st) {
  ja.B = B;
  ja.b0 = 0;
  if (stage_half(d, 1, B)) {
    if (B / 2 > 0) hipLaunchKernelGGL(k_joint_stage<32>, dim3(B / 2), dim3(64), 2 * joint_lds(d), st, d, ja, ws);
    if (B & 1) {  // the odd last instance: a launch of its own (no wave works on one instance twice)
      ja.b0 = B - 1;
      hipLaunchKernelGGL(k_joint_stage<64>, dim3(1), dim3(64), joint_lds(d), st, d, ja, ws);
    }
  } else {
    if (d.general) hipLaunchKernelGGL((k_joint_stage<64, true>), dim3(B), dim3(64), joint_lds(d), st, d, ja, ws);
    else hipLaunchKernelGGL(k_joint_stage<64>, dim3(B), dim3(64), joint_lds(d), st, d, ja, ws);
  }
}
void launch_refine(const DevModel& d, RefineArgs ra, const Workspace& ws, int B, hipStream_t st) {
  ra.B = B;
  ra.b0 = 0;
  if (stage_half(d, 4, B)) {
    if (B / 2 > 0) hipLaunchKernelGGL(k_refine_epilogue<32>, dim3(B / 2), dim3(64), 2 * joint_lds(d, 1), st, d, ra, ws);
    if (B & 1) {
      ra.b0 = B - 1;
      hipLaunchKernelGGL(k_refine_epilogue<64>, dim3(1), dim3(64), joint_lds(d, 1), st, d, ra, ws);
    }
  } else {
    hipLaunchKernelGGL(k_refine_epilogue<64>, dim3(B), dim3(64), joint_lds(d, 1), st, d, ra, ws);
  }
}
// first / count / cen_b0: a sub-range of the batch (general path: the chunks of a share_beta solve, see share_sum); the
// other paths always launch the whole batch
void launch_shape_solve(const DevModel& d, const Workspace& ws, int B, hipStream_t st, float beta_reg, float beta_reg2,
                        float kid_reg, int pair_form, int use_ref, int mode = 0, int first = 0, int count = -1,
                        int cen_b0 = 0) {
  if (count < 0) count = B;
  if (d.general) {
    ensure_max_lds(reinterpret_cast<const void*>(&k_shape_solve<64, true>));
    hipLaunchKernelGGL((k_shape_solve<64, true>), dim3(count), dim3(d.S > 128 ? 1024 : d.S > 64 ? 256 : 64), solve_lds(d), st, d,
                       ws, B, beta_reg, beta_reg2, kid_reg, pair_form, use_ref, mode, first, cen_b0);
  } else if (stage_half(d, 2, B)) {
    if (B / 2 > 0)
      hipLaunchKernelGGL(k_shape_solve<32>, dim3(B / 2), dim3(64), 2 * solve_lds(d), st, d, ws, B, beta_reg,
                         beta_reg2, kid_reg, pair_form, use_ref, mode, 0, 0);
    if (B & 1)
      hipLaunchKernelGGL(k_shape_solve<64>, dim3(1), dim3(64), solve_lds(d), st, d, ws, B, beta_reg, beta_reg2,
                         kid_reg, pair_form, use_ref, mode, B - 1, 0);
  } else {
    hipLaunchKernelGGL(k_shape_solve<64>, dim3(B), dim3(64), solve_lds(d), st, d, ws, B, beta_reg, beta_reg2,
                       kid_reg, pair_form, use_ref, mode, 0, 0);
  }
}

// K4' (k_solve_bm): the normal-equation combine and the shape solve of the batch-major path as one launch, lane =
// instance.  Applies to the plain per-instance solve on the sums of k_residual_bm + k_pair_gram_bm (unit vertex weights
// in the solve, no share_beta, no scale unknown) for 10 / 11 shape unknowns; everything else keeps k_gram_combine_bm +
// k_shape_solve (SMPLFIT_SOLVE_BM=0: everywhere, A/B).
constexpr int kSolveIB = 16;
// what k_solve_bm needs beyond the model: the longest run of moment rows of a joint in the residual table of this batch
// (rounded up to 8), the sizes of its three buffers (their descriptors take byte offsets below 2^31)
struct SolveBmPlan {
  bool ok = false;
  SolveBmArgs a{};
  ShareView sv{};
  size_t lds = 0;
};
SolveBmPlan solve_bm_plan(const smplfit_handle* h, int B) {
  SolveBmPlan p;
  const DevModel& d = h->d;
  if (!tune().solve_bm || d.general || !(d.S == 10 || d.S == 11) || !d.bm_tables) return p;
  const int idx = share_index(sf::kShareResidual, B);
  if (idx >= sf::kShareFine || (size_t)idx >= h->views.size()) return p;  // (small batches: the fine cell tables keep k_gram_combine_split + k_shape_solve)
  const sf::ShareTable& t = h->t.shares[idx];
  const int maxn = h->views[idx].aux_pitch;
  const size_t Mp = align_up((size_t)B, 128);
  const size_t res_rows = (size_t)t.ncells * ((d.S + 3 + 3) / 4 * 4) + (size_t)t.nrows * 3 * sf::kGroupJoints;
  const size_t res_bytes = res_rows * Mp * 4;
  const size_t gram_bytes = (size_t)pair_gram_workgroups(d.J, d.jt.np) * sf::ne_ng(d.S) * Mp * 4;
  const size_t jdt_bytes = Mp * align_up((size_t)d.J * sf::jd_stride(d.S), 64) * 4;
  int lg = 0;
  while ((8 << lg) < t.ncells) ++lg;
  if (maxn > kSolveT3 || (8 << lg) != t.ncells || res_bytes >= (1u << 31) || gram_bytes >= (1u << 31) || jdt_bytes >= (1u << 31) ||
      !h->views[idx].aux_pad)
    return p;
  p.sv = share_view(h, sf::kShareResidual, B);
  const bool stage_px = solve_bm_lds_bytes(d.S, kSolveIB, d.J, t.ncells, maxn, true) <= 156 * 1024;
  p.lds = solve_bm_lds_bytes(d.S, kSolveIB, d.J, t.ncells, maxn, stage_px);
  if (p.lds > 156 * 1024) return p;
  p.a.stage_px = stage_px ? 1 : 0;
  p.a.maxn = maxn;
  p.a.lg_ngrp = lg;
  p.a.res_bytes = (uint32_t)res_bytes;
  p.a.gram_bytes = (uint32_t)gram_bytes;
  p.a.jdt_bytes = (uint32_t)jdt_bytes;
  p.ok = true;
  return p;
}
bool solve_bm_applies(const smplfit_handle* h, int B) { return solve_bm_plan(h, B).ok; }
template <int S>
void launch_solve_bm_s(const smplfit_handle* h, const Workspace& ws, int B, hipStream_t st, SolveBmPlan p) {
  const DevModel& d = h->d;
  const int Mp = (int)align_up((size_t)B, 128);
  const int groups = (int)align_up((size_t)Mp / kSolveIB, 32);
  if (p.lds > 64 * 1024) ensure_max_lds(reinterpret_cast<const void*>(&k_solve_bm<S, kSolveIB>));
  hipLaunchKernelGGL((k_solve_bm<S, kSolveIB>), dim3(groups), dim3(64 * kSolveWaves), p.lds, st, d, p.sv, ws, B, Mp, p.a);
}
// pro: the joint block and the FK rows come from k_prologue_bm (ws.gramjP, ws.pextT) instead of k_joint_stage
void launch_solve_bm(const smplfit_handle* h, const Workspace& ws, int B, hipStream_t st, float beta_reg, float beta_reg2,
                     float kid_reg, int use_ref, bool pro = false) {
  SolveBmPlan p = solve_bm_plan(h, B);  // (callers ask solve_bm_applies first)
  p.a.gj_parts = pro ? prologue_splits(h->d.J) : 0;
  p.a.pext_t = pro ? 1 : 0;
  p.a.beta_reg = beta_reg;
  p.a.beta_reg2 = beta_reg2;
  p.a.kid_reg = kid_reg;
  p.a.use_ref = use_ref;
  if (h->d.S == 11) launch_solve_bm_s<11>(h, ws, B, st, p);
  else launch_solve_bm_s<10>(h, ws, B, st, p);
}

// K1p (k_prologue_bm): the shape prologue of the joint stage on the batch-major path — k_joint_stage then fits the
// rotations only and leaves ws.GT.  Applies when every shape solve of the call is k_solve_bm (the one consumer of its
// ws.gramjP / ws.pextT): the plain per-instance solve, unit vertex weights in the solve, 10 / 11 unknowns.
bool prologue_bm_applies(const smplfit_handle* h, int B) { return tune().prologue_bm && solve_bm_applies(h, B); }
void launch_prologue_bm(const smplfit_handle* h, const JointStageArgs& ja, const Workspace& ws, int B, hipStream_t st) {
  const DevModel& d = h->d;
  const int Mp = (int)align_up((size_t)B, 128);
  PrologueArgs pa{ja.tj, ja.jw, ja.joint_block, ja.joint_block_weighted, ja.vertex_sa_closed_form, B};
  const size_t lds = (size_t)(kProWaves / 2) * (sf::ne_size(d.S) + 1) * 64 * 4;
  const dim3 grid(Mp / 64, prologue_splits(d.J));
  if (d.S == 11) {
    if (lds > 64 * 1024) ensure_max_lds(reinterpret_cast<const void*>(&k_prologue_bm<11>));
    hipLaunchKernelGGL(k_prologue_bm<11>, grid, dim3(64 * kProWaves), lds, st, d, pa, ws, Mp);
  } else {
    if (lds > 64 * 1024) ensure_max_lds(reinterpret_cast<const void*>(&k_prologue_bm<10>));
    hipLaunchKernelGGL(k_prologue_bm<10>, grid, dim3(64 * kProWaves), lds, st, d, pa, ws, Mp);
  }
}
// the joint stage of a fit on the batch-major path: rotations by k_joint_stage, the prologue by k_prologue_bm (pro), or
// both by k_joint_stage
// K1r (k_rotations_bm): the part rotations with lane = instance, adding the part-sum rows of the pass in front of it
// itself.  rot_kind: the cell table of that pass (its rows), or -1: k_joint_stage fits the rotations.  gprev_mode: where
// the previous rotations come from (0 none, 1 ws.GT, 2 the instance-major ja.Gprev).
void launch_rotations_bm(const smplfit_handle* h, const JointStageArgs& ja, int rot_kind, int gprev_mode, const Workspace& ws, int B,
                         hipStream_t st) {
  const DevModel& d = h->d;
  const int Mp = (int)align_up((size_t)B, 128);
  RotArgs ra{ja.tj, ja.rj, ja.jw, ja.Gprev, ja.rj_shared, gprev_mode, B, {}};
  std::memcpy(ra.slot, h->rot_slots, sizeof(ra.slot));
  const size_t lds = (size_t)rot_bm_lds_floats(d.J) * 4;
  if (d.J <= kRotJoints * kRefWaves) {
    if (lds > 64 * 1024) ensure_max_lds(reinterpret_cast<const void*>(&k_rotations_bm));
    hipLaunchKernelGGL(k_rotations_bm, dim3((B + 63) / 64), dim3(64 * kRefWaves), lds, st, d, ra, share_view(h, rot_kind, B), ws, Mp);
  } else {
    if (lds > 64 * 1024) ensure_max_lds(reinterpret_cast<const void*>(&k_rotations_bm_rounds));
    hipLaunchKernelGGL(k_rotations_bm_rounds, dim3((B + 63) / 64), dim3(64 * kRefWaves), lds, st, d, ra, share_view(h, rot_kind, B), ws, Mp);
  }
}
void launch_joint_stage_fit(const smplfit_handle* h, JointStageArgs ja, const Workspace& ws, int B, hipStream_t st, bool pro,
                            int rot_kind = -1, int gprev_mode = 0) {
  if (pro && ja.do_prologue) {
    if (rot_kind >= 0) {
      launch_rotations_bm(h, ja, rot_kind, gprev_mode, ws, B, st);
    } else {
      ja.do_prologue = 0;
      ja.gt_pitch = (int)align_up((size_t)B, 128);
      launch_joint_stage(h->d, ja, ws, B, st);
      ja.do_prologue = 1;
    }
    launch_prologue_bm(h, ja, ws, B, st);
  } else {
    launch_joint_stage(h->d, ja, ws, B, st);
  }
}

// K6' (k_refine_bm): the refinement + epilogue with lane = instance, reading the part-sum rows of the last LBS pass
// itself.  Applies where k_prologue_bm ran (pro: ws.GT holds the rotations of the last joint stage, coarse cell tables)
// on models whose joint arrays of 64 instances fit the LDS (at most 32 joints).
// (groups of adjustable parts by their top-most adjustable ancestor: a wave of k_refine_bm takes a group — the groups are
// dealt to the waves in turn — and keeps the part sums of its at most kRefParts parts in registers)
int refine_bm_groups(const sf::HostTables& t, RefGroups* rg) {  // -> the most parts a wave gets
  const int nadj = t.adj_level_start[t.adj_last_level + 1];
  if (nadj > kRefMaxAdj) return 1 << 20;
  std::vector<int> top(nadj), per_wave(kRefWaves, 0);
  int ntop = 0, mx = 0;
  for (int ai = 0; ai < nadj; ++ai) {
    top[ai] = ai;
    for (int p = t.adj_parts[ai]; p > 0;) {  // up to the root: the group of the nearest adjustable ancestor is the group of the top-most one
      p = t.parents[p];
      for (int a2 = 0; a2 < ai; ++a2)
        if (t.adj_parts[a2] == p && top[ai] == ai) top[ai] = top[a2];
    }
    rg->wave[ai] = (int8_t)(top[ai] == ai ? (ntop++) % kRefWaves : rg->wave[top[ai]]);
    mx = std::max(mx, ++per_wave[rg->wave[ai]]);
  }
  return mx;
}
bool refine_bm_applies(const smplfit_handle* h, bool pro) {
  return pro && tune().refine_bm && h->d.J <= 32 && h->refine_group_max <= kRefParts &&
         (size_t)refine_bm_lds_floats(h->d.J) * 4 <= 160 * 1024;
}
// (once k_rotations_bm runs nothing writes the instance-major ws.G any more: a model whose refinement stays on
// k_refine_epilogue — more than 32 joints — gets it from k_gt_to_g in front of that kernel)
bool rot_bm_applies(const smplfit_handle* h, bool pro) {
  return pro && tune().rot_bm && h->d.J <= kRotMaxJ && h->rot_nslots <= kRotSlots && h->rot_toes_per_wave <= 2 &&
         (size_t)rot_bm_lds_floats(h->d.J) * 4 <= 160 * 1024;
}
// sv: the table of the LBS pass whose rows hold the part sums (unused without final_adjust)
void launch_refine_bm(const smplfit_handle* h, RefineArgs ra, const ShareView& sv, const Workspace& ws, int B, hipStream_t st) {
  const DevModel& d = h->d;
  const int Mp = (int)align_up((size_t)B, 128);
  ra.B = B;
  ra.b0 = 0;
  const size_t lds = (size_t)refine_bm_lds_floats(d.J) * 4;
  if (d.S == 11) {
    if (lds > 64 * 1024) ensure_max_lds(reinterpret_cast<const void*>(&k_refine_bm<11>));
    hipLaunchKernelGGL(k_refine_bm<11>, dim3((B + 63) / 64), dim3(64 * kRefWaves), lds, st, d, ra, sv, ws, ws.rjoints, Mp, h->refine_groups);
  } else {
    if (lds > 64 * 1024) ensure_max_lds(reinterpret_cast<const void*>(&k_refine_bm<10>));
    hipLaunchKernelGGL(k_refine_bm<10>, dim3((B + 63) / 64), dim3(64 * kRefWaves), lds, st, d, ra, sv, ws, ws.rjoints, Mp, h->refine_groups);
  }
}

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
  smplfit_share_allreduce_fn share_allreduce = nullptr;  // completes the sum over the ranks of a sharded batch
  void* share_user = nullptr;
  int scale_mode = 0;                 // 1 scale_target, 2 scale_fit: the last solve has a scale unknown
  float scale_reg = 0.f;
  float* scale_out = nullptr;         // (B) scale_corr
  // fused conversion (smplfit_convert_f32): the targets are produced on the device — forward of the input model on
  // the batch-major kernels, topology transfer straight into this fit's target stream — instead of being read
  // from target_vertices
  const struct ConvertSource* source = nullptr;
};

struct ConvertSource {
  const smplfit_convert_plan* plan;
  const float *pose, *betas, *trans;  // this chunk's rows of the input parameters; betas (B, nb) / trans may be null
  int nb;
  Workspace wsi;  // forward-only workspace slice of the input model (carve(..., fwd_only))
};
int launch_convert_source(const ConvertSource& src, const DevModel& d_out, const Workspace& ws, int B, hipStream_t st);

// The solve of one shape pass on the sums already in the workspace: the plain per-instance solve, the
// scaled solve (one more unknown; extra vertex sums first) or the shared solve (assemble, sum over the
// batch [and the ranks], solve the sum).  The all-shared branch of the reference's lstsq_partial_share
// drops the ridge reference (pt/lstsq.py:45-47): so does this.
// the sum of the instances' systems of a share_beta solve into ws.censum: `assemble(first, count, cen_b0)` launches the
// kernel that writes the systems of the instances [first, first + count) to the rows [first - cen_b0, ...) of ws.cen.
// The partial sums are those of 64 consecutive instances whatever the chunking (general path: the workspace holds
// share_chunk(B) rows), so the sum does not depend on it.
template <class Assemble>
int share_sum(const DevModel& d, const Workspace& ws, int B, const FitOptions& o, hipStream_t st, Assemble assemble) {
  const int NC = d.S * d.S + d.S, ey = (NC + 511) / 512;
  const int chunk = d.general ? share_chunk(B) : B;
  for (int c0 = 0; c0 < B; c0 += chunk) {
    const int cnt = std::min(chunk, B - c0);
    assemble(c0, cnt, d.general ? c0 : 0);
    hipLaunchKernelGGL(k_share_partial, dim3((cnt + 63) / 64, ey), dim3(512), 0, st, ws, cnt, NC, c0 / 64);
  }
  hipLaunchKernelGGL(k_share_reduce, dim3(ey), dim3(512), 0, st, ws, (B + 63) / 64, NC);
  if (o.share_allreduce && o.share_allreduce(o.share_user, ws.censum, NC, (void*)st) != 0)
    return fail(SMPLFIT_ERR_HIP, "the share_allreduce callback failed");
  return 0;
}

// fused: the handle when the sums in the workspace are the PARTIAL sums of k_residual_bm + k_pair_gram_bm (the combine
// was not launched: fused_solve() said so) and k_solve_bm takes them from there; null: the record is in ws.gramv
bool fused_solve(const smplfit_handle* h, int B, const FitOptions& o, int pair_in, bool scaled) {
  return pair_in && !scaled && !o.share_beta && solve_bm_applies(h, B);
}
int enqueue_solve(const DevModel& d, const Workspace& ws, int B, const FitOptions& o, bool joints, bool eff_v,
                  bool eff_j, const float* jw, int pair_in, int use_ref, bool scaled, hipStream_t st,
                  bool extras_done = false, const smplfit_handle* fused = nullptr, bool pro = false) {
  if (d.general && scaled && !tune().gen_mfma)
    return fail(SMPLFIT_ERR_UNSUPPORTED, "general path with SMPLFIT_GEN_MFMA=0: the scale unknown's extra sums come from the "
                                         "matrix-core accumulate kernel only");
  if (scaled) {
    // (extras_done: the batch-major accumulate of this iteration has left the extra sums in ws.vextra; general path:
    // they are entries of the accumulate kernel's rank-k update — its target column — in ws.gvex, joints included)
#define SF_CALL_EXTRAS(S_, KW_)                                                                       \
  hipLaunchKernelGGL((k_scale_extras<S_, KW_>), dim3(B), dim3(64),                                    \
                     (size_t)d.J * sf::jd_stride(S_) * 4, st, d, ws, eff_v ? 1 : 0)
    if (!extras_done && !d.general) SF_DISPATCH_SKW(d, SF_CALL_EXTRAS);
#undef SF_CALL_EXTRAS
    ScaledSolveArgs sa{};
    const bool joint_rows = gen_joint_rows(d) && joints;  // the joints' terms are in the records already
    sa.tj = (joints && !joint_rows) ? ws.tjc : nullptr;
    sa.jw = eff_j ? jw : nullptr;
    sa.joint_block = (joints && !joint_rows) ? 1 : 0;
    sa.mode = o.scale_mode;
    sa.pair_form = pair_in;
    sa.use_ref = use_ref;
    sa.beta_reg = o.beta_reg; sa.beta_reg2 = o.beta_reg2; sa.kid_reg = o.kid_reg; sa.scale_reg = o.scale_reg;
    sa.B = B;
    const size_t lds = d.general ? (size_t)(sf::kSolvePanel * (d.S + 1) + sf::kSolvePanel) * 8 : (size_t)sf::scaled_solve_scratch_floats(d.S) * 4;
    const int threads = d.general ? (d.S > 128 ? 1024 : d.S > 64 ? 256 : 64) : 64;
    auto launch = [&](int first, int count, int cen_b0) {
      sa.b0 = first;
      sa.cen_b0 = cen_b0;
      if (d.general) {
        ensure_max_lds(reinterpret_cast<const void*>(&k_shape_solve_scaled<true>));
        hipLaunchKernelGGL(k_shape_solve_scaled<true>, dim3(count), dim3(threads), lds, st, d, ws, sa);
      } else {
        hipLaunchKernelGGL(k_shape_solve_scaled<false>, dim3(count), dim3(64), lds, st, d, ws, sa);
      }
    };
    if (o.share_beta) {  // shared shape, own scale: reduced systems, their sum, solve (pt/lstsq.py:50-90)
      sa.share = 1;
      if (int rc = share_sum(d, ws, B, o, st, launch)) return rc;
      sa.share = 2;
    }
    launch(0, B, 0);
  } else if (o.share_beta) {  // assemble per instance, sum over the batch, solve the sum + own translation
    auto assemble = [&](int first, int count, int cen_b0) {
      launch_shape_solve(d, ws, B, st, o.beta_reg, o.beta_reg2, o.kid_reg, pair_in, 0, 1, first, count, cen_b0);
    };
    if (int rc = share_sum(d, ws, B, o, st, assemble)) return rc;
    launch_shape_solve(d, ws, B, st, o.beta_reg, o.beta_reg2, o.kid_reg, pair_in, 0, 2);
  } else if (fused) {
    launch_solve_bm(fused, ws, B, st, o.beta_reg, o.beta_reg2, o.kid_reg, use_ref, pro);
  } else {
    launch_shape_solve(d, ws, B, st, o.beta_reg, o.beta_reg2, o.kid_reg, pair_in, use_ref);
  }
  return 0;
}

int run_fit(const smplfit_handle* h, const float* tv, const float* tj, const float* vw,
            const float* jw, int B, const FitOptions& o, float* pose, float* betas, float* trans,
            float* kid, float* orient, float* rel, const Workspace& ws, hipStream_t st, int ph_lo = 0,
            int ph_hi = 1 << 30) {
  // PHASES.  The launches of a fit are numbered in phases — 0: the prologue up to the first rotation pass; 1 + 2 it:
  // the vertex block of iteration `it` up to the normal equations; 2 + 2 it: solve, vertices at the solution, next
  // rotation pass; 1 + 2 num_iter: refinement and epilogue — and a call enqueues the phases [ph_lo, ph_hi) only (the
  // host-side state is rebuilt every time): a chunked fit enqueues its chunks phase by phase, alternating between
  // their streams, instead of one whole chunk after the other (fit_impl).
  const auto on = [&](int ph) { return ph >= ph_lo && ph < ph_hi; };
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
  // vertex weights on the batch-major path: the weight stream, weighted part sums, and — when the weights enter the
  // shape solve — the weighted accumulate (built for 10 unknowns; with the kid unknown such a fit stays on the
  // wave-per-instance kernels)
  // scale_target / scale_fit: the LAST iteration's solve has one more unknown and needs extra vertex sums — that
  // iteration runs the accumulate kernel (with or without weights) in its EXTRAS form
  // (the accumulate kernel k_accum_w_bm — vertex weights in the solve, the scaled iteration — holds four joints per piece)
  const bool bm_base = bm_applies(h) && !o.rotations_only && (!o.scale_mode || (tune().bm_scale && d.S == 10 && d.KW == 4));
  const bool bm = bm_base && (!vw || (tune().bm_weighted && (!eff_v || (d.S == 10 && d.KW == 4))));
  if (o.source && !bm) return fail(SMPLFIT_ERR_UNSUPPORTED, "fused conversion: the batch-major path does not apply");
  // (every solve of this call is k_solve_bm: the prologue runs as k_prologue_bm, which also writes ws.jdT)
  const bool pro = bm && !o.rotations_only && !eff_v && !o.scale_mode && !o.share_beta && prologue_bm_applies(h, B);
  const bool rbm = refine_bm_applies(h, pro);  // the refinement as k_refine_bm (adds the last pass's part-sum rows itself)
  const bool rotbm = rot_bm_applies(h, pro);   // the part rotations as k_rotations_bm (adds the part-sum rows itself)
  // (a warm-started fit evaluates its first part sums against the posed initial model: on the batch-major path with
  // the LBS pass of the iterations — until round 4 with the wave-per-instance kernel over a second, sorted copy)
  if (on(0) && !bm) launch_center_sort(d, tv, tj, vw, ws, B, st);
  if (!on(0)) {
  } else if (bm && o.source) {
    if (int rc = launch_convert_source(*o.source, d, ws, B, st)) return rc;
  } else if (bm) {
    launch_layout_bm(h, tv, tj, ws, B, st, vw, !(o.init_pose || o.init_betas), !rotbm);
  }
  const float* tj_rot = ws.tjc;
  if (!joints) {  // regressed target joints from the centred vertices (bodyfitter.py:1342-1344)
    if (!on(0)) {
    } else if (bm)
      hipLaunchKernelGGL(k_regress_joints_bm<true>, dim3((int)align_up((size_t)B, 128) / 64, d.J), dim3(64), 0, st, d,
                         ws.tT, ws.mean, ws.tjreg, B);
    else
      hipLaunchKernelGGL(k_regress_joints, dim3(B), dim3(64), 0, st, d, ws.tvs, ws.tjreg);
    tj_rot = ws.tjreg;
  }
  JointStageArgs ja{};
  ja.tj = tj_rot;
  ja.jw = jw;
  const bool gjr = gen_joint_rows(d) && joints;
  ja.joint_block = (joints && !gjr) ? 1 : 0;
  ja.joint_block_weighted = eff_j ? 1 : 0;
  ja.vertex_sa_closed_form = (eff_v || d.general) ? 0 : 1;  // (the general accumulate sums SA itself)
  ja.do_prologue = o.rotations_only ? 0 : 1;
  ja.fit_rotations = 1;
  ja.Gprev = nullptr;
  // bodyfitter.py:363-382: the first rotation pass runs against the posed model only when a pose or a
  // shape is given; the ridge references reach EVERY shape solve whenever they are given — also an
  // initial_kid_factor on its own (:413-414, :448-449)
  const bool warm = o.init_pose || o.init_betas;
  const int use_ref = (o.init_betas || o.init_kid) ? 1 : 0;
  if (on(0) && (warm || use_ref))
    hipLaunchKernelGGL(k_fill_shape, dim3((B + 255) / 256), dim3(256), 0, st, ws, B, d.S, d.jt.n_kid,
                       o.init_betas, std::min(o.init_nb, d.S - d.jt.n_kid - d.jt.n_pad), o.init_kid);
  if (warm) {
    ForwardArgs fa{};
    fa.pose = o.init_pose;
    fa.betas = ws.beta;  // (B,S) incl. the kid column
    fa.nb = d.S;
    fa.joints = ws.rjoints;
    fa.orient = ws.G;
    if (on(0) && bm) {
      launch_forward_joint(d, fa, ws, B, st);
      if (int rc = launch_gemm(d, ws, B, st, true)) return rc;
      launch_jd_transpose(d, ws, B, st);
#define SF_CALL_LBS(S_, KW_) launch_lbs_bm<S_, KW_>(h, ws, B, st, !joints, false, vweighted, false, -1, !rotbm)
      SF_DISPATCH_SKW(d, SF_CALL_LBS);
#undef SF_CALL_LBS
    } else if (on(0)) {
      launch_forward_joint(d, fa, ws, B, st);
      launch_gemm(d, ws, B, st);
      if (int rc = launch_lbs_any<1>(d, ws, B, vweighted, d.S, ws.beta, ws.trans, nullptr, st)) return rc;
      if (!joints)
        hipLaunchKernelGGL(k_regress_joints, dim3(B), dim3(64), 0, st, d, ws.rverts, ws.rjreg);
    }
    ja.rj = joints ? ws.rjoints : ws.rjreg;
    ja.rj_shared = 0;
    ja.Gprev = ws.G;  // compose with the initial orientations
  } else if (joints) {
    ja.rj = d.j_template;
    ja.rj_shared = 1;
  } else {  // template joints regressed from the default mesh: same regressor on a (1,3,Vp) source
    if (on(0)) hipLaunchKernelGGL(k_regress_joints, dim3(1), dim3(64), 0, st, d, d.dm, ws.rjreg);
    ja.rj = ws.rjreg;
    ja.rj_shared = 1;
  }
  // (first rotations: the rows of the template pass — or of the warm start's LBS pass —; previous rotations: the
  // instance-major ws.G of the warm start's forward stage, if any)
  if (on(0))
    launch_joint_stage_fit(h, ja, ws, B, st, pro, !rotbm ? -1 : (warm && !joints) ? sf::kShareLbsAll : sf::kShareLbsUsed,
                           ja.Gprev ? 2 : 0);
  if (o.rotations_only) {
    if (on(0)) hipLaunchKernelGGL(k_copy, dim3(256), dim3(256), 0, st, ws.G, orient, (size_t)B * d.J * 9);
    return post_launch_check();
  }
  for (int it = 0; it < o.num_iter; ++it) {
    const bool pa = on(1 + 2 * it), pb = on(2 + 2 * it);
    if (!pa) {
    } else if (bm) {
      // batch-major vertex block: one transposed GEMM feeds the residual pass and, after the solve,
      // the LBS / part-sum pass of this iteration
      const int Mp = (int)align_up((size_t)B, 128);
      launch_gemm(d, ws, B, st, true);
      // joint rows instance-innermost for the three kernels below; AFTER the GEMM: in front of it the
      // chunk's GEMM starts later and the chunks overlap worse (1.37 vs 1.40 M fits/s).  (Round 4, measured and
      // not kept: k_joint_stage writing ws.jdT itself — 64 waves of one XCD completing every 256-byte row with
      // one float each — instead of this 8 us launch: 2.54 -> 2.50 M fits/s, SMPL-X 1.31 -> 1.24: the scattered
      // stores cost the latency-bound stage more than the transpose.)
      if (!pro) launch_jd_transpose(d, ws, B, st);
      if (o.scale_mode && it + 1 == o.num_iter) launch_accum_w_bm(h, ws, B, st, eff_v, true);
      else if (eff_v) launch_accum_w_bm(h, ws, B, st);
      else launch_residual_bm(h, ws, B, st, fused_solve(h, B, o, 1, false) ? 3 : 7);  // (k_solve_bm adds the partial sums itself)
    } else {
      launch_gemm(d, ws, B, st);
      if (int rc = launch_accum_any(d, ws, B, eff_v, st, gjr ? tj_rot : nullptr, gjr && eff_j ? jw : nullptr,
                                    o.scale_mode && it + 1 == o.num_iter))
        return rc;
    }
    // K4 stays its own launch: fused into the prologue of the LBS kernel (template flag SOLVE) its
    // ~40 serial barriers stall all four waves of the workgroup and the kernel ran 230 us longer
    const bool scaled_now = o.scale_mode && it + 1 == o.num_iter;  // only the last solve (:434-455)
    // (the accumulate kernel — weighted fits, and the scaled iteration on the batch-major path — leaves the complete
    // record: the classic form of the solve; the residual pass the pair-Gram form)
    const int pair_in = (!eff_v && !d.general && !(bm && scaled_now) && (bm || use_pair_form())) ? 1 : 0;
    if (pb)
      if (int rc = enqueue_solve(d, ws, B, o, joints, eff_v, eff_j, jw, pair_in, use_ref, scaled_now, st, bm && scaled_now,
                                 bm && fused_solve(h, B, o, pair_in, scaled_now) ? h : nullptr, pro))
        return rc;
    const bool last = it + 1 == o.num_iter;
    if (last && !o.final_adjust) break;  // nothing consumes the re-evaluated mesh
    if (!pb) {
    } else if (bm) {
#define SF_CALL_LBS(S_, KW_) \
  launch_lbs_bm<S_, KW_>(h, ws, B, st, !joints, last && joints && !tune().lbs_all_last, vweighted, false, -1, !(last ? rbm : rotbm))
      SF_DISPATCH_SKW(d, SF_CALL_LBS);
#undef SF_CALL_LBS
    } else if (joints) {
      if (int rc = launch_lbs_any<0>(d, ws, B, vweighted, d.S, ws.beta, ws.trans, nullptr, st)) return rc;
    } else {
      if (int rc = launch_lbs_any<1>(d, ws, B, vweighted, d.S, ws.beta, ws.trans, nullptr, st)) return rc;
      hipLaunchKernelGGL(k_regress_joints, dim3(B), dim3(64), 0, st, d, ws.rverts, ws.rjreg);
    }
    if (last) break;
    ja.rj = joints ? ws.rjoints : ws.rjreg;
    ja.rj_shared = 0;
    ja.Gprev = ws.G;
    if (pb) launch_joint_stage_fit(h, ja, ws, B, st, pro, !rotbm ? -1 : !joints ? sf::kShareLbsAll : sf::kShareLbsUsed, 1);
  }
  if (!on(1 + 2 * o.num_iter)) return post_launch_check();
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
  if (rbm)  // (the table of the last LBS pass: every slot without target joints, else the adjustable / the used parts)
    launch_refine_bm(h, ra, share_view(h, !joints ? sf::kShareLbsAll : tune().lbs_all_last ? sf::kShareLbsUsed : sf::kShareLbsAdj, B), ws, B, st);
  else {
    if (rotbm)  // (k_rotations_bm left the rotations instance-innermost only)
      hipLaunchKernelGGL(k_gt_to_g, dim3((B + 63) / 64, (d.J + 15) / 16), dim3(256), 0, st, ws, d.J, B, (int)align_up((size_t)B, 128));
    launch_refine(d, ra, ws, B, st);
  }
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
  // the batch-major vertex kernels (round 4): target (+ weight) streams, transposed GEMM, part sums with lane =
  // instance (the LAST pass leaves the posed vertices in ws.vpT), alignment sums over the streams
  const bool bm = bm_applies(h) && (!vw || tune().bm_weighted) && tune().bm_known_shape;
  const int Mp = (int)align_up((size_t)B, 128);
  const float* tj_rot = ws.tjc;
  if (bm) {
    const int nslab = (d.V + kSlabV - 1) / kSlabV;
    if (vw) hipLaunchKernelGGL(k_layout_weights, dim3((d.V + 63) / 64 + 1, Mp / 64), dim3(256), 0, st, d, vw, ws.wT, B);
    hipLaunchKernelGGL(k_layout_targets, dim3(nslab, Mp / 64), dim3(256), (size_t)64 * kSlabRow * 4, st, d, tv, ws.tT, ws.resP,
                       B, Mp);
    hipLaunchKernelGGL(k_mean_finish, dim3(Mp / 64), dim3(64 * kMeanWaves), 0, st, d, tj, ws.resP, ws, B, Mp, nslab);
    if (!joints) {
      hipLaunchKernelGGL(k_regress_joints_bm<true>, dim3(Mp / 64, d.J), dim3(64), 0, st, d, ws.tT, ws.mean, ws.tjreg, B);
      tj_rot = ws.tjreg;
    }
  } else {
    launch_center_sort(d, tv, tj, vw, ws, B, st);
    if (!joints) {
      hipLaunchKernelGGL(k_regress_joints, dim3(B), dim3(64), 0, st, d, ws.tvs, ws.tjreg);
      tj_rot = ws.tjreg;
    }
  }
  hipLaunchKernelGGL(k_fill_shape, dim3((B + 255) / 256), dim3(256), 0, st, ws, B, d.S, d.jt.n_kid, betas,
                     std::min(nb, d.S - d.jt.n_kid - d.jt.n_pad), kid);
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
    launch_forward_joint(d, fa, ws, B, st);
    if (bm) {
      if (int rc = launch_gemm(d, ws, B, st, true)) return rc;
      launch_jd_transpose(d, ws, B, st);
      // the posed mesh is kept (in place, ws.vpT) where it is read: regressed joints, and the alignment sums behind the
      // last pass
#define SF_CALL_LBS(S_, KW_) launch_lbs_bm<S_, KW_>(h, ws, B, st, !joints || it == o.num_iter, false, vweighted, it == o.num_iter, joints ? 0 : 1)
      SF_DISPATCH_SKW(d, SF_CALL_LBS);
#undef SF_CALL_LBS
    } else {
      launch_gemm(d, ws, B, st);
      // MODE 1: the posed mesh is kept (regressed joints, alignment sums) next to the part sums
      if (int rc = launch_lbs_any<1>(d, ws, B, vweighted, d.S, ws.beta, ws.trans, nullptr, st)) return rc;
      if (!joints)
        hipLaunchKernelGGL(k_regress_joints, dim3(B), dim3(64), 0, st, d, ws.rverts, ws.rjreg);
    }
    if (it == o.num_iter) break;
    ja.rj = joints ? ws.rjoints : ws.rjreg;
    launch_joint_stage(d, ja, ws, B, st);
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
  if (bm) {
    AlignArgs aa{};
    aa.tj = sa.tj;
    aa.jw = sa.jw;
    aa.with_scale = o.scale_fit;
    aa.regressed = sa.regressed;
    aa.scale_out = sa.scale_out;
    aa.nchunk = B <= sf::kFineMaxBatch ? 256 : 64;
    const dim3 grid(Mp / 64, aa.nchunk);
    if (sa.weighted_v) hipLaunchKernelGGL((k_align_partial_bm<1, true>), grid, dim3(64), 0, st, d, ws, B, Mp, aa.nchunk);
    else hipLaunchKernelGGL((k_align_partial_bm<1, false>), grid, dim3(64), 0, st, d, ws, B, Mp, aa.nchunk);
    aa.mode = 1;
    hipLaunchKernelGGL(k_align_finish, dim3(B), dim3(64), 0, st, d, aa, ws, B, Mp);
    if (o.scale_fit) {
      if (sa.weighted_v) hipLaunchKernelGGL((k_align_partial_bm<2, true>), grid, dim3(64), 0, st, d, ws, B, Mp, aa.nchunk);
      else hipLaunchKernelGGL((k_align_partial_bm<2, false>), grid, dim3(64), 0, st, d, ws, B, Mp, aa.nchunk);
      aa.mode = 2;
      hipLaunchKernelGGL(k_align_finish, dim3(B), dim3(64), 0, st, d, aa, ws, B, Mp);
    }
  } else {
    hipLaunchKernelGGL(k_scale_trans, dim3(B), dim3(256), 0, st, d, sa, ws);
  }
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
  launch_refine(d, ra, ws, B, st);
  return post_launch_check();
}

// The target stream of a fused conversion (smplfit_convert_f32), per chunk:
//   forward of the INPUT model on the batch-major kernels (k_forward_joint, transposed GEMM, joint-row transpose,
//   forward-only LBS pass: the posed vertices stay in the input model's instance-innermost buffer),
//   k_transfer_bm into the OUTPUT model's target stream + slab sums, then the tail of launch_layout_bm.
int launch_convert_source(const ConvertSource& src, const DevModel& d, const Workspace& ws, int B, hipStream_t st) {
  const smplfit_convert_plan& pl = *src.plan;
  const DevModel& di = pl.in->d;
  const Workspace& wi = src.wsi;
  const int Mp = (int)align_up((size_t)B, 128);
  hipLaunchKernelGGL(k_fill_shape, dim3((B + 255) / 256), dim3(256), 0, st, wi, B, di.S, di.jt.n_kid, src.betas,
                     src.betas ? std::min(src.nb, di.S - di.jt.n_kid - di.jt.n_pad) : 0, (const float*)nullptr, src.trans);
  ForwardArgs fa{};
  fa.pose = src.pose;
  fa.betas = wi.beta;  // (B,S) rows, zero beyond the given betas
  fa.nb = di.S;
  fa.joints = wi.rjoints;
  launch_forward_joint(di, fa, wi, B, st);
  if (int rc = launch_gemm(di, wi, B, st, true)) return rc;
  launch_jd_transpose(di, wi, B, st);
  {
    const ShareView sv = share_view(pl.in, sf::kShareLbsAll, B);
    const dim3 grid = share_grid(sv, Mp);
    if (int rc_f = launch_lbs_fwd_bm(di, sv, wi, B, Mp, st)) return rc_f;
  }
  TransferTabs tt{pl.d_oslot, pl.d_start, pl.d_islot, pl.d_w, d.V};
  hipLaunchKernelGGL(k_transfer_bm, dim3(pl.nslab, Mp / 64), dim3(256), 0, st, tt, wi.vpT, di.Vp, ws.tT, d.Vp, ws.resP, Mp);
  hipLaunchKernelGGL(k_mean_finish, dim3(Mp / 64), dim3(64 * kMeanWaves), 0, st, d, (const float*)nullptr, ws.resP, ws, B, Mp,
                     pl.nslab);
  launch_template_partsum_bm(pl.out, ws, B, st);
  return 0;
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

// Chunk plan of one fit call: a large batch may be split into chunks (SMPLFIT_CHUNKS=1..4) that run concurrently on
// the caller's stream and the handle's side streams, so that the small latency-bound kernels of one chunk run beside
// the heavy kernels of the other (the GEMM itself never shares a CU).  Default, measured at B = 4096 in round 4 (the
// streams non-temporal, the vertex passes one balanced round): the SMPL-shaped model 2.40 M fits/s in one chunk,
// 2.37 in two, 2.21 in three; the SMPL-X-shaped one 1.13 M in one, 1.20 in two — its per-instance stages (55 joints)
// are a larger share of the fit: two chunks for models with more than 32 joints, one otherwise.
// Chunk sizes are multiples of 128 (the GEMM's instance tile).
int chunk_count(const sf::HostTables& t) {
  const int c = tune().chunks;
  return std::min(c > 0 ? c : (t.J > 32 ? 2 : 1), kMaxChunks);
}
int chunk_plan_n(int n, int batch, int* sizes) {
  // every chunk at least 896 instances (128 above sf::kFineMaxBatch): the chunks of a call all walk the coarse cell
  // tables, and which tables a call takes depends on its batch alone (sf_tables.h).  Chunks are multiples of 128
  // instances, the last one takes the remainder.
  constexpr int kMinChunk = (sf::kFineMaxBatch / 128 + 1) * 128;
  static_assert(kMinChunk > sf::kFineMaxBatch && kMinChunk % 128 == 0, "chunks stay on the coarse tables");
  while (n > 1 && batch < n * kMinChunk) --n;
  const int per = batch / n / 128 * 128;
  for (int k = 0; k + 1 < n; ++k) sizes[k] = per;
  sizes[n - 1] = batch - (n - 1) * per;
  return n;
}
int chunk_plan(const sf::HostTables& t, int batch, int* sizes) { return chunk_plan_n(chunk_count(t), batch, sizes); }

// `tin`: the input model of a fused conversion — every chunk's slice is followed by that model's forward-only slice.
// Sized for every chunk count a call may pick (the tuning options can be reloaded between the two calls).
size_t chunked_workspace_bytes(const sf::HostTables& t, int batch, const sf::HostTables* tin) {
  size_t need = 0;
  for (int n = 1; n <= kMaxChunks; ++n) {
    int sizes[kMaxChunks];
    const int k = chunk_plan_n(n, batch, sizes);
    size_t total = 0;
    for (int i = 0; i < k; ++i)
      total += carve(t, sizes[i], nullptr, nullptr) + (tin ? carve(*tin, sizes[i], nullptr, nullptr, true) : 0);
    need = std::max(need, total);
  }
  return need;
}

// the input side of a fused conversion, whole batch (fit_impl cuts it into the chunks' ConvertSource)
struct ConvertJob {
  const smplfit_convert_plan* plan;
  const float *pose, *betas, *trans;
  int nb;
};
int fit_impl(const smplfit_handle* h, const smplfit_fit_args* args, const ConvertJob* job);

int fit_impl(const smplfit_handle* h, const smplfit_fit_args* args, const ConvertJob* job) {
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
  int rc = check_common(h, batch, workspace, job ? (size_t)-1 : workspace_bytes);
  if (rc) return rc;
  if (job && workspace_bytes < chunked_workspace_bytes(h->t, batch, &job->plan->in->t))
    return fail(SMPLFIT_ERR_WORKSPACE, "workspace too small (see smplfit_convert_workspace_bytes)");
  if ((!target_vertices && !job) || !pose_rotvecs || !shape_betas || !trans)
    return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_fit_f32: null input/output pointer");
  if (num_iter < 1) return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_fit_f32: num_iter must be >= 1");
  if (initial_kid_factor && !h->t.n_kid)
    return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_fit_warm_f32: initial_kid_factor given to a handle without kid");
  if (initial_shape_betas && (num_initial_betas < 0 || num_initial_betas > h->t.num_betas()))
    return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_fit_warm_f32: num_initial_betas must lie in [0, the model's betas]; slice first");
  const int inb = initial_shape_betas ? num_initial_betas : 0;
  FitOptions o{num_iter, beta_regularizer, beta_regularizer2, kid_regularizer,
               final_adjust_rots ? 1 : 0, 0};
  o.share_beta = args->share_beta ? 1 : 0;
  if (args->scale_mode < 0 || args->scale_mode > 2)
    return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_fit_ex_f32: scale_mode must be 0, 1 (scale_target) or 2 (scale_fit)");
  if (args->scale_mode && !args->scale_corr)
    return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_fit_ex_f32: a scale option needs the scale_corr output");
  if (args->share_allreduce && !o.share_beta)
    return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_fit_ex_f32: share_allreduce without share_beta");
  o.share_allreduce = args->share_allreduce;
  o.share_user = args->share_user;
  o.scale_mode = args->scale_mode;
  o.scale_reg = args->scale_regularizer;
  hipStream_t st = (hipStream_t)hip_stream;
  int sizes[kMaxChunks];
  // share_beta couples all instances in every shape solve: one chunk
  const int nchunk = (h->have_streams && !o.share_beta) ? chunk_plan(h->t, batch, sizes) : 1;
  const int J = h->t.J, V = h->t.V, Sb = h->t.num_betas();
  const int Jin = job ? job->plan->in->t.J : 0;
  auto chunk_bytes = [&](int nb) {
    return carve(h->t, nb, nullptr, nullptr) + (job ? carve(job->plan->in->t, nb, nullptr, nullptr, true) : 0);
  };
  auto run_chunk = [&](int b0, int nb, char* wsbase, hipStream_t cs, int ph_lo, int ph_hi) -> int {
    Workspace ws;
    const size_t own = carve(h->t, nb, wsbase, &ws);
    FitOptions oc = o;
    ConvertSource src{};
    if (job) {
      src.plan = job->plan;
      src.pose = job->pose + (size_t)b0 * Jin * 3;
      src.betas = job->betas ? job->betas + (size_t)b0 * job->nb : nullptr;
      src.trans = job->trans ? job->trans + (size_t)b0 * 3 : nullptr;
      src.nb = job->nb;
      carve(job->plan->in->t, nb, wsbase + own, &src.wsi, true);
      oc.source = &src;
    }
    oc.init_pose = initial_pose_rotvecs ? initial_pose_rotvecs + (size_t)b0 * J * 3 : nullptr;
    oc.init_betas = initial_shape_betas ? initial_shape_betas + (size_t)b0 * inb : nullptr;
    oc.init_nb = inb;
    oc.init_kid = initial_kid_factor ? initial_kid_factor + b0 : nullptr;
    oc.scale_out = args->scale_corr ? args->scale_corr + b0 : nullptr;
    return run_fit(h, target_vertices ? target_vertices + (size_t)b0 * V * 3 : nullptr,
                   target_joints ? target_joints + (size_t)b0 * J * 3 : nullptr,
                   vertex_weights ? vertex_weights + (size_t)b0 * V : nullptr,
                   joint_weights ? joint_weights + (size_t)b0 * J : nullptr, nb, oc,
                   pose_rotvecs + (size_t)b0 * J * 3, shape_betas + (size_t)b0 * Sb,
                   trans + (size_t)b0 * 3, kid_factor ? kid_factor + b0 : nullptr,
                   orientations ? orientations + (size_t)b0 * J * 9 : nullptr,
                   relative_orientations ? relative_orientations + (size_t)b0 * J * 9 : nullptr, ws, cs, ph_lo, ph_hi);
  };
  if (nchunk <= 1) return run_chunk(0, batch, (char*)workspace, st, 0, 1 << 30);
  // fork: every chunk is an independent fit with its own workspace slice; chunk 0 stays on the
  // caller's stream, the others go to the handle's side streams and are joined back by events
  // (stream-ordered with respect to the caller, hipGraph-capturable).  The handle's streams and
  // events are shared state: concurrent fit calls on one handle serialise their ENQUEUE here.
  // The chunks are enqueued PHASE BY PHASE (run_fit), alternating between their streams: enqueued one whole chunk
  // after the other, the second chunk's first kernel waits for the host to get through the ~30 launches of the
  // first.  (Measured and not kept, round 4: a deliberate offset between the chunks — chunk c starting when chunk
  // c - 1 has finished k phases, so that the bandwidth-bound kernels of one meet the latency-bound ones of the other:
  // 2.37 M fits/s without, 2.23 / 2.19 / 2.10 with k = 1 / 2 / 3: the offset is paid again at the join of every call.
  // Nor an enforced anti-phase: a "bandwidth token" of events that orders the vertex sections (layout, GEMM + residual,
  // LBS) of all chunks into one sequence, so that a chunk's per-instance stages run beside another chunk's vertex
  // section: 2.56 -> 2.15 M fits/s with two chunks, 1.75 / 1.51 with three / four (SMPL-X 1.34 -> 1.19): every
  // cross-queue event wait costs ~25 us of idle queue, twelve of them per fit.)
  std::lock_guard<std::mutex> lock(h->mu);
  SF_HIP_TRY(hipEventRecord(h->ev_fork, st));
  const int nphase = 2 + 2 * num_iter;
  int b0s[kMaxChunks], first_error = 0;
  char* wsps[kMaxChunks];
  {
    char* wsp = (char*)workspace;
    int b0 = 0;
    for (int c = 0; c < nchunk; ++c) {
      b0s[c] = b0;
      wsps[c] = wsp;
      wsp += chunk_bytes(sizes[c]);
      b0 += sizes[c];
    }
  }
  std::string first_msg;
  bool forked[kMaxChunks] = {true, false, false, false};
  auto note = [&](int rc2) {
    if (rc2 && !first_error) {
      first_error = rc2;
      first_msg = g_last_error;
    }
  };
  for (int ph = 0; ph < nphase && !first_error; ++ph)
    for (int c = 0; c < nchunk && !first_error; ++c) {
      hipStream_t cs = c == 0 ? st : h->side[c - 1];
      if (ph == 0 && c > 0) {
        // a failed fork is an error like any other: the chunks forked before it are still joined below
        const hipError_t fe = hipStreamWaitEvent(cs, h->ev_fork, 0);
        if (fe != hipSuccess) {
          note(fail(SMPLFIT_ERR_HIP, std::string("hipStreamWaitEvent: ") + hipGetErrorString(fe)));
          break;
        }
        forked[c] = true;
      }
      note(run_chunk(b0s[c], sizes[c], wsps[c], cs, ph, ph + 1));
    }
  // a chunk that was forked is always joined back, also after an error: an unjoined fork would
  // invalidate a stream capture and leave work in flight that the caller's stream does not wait for
  for (int c = 1; c < nchunk; ++c)
    if (forked[c]) {
      (void)hipEventRecord(h->ev_join[c - 1], h->side[c - 1]);
      (void)hipStreamWaitEvent(st, h->ev_join[c - 1], 0);
    }
  if (first_error) return fail(first_error, first_msg);
  return SMPLFIT_OK;
}


}  // namespace

// ================================================================================================
// C-ABI
// ================================================================================================
extern "C" {

const char* smplfit_last_error(void) { return g_last_error.c_str(); }
#ifndef SMPLFIT_BUILD_ID
#define SMPLFIT_BUILD_ID "unknown"
#endif
const char* smplfit_version(void) { return "smplfit-hip 0.4 (gfx950) build " SMPLFIT_BUILD_ID; }
int smplfit_abi_version(void) { return SMPLFIT_ABI_VERSION; }

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
  if (!h->t.general && h->t.S + 3 > 3 * h->t.J) {
    delete h;
    return fail(SMPLFIT_ERR_UNSUPPORTED, "smplfit_create: num_betas too large for this joint count");
  }
  if (flags & SMPLFIT_CREATE_HOST_ONLY) {
    *out = h;
    return SMPLFIT_OK;
  }
  // stage images of the tiled split-bf16 GEMM (94 MB for SMPL-X): only for a model whose fits can take the
  // batch-major path (the only launches that read them; the structural part of bm_applies)
  if ((h->t.KW == 4 || h->t.KW == 8) && sf::bm_shape_count(h->t.S) && h->t.wsum_dev <= 1e-5f)
    sf::build_tiled_gemm_images(h->t);
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
  d.general = t.general ? 1 : 0;
  {
    std::vector<int32_t> sa;
    for (auto& sg : t.segments_all) {
      sa.push_back(sg.start);
      sa.push_back(sg.count);
      sa.push_back(sg.part);
    }
    d.nsegall = (int)t.segments_all.size();
    up(sa, &d.segall);
  }
  up(t.sdg, &d.sdg);
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
  up(t.pdSw, &d.pdSw);
  up(t.pdB, &d.pdB);
  up(t.pdB2, &d.pdB2);
  d.kc32 = t.kc32;
  {
    int regs = 1 << 20;
    for (const void* fn : {reinterpret_cast<const void*>(&k_posedirs_gemm_bf16x3<true>),
                           reinterpret_cast<const void*>(&k_posedirs_gemm_bf16x3<false>),
                           reinterpret_cast<const void*>(&k_posedirs_gemm_bf16x3_tiled)}) {
      hipFuncAttributes fa{};
      regs = std::min(regs, hipFuncGetAttributes(&fa, fn) == hipSuccess ? fa.numRegs : 0);
    }
    h->gemm_vgprs = regs;
    d.gemm_exclusive = regs >= 256 ? 1 : 0;
#ifdef SMPLFIT_GEMM_SHARED_CU  // debug builds of tools/dbg_pg.py: the split GEMM on shared CUs, to study the interaction
    d.gemm_exclusive = 1;
#endif
    // the A-stationary split kernel (Kp == 208) gives the bias row its third term in the LAST k-step, where SMPL's
    // 207 pose features put it; a model with Kp == 208 whose bias row sits elsewhere takes the fp32-MFMA GEMM
    if (sf::kGemm3 && t.Kp == 208 && sf::rp_pos(t.P, t.Kp) / 16 != kGemmKS - 1) d.gemm_exclusive = 0;
    if (!d.gemm_exclusive && !tune().gemm_f32) {  // said once per process: every fit of this handle takes the ~3x slower GEMM
      static std::once_flag warned;
      const int regs_now = regs;
      std::call_once(warned, [regs_now] {
        std::fprintf(stderr, "smplfit: the split-bf16 posedirs GEMM is disabled for this model (kernel registers %d < 256 or a "
                             "bias row outside the last k-step): using the fp32-MFMA GEMM (smplfit_info.gemm_vgprs)\n", regs_now);
      });
    }
  }
  std::vector<uint16_t>().swap(h->t.pdB2);  // the host copy of the stage images is not needed any more
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
    {
      std::vector<int32_t> inv(t.V, 0);
      for (int i = 0; i < t.Vp; ++i)
        if (t.perm[i] >= 0) inv[t.perm[i]] = i;
      up(inv, &d.inv_slot);
    }
    // the share tables of the batch-major vertex kernels
    d.bm_tables = t.shares.empty() ? 0 : 1;
    h->views.assign(t.shares.size(), ShareView{});  // (coarse, fine) x kinds
    for (size_t i = 0; i < t.shares.size(); ++i) {
      const sf::ShareTable& stb = t.shares[i];
      ShareView& sv = h->views[i];
      sv.ncells = stb.ncells;
      sv.nrows = stb.nrows;
      sv.rec = stb.rec;
      sv.mult = 1;
      up(stb.piece_start, &sv.piece_start);
      up(stb.pieces, &sv.pieces);
      std::vector<int32_t> astart(t.J + 1, 0), arows;
      if ((int)(i % sf::kShareKinds) == sf::kShareResidual) {  // per joint: the resP rows holding its moments
        for (int j = 0; j < t.J; ++j) {
          for (int r = 0; r < stb.nrows; ++r)
            for (int q = 0; q < sf::kGroupJoints; ++q)
              if (stb.row_joints[(size_t)r * sf::kGroupJoints + q] == j)
                arows.push_back((int32_t)(stb.ncells * res_share_rec(t.S) + r * kResRowRec + 3 * q));
          astart[j + 1] = (int32_t)arows.size();
        }
      } else {  // per part: its rows of ws.psumP
        for (int j = 0; j < t.J; ++j) {
          for (int r = 0; r < stb.nrows; ++r)
            if (stb.row_part[r] == j) arows.push_back(r);
          astart[j + 1] = (int32_t)arows.size();
        }
      }
      sv.max_aux = 0;
      for (int j = 0; j < t.J; ++j) sv.max_aux = std::max(sv.max_aux, astart[j + 1] - astart[j]);
      up(astart, &sv.aux_start);
      up(arows, &sv.aux_rows);
      sv.aux_pitch = std::max(16, (sv.max_aux + 15) / 16 * 16);
      sv.aux_pad = nullptr;
      if ((int)(i % sf::kShareKinds) == sf::kShareResidual) {
        std::vector<int32_t> pad((size_t)t.J * sv.aux_pitch, -1);
        for (int j = 0; j < t.J; ++j)
          for (int k = astart[j]; k < astart[j + 1]; ++k) pad[(size_t)j * sv.aux_pitch + (k - astart[j])] = arows[k];
        up(pad, &sv.aux_pad);
      }
    }
    up(t.brec, &d.brec);
    up(t.pair_E, &d.pair_E);
    up(t.jn_start, &d.jn_start);
    up(t.jn, &d.jn);
    {  // ancestors of every joint, root first (k_prologue_bm walks them: the level-by-level FK as a sum along the chain)
      std::vector<int32_t> as(t.J + 1, 0), an;
      for (int j = 0; j < t.J; ++j) {
        std::vector<int32_t> ch;
        for (int a2 = j; a2 > 0;) {
          a2 = t.parents[a2];
          ch.push_back(a2);
        }
        an.insert(an.end(), ch.rbegin(), ch.rend());
        as[j + 1] = (int32_t)an.size();
      }
      if (an.empty()) an.push_back(0);
      up(as, &d.anc_start);
      up(an, &d.anc);
    }
    up(t.pair_c2e, &d.pair_c2e);
    up(t.diag_c2e, &d.diag_c2e);
  }
  sf::JointTabs& jt = d.jt;
  jt.J = t.J; jt.S = t.S; jt.num_levels = t.num_levels(); jt.adj_last_level = t.adj_last_level;
  jt.P = t.P; jt.Kp = t.Kp;
  jt.n_kid = t.n_kid;
  jt.n_pad = t.n_pad;
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
  up(t.fk_jp, &jt.fk_jp);
  up(t.bone_lv, &jt.bone_lv);
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
  h->refine_group_max = refine_bm_groups(h->t, &h->refine_groups);
  {
    std::fill(h->rot_slots, h->rot_slots + 64, (int8_t)-1);
    for (int j = 0; j < (int)h->t.toe_src.size() && j < 64; ++j) {
      const int src = h->t.toe_src[j];
      if (src >= 0 && src < 64 && h->rot_slots[src] < 0) h->rot_slots[src] = (int8_t)h->rot_nslots++;
    }
    int per_wave[kRefWaves] = {};  // toes of one wave of k_rotations_bm (joint j belongs to wave j % kRefWaves)
    for (int j = 0; j < (int)h->t.toe_src.size(); ++j)
      if (h->t.toe_src[j] >= 0) h->rot_toes_per_wave = std::max(h->rot_toes_per_wave, ++per_wave[j % kRefWaves]);
  }
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
  info->num_betas = t.num_betas();
  info->has_kid = t.n_kid;
  info->padded_vertices = t.Vp;
  info->num_used_vertices = t.n_used;
  info->skin_width = t.KW;
  info->num_segments = (int)t.segments.size();
  info->num_fk_levels = t.num_levels();
  info->adj_last_level = t.adj_last_level;
  info->has_device = h->has_device ? 1 : 0;
  info->gemm_vgprs = h->gemm_vgprs;
  info->vertex_path = t.general ? SMPLFIT_PATH_GENERAL : bm_applies(h) ? SMPLFIT_PATH_BATCH_MAJOR : SMPLFIT_PATH_WAVE;
  info->share_fallback = t.share_fallback;
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
    case SMPLFIT_TAB_VERTEX_PIECES:
      for (auto& g : t.vpieces) {
        tmp.push_back(g.start);
        tmp.push_back(g.count);
        tmp.push_back(g.part);
        tmp.push_back(g.used);
        tmp.push_back(g.nj);
      }
      src = &tmp;
      break;
    case SMPLFIT_TAB_JOINT_PAIRS: src = &t.pair_j; break;
    case SMPLFIT_TAB_CELL_COUNTS:
      for (auto& st : t.shares) tmp.push_back(st.ncells);
      src = &tmp;
      break;
    default: return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_get_table: unknown table id");
  }
  *n = src->size();
  if (dst) std::memcpy(dst, src->data(), std::min(cap, src->size()) * sizeof(int32_t));
  return SMPLFIT_OK;
}

int smplfit_get_share_table(const smplfit_handle* h, int kind, int what, int32_t* dst, size_t cap, size_t* n) {
  if (!h || !n) return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_get_share_table: null argument");
  const sf::HostTables& t = h->t;
  if (kind < 0 || kind >= (int)t.shares.size() || what < 0 || what > 2)
    return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_get_share_table: unknown kind / table (or a model without batch-major tables)");
  const sf::ShareTable& st = t.shares[kind];
  const std::vector<int32_t>* src =
      what == 0 ? &st.piece_start : what == 1 ? &st.pieces : (kind % sf::kShareKinds == sf::kShareResidual ? &st.row_joints : &st.row_part);
  *n = src->size();
  if (dst) std::memcpy(dst, src->data(), std::min(cap, src->size()) * sizeof(int32_t));
  return SMPLFIT_OK;
}

int smplfit_pick_share_mult(const smplfit_handle* h, int kind, int batch) {
  if (!h || h->t.shares.empty() || kind < 0 || kind >= sf::kShareKinds || batch <= 0) return -1;
  return sf::pick_share_mult(h->t, share_index(kind, batch), (int)align_up((size_t)batch, 128) / 64, tune().bm_slots);
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
  return fit_impl(h, args, nullptr);
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
  if (num_betas_given < 0 || num_betas_given > t.num_betas())
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
  smplfit_forward_args a{};
  a.pose_rotvecs = pose_rotvecs; a.glob_rotmats = glob_rotmats; a.shape_betas = shape_betas;
  a.num_betas_given = num_betas_given; a.trans = trans; a.kid_factor = kid_factor; a.batch = batch;
  a.vertices = vertices; a.joints = joints; a.orientations = orientations;
  a.workspace = workspace; a.workspace_bytes = workspace_bytes; a.hip_stream = hip_stream;
  return smplfit_forward_ex_f32(h, &a);
}

int smplfit_forward_ex_f32(const smplfit_handle* h, const smplfit_forward_args* args) {
  if (!args) return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_forward_ex_f32: null arguments");
  const int batch = args->batch, num_betas_given = args->num_betas_given;
  const float *shape_betas = args->shape_betas, *trans = args->trans, *kid_factor = args->kid_factor;
  float *vertices = args->vertices, *joints = args->joints;
  int rc = check_common(h, batch, args->workspace, args->workspace_bytes);
  if (rc) return rc;
  if ((args->pose_rotvecs != nullptr) + (args->glob_rotmats != nullptr) + (args->rel_rotmats != nullptr) > 1)
    return fail(SMPLFIT_ERR_BAD_ARG, "Only one rotation input may be provided");
  if (!joints) return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_forward_f32: joints output is required");
  const DevModel& d = h->d;
  hipStream_t st = (hipStream_t)args->hip_stream;
  Workspace ws;
  carve(h->t, batch, (char*)args->workspace, &ws);
  ForwardArgs fa{};
  fa.pose = args->pose_rotvecs;
  fa.glob = args->glob_rotmats;
  fa.rel = args->rel_rotmats;
  fa.betas = shape_betas;
  fa.nb = shape_betas ? std::min(num_betas_given, d.S - d.jt.n_kid - d.jt.n_pad) : 0;
  if (kid_factor && !d.jt.n_kid)
    return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_forward_f32: kid_factor given to a handle without kid");
  fa.kid = kid_factor;
  if (shape_betas && num_betas_given > d.S - d.jt.n_kid - d.jt.n_pad)
    return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_forward_f32: more betas than the model holds; slice first");
  fa.trans = trans;
  fa.joints = joints;
  fa.orient = args->orientations;
  launch_forward_joint(d, fa, ws, batch, st);
  if (vertices && bm_applies(h) && tune().bm_forward) {
    // the batch-major kernels (round 4; what the input side of a fused conversion runs): shape / translation rows, the
    // transposed GEMM, the forward-only LBS pass over every slot (posed vertices in place in ws.vpT), and the inverse
    // of the target layout into the caller's (B, V, 3)
    const int Mp = (int)align_up((size_t)batch, 128);
    hipLaunchKernelGGL(k_fill_shape, dim3((batch + 255) / 256), dim3(256), 0, st, ws, batch, d.S, d.jt.n_kid, shape_betas, fa.nb,
                       kid_factor, trans);
    if (int rc2 = launch_gemm(d, ws, batch, st, true)) return rc2;
    launch_jd_transpose(d, ws, batch, st);
    const ShareView sv = share_view(h, sf::kShareLbsAll, batch);
    const dim3 grid = share_grid(sv, Mp);
    if (int rc_f = launch_lbs_fwd_bm(d, sv, ws, batch, Mp, st)) return rc_f;
    hipLaunchKernelGGL(k_unlayout_vertices, dim3((d.V + kSlabV - 1) / kSlabV, Mp / 64), dim3(256), (size_t)64 * kSlabRow * 4, st, d,
                       ws.vpT, vertices, batch);
  } else if (vertices) {
    launch_gemm(d, ws, batch, st);
    if (int rc = launch_lbs_any<2>(d, ws, batch, false, fa.nb, shape_betas, trans, vertices, st, kid_factor)) return rc;
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
  smplfit_shape_solve_args a{};
  a.glob_rotmats = glob_rotmats; a.target_vertices = target_vertices; a.target_joints = target_joints;
  a.vertex_weights = vertex_weights; a.joint_weights = joint_weights; a.batch = batch;
  a.beta_regularizer = beta_regularizer; a.beta_regularizer2 = beta_regularizer2;
  a.kid_regularizer = kid_regularizer; a.add_mean = add_mean;
  a.shape_betas = shape_betas; a.trans = trans; a.kid_factor = kid_factor;
  a.vertices_out = vertices_out; a.joints_out = joints_out;
  a.workspace = workspace; a.workspace_bytes = workspace_bytes; a.hip_stream = hip_stream;
  return smplfit_shape_solve_ex_f32(h, &a);
}

int smplfit_shape_solve_ex_f32(const smplfit_handle* h, const smplfit_shape_solve_args* args) {
  if (!args) return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_shape_solve_ex_f32: null arguments");
  const int batch = args->batch;
  int rc = check_common(h, batch, args->workspace, args->workspace_bytes);
  if (rc) return rc;
  if (!args->glob_rotmats || !args->target_vertices || !args->shape_betas || !args->trans)
    return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_shape_solve_f32: null pointer");
  if (args->scale_mode < 0 || args->scale_mode > 2)
    return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_shape_solve_ex_f32: scale_mode must be 0, 1 (scale_target) or 2 (scale_fit)");
  if (args->scale_mode && !args->scale_corr)
    return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_shape_solve_ex_f32: a scale option needs the scale_corr output");
  if (args->scale_mode && (args->vertices_out || args->joints_out))
    return fail(SMPLFIT_ERR_UNSUPPORTED, "smplfit_shape_solve_ex_f32: no mesh outputs with a scale unknown");
  if (args->share_allreduce && !args->share_beta)
    return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_shape_solve_ex_f32: share_allreduce without share_beta");
  if (args->kid_regularizer_reference && !h->t.n_kid)
    return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_shape_solve_ex_f32: kid_regularizer_reference given to a handle without kid");
  if (args->beta_regularizer_reference && (args->num_reference_betas < 0 || args->num_reference_betas > h->t.num_betas()))
    return fail(SMPLFIT_ERR_BAD_ARG,
                "smplfit_shape_solve_ex_f32: num_reference_betas must lie in [0, the model's betas]; slice first");
  const DevModel& d = h->d;
  const float *vertex_weights = args->vertex_weights, *joint_weights = args->joint_weights;
  hipStream_t st = (hipStream_t)args->hip_stream;
  Workspace ws;
  carve(h->t, batch, (char*)args->workspace, &ws);
  const bool joints = args->target_joints != nullptr;
  const bool eff_v = joints ? (vertex_weights && joint_weights) : (vertex_weights != nullptr);
  const bool eff_j = joints && vertex_weights && joint_weights;
  FitOptions o{1, args->beta_regularizer, args->beta_regularizer2, args->kid_regularizer, 0, 0};
  o.share_beta = args->share_beta ? 1 : 0;
  o.share_allreduce = args->share_allreduce;
  o.share_user = args->share_user;
  o.scale_mode = args->scale_mode;
  o.scale_reg = args->scale_regularizer;
  const int use_ref = (args->beta_regularizer_reference || args->kid_regularizer_reference) ? 1 : 0;
  if (use_ref)  // the ridge pulls towards these (pt/bodyfitter.py:1224-1255); missing columns are 0
    hipLaunchKernelGGL(k_fill_shape, dim3((batch + 255) / 256), dim3(256), 0, st, ws, batch, d.S, d.jt.n_kid,
                       args->beta_regularizer_reference,
                       args->beta_regularizer_reference ? std::min(args->num_reference_betas, d.S - d.jt.n_kid - d.jt.n_pad) : 0,
                       args->kid_regularizer_reference);
  // the batch-major vertex kernels (round 4): streams, transposed GEMM, residual pass + pair-Gram (unit weights) or the
  // accumulate kernel (vertex weights in the solve / a scale unknown), as one iteration of fit()
  const bool scaled = o.scale_mode != 0;
  const bool bm = bm_applies(h) && tune().bm_known_pose &&
                  (!vertex_weights || (tune().bm_weighted && (!eff_v || (d.S == 10 && d.KW == 4)))) &&
                  (!scaled || (tune().bm_scale && d.S == 10 && d.KW == 4));
  const int Mp = (int)align_up((size_t)batch, 128);
  if (bm) {
    const int nslab = (d.V + kSlabV - 1) / kSlabV;
    if (eff_v)
      hipLaunchKernelGGL(k_layout_weights, dim3((d.V + 63) / 64 + 1, Mp / 64), dim3(256), 0, st, d, vertex_weights, ws.wT, batch);
    hipLaunchKernelGGL(k_layout_targets, dim3(nslab, Mp / 64), dim3(256), (size_t)64 * kSlabRow * 4, st, d, args->target_vertices,
                       ws.tT, ws.resP, batch, Mp);
    hipLaunchKernelGGL(k_mean_finish, dim3(Mp / 64), dim3(64 * kMeanWaves), 0, st, d, args->target_joints, ws.resP, ws, batch, Mp,
                       nslab);
  } else {
    launch_center_sort(d, args->target_vertices, args->target_joints, vertex_weights, ws, batch, st);
  }
  JointStageArgs ja{};
  ja.tj = joints ? ws.tjc : ws.tjreg;  // unused without the joint block
  ja.rj = nullptr;
  ja.rj_shared = 1;
  ja.Gprev = args->glob_rotmats;
  ja.jw = joint_weights;
  ja.fit_rotations = 0;
  ja.do_prologue = 1;
  const bool gjr = gen_joint_rows(d) && joints;
  ja.joint_block = (joints && !gjr) ? 1 : 0;
  ja.joint_block_weighted = eff_j ? 1 : 0;
  ja.vertex_sa_closed_form = (eff_v || d.general) ? 0 : 1;  // (the general accumulate sums SA itself)
  if (!joints) hipMemsetAsync(ws.tjreg, 0, (size_t)batch * d.J * 3 * 4, st);
  launch_joint_stage(d, ja, ws, batch, st);
  if (bm) {
    if (int rc2 = launch_gemm(d, ws, batch, st, true)) return rc2;
    launch_jd_transpose(d, ws, batch, st);
    if (scaled) launch_accum_w_bm(h, ws, batch, st, eff_v, true);
    else if (eff_v) launch_accum_w_bm(h, ws, batch, st);
    else launch_residual_bm(h, ws, batch, st, fused_solve(h, batch, o, 1, false) ? 3 : 7);
    const int pair_kp = (!eff_v && !scaled) ? 1 : 0;
    rc = enqueue_solve(d, ws, batch, o, joints, eff_v, eff_j, joint_weights, pair_kp, use_ref, scaled, st, scaled,
                       fused_solve(h, batch, o, pair_kp, scaled) ? h : nullptr);
  } else {
    launch_gemm(d, ws, batch, st);
    if (int rc = launch_accum_any(d, ws, batch, eff_v, st, gjr ? ja.tj : nullptr, gjr && eff_j ? joint_weights : nullptr, scaled))
      return rc;
    rc = enqueue_solve(d, ws, batch, o, joints, eff_v, eff_j, joint_weights, (!eff_v && !d.general && use_pair_form()) ? 1 : 0,
                       use_ref, scaled, st);
  }
  if (rc) return rc;
  // a scaled solve leaves the shape as the reference returns it (undivided, :1277-1283) in beta_out
  hipLaunchKernelGGL(k_emit_solution, dim3((batch + 255) / 256), dim3(256), 0, st, ws,
                     o.scale_mode ? ws.beta_out : ws.beta, batch, d.S, d.jt.n_kid, d.S - d.jt.n_kid - d.jt.n_pad, args->add_mean,
                     args->shape_betas, args->trans, args->kid_factor);
  if (o.scale_mode)
    hipLaunchKernelGGL(k_copy, dim3(16), dim3(256), 0, st, ws.scale, args->scale_corr, (size_t)batch);
  if (args->joints_out)
    hipLaunchKernelGGL(k_copy, dim3(64), dim3(256), 0, st, ws.rjoints, args->joints_out,
                       (size_t)batch * d.J * 3);
  if (args->vertices_out && bm) {  // the mesh at the solution: forward-only LBS pass + the inverse of the target layout
    const ShareView sv = share_view(h, sf::kShareLbsAll, batch);
    const dim3 grid = share_grid(sv, Mp);
    if (int rc_f = launch_lbs_fwd_bm(d, sv, ws, batch, Mp, st)) return rc_f;
    hipLaunchKernelGGL(k_unlayout_vertices, dim3((d.V + kSlabV - 1) / kSlabV, Mp / 64), dim3(256), (size_t)64 * kSlabRow * 4, st, d,
                       ws.vpT, args->vertices_out, batch);
  } else if (args->vertices_out) {
    float* vertices_out = args->vertices_out;
    if (int rc = launch_lbs_any<2>(d, ws, batch, false, d.S, ws.beta, ws.trans, vertices_out, st)) return rc;
  }
  return post_launch_check();
}

// ---- topology transfer + fused conversion ---------------------------------------------------------
int smplfit_transfer_create(int32_t num_vertices_in, int32_t num_vertices_out, const int32_t* indptr,
                            const int32_t* indices, const float* values, int flags, smplfit_transfer** out) {
  if (!out || !indptr || num_vertices_in <= 0 || num_vertices_out <= 0)
    return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_transfer_create: null argument / empty matrix");
  *out = nullptr;
  if (indptr[0] != 0) return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_transfer_create: indptr[0] must be 0");
  for (int r = 0; r < num_vertices_out; ++r)
    if (indptr[r + 1] < indptr[r]) return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_transfer_create: indptr must not decrease");
  const int nnz = indptr[num_vertices_out];
  if (nnz > 0 && (!indices || !values)) return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_transfer_create: null indices / values");
  for (int e = 0; e < nnz; ++e)
    if (indices[e] < 0 || indices[e] >= num_vertices_in)
      return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_transfer_create: column index outside the input vertices");
  auto* t = new smplfit_transfer();
  t->v_in = num_vertices_in;
  t->v_out = num_vertices_out;
  t->indptr.assign(indptr, indptr + num_vertices_out + 1);
  t->indices.assign(indices, indices + nnz);
  t->values.assign(values, values + nnz);
  if (!(flags & SMPLFIT_CREATE_HOST_ONLY)) {
    auto up = [&](const void* src, size_t bytes, void** dst) {
      if (hipMalloc(dst, std::max<size_t>(bytes, 16)) != hipSuccess) return false;
      return bytes == 0 || hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess;
    };
    if (!up(t->indptr.data(), t->indptr.size() * 4, (void**)&t->d_indptr) ||
        !up(t->indices.data(), t->indices.size() * 4, (void**)&t->d_indices) ||
        !up(t->values.data(), t->values.size() * 4, (void**)&t->d_values)) {
      smplfit_transfer_destroy(t);
      return fail(SMPLFIT_ERR_HIP, "smplfit_transfer_create: device upload failed");
    }
  }
  *out = t;
  return SMPLFIT_OK;
}

void smplfit_transfer_destroy(smplfit_transfer* t) {
  if (!t) return;
  if (t->d_indptr) (void)hipFree(t->d_indptr);
  if (t->d_indices) (void)hipFree(t->d_indices);
  if (t->d_values) (void)hipFree(t->d_values);
  delete t;
}

int smplfit_transfer_f32(const smplfit_transfer* t, const float* in_vertices, int batch, float* out_vertices,
                         void* hip_stream) {
  if (!t || !in_vertices || !out_vertices || batch < 0)
    return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_transfer_f32: null pointer / negative batch");
  if (!t->d_indptr) return fail(SMPLFIT_ERR_HIP, "smplfit_transfer_f32: the matrix was created host-only (no device)");
  if (batch == 0) return SMPLFIT_OK;
  hipStream_t st = (hipStream_t)hip_stream;
  const size_t lds = (size_t)t->v_in * 12;
  if (lds <= 160 * 1024) {
    ensure_max_lds(reinterpret_cast<const void*>(&k_transfer_rows<true>));
    hipLaunchKernelGGL(k_transfer_rows<true>, dim3(batch), dim3(1024), lds, st, in_vertices, out_vertices, t->d_indptr,
                       t->d_indices, t->d_values, t->v_in, t->v_out);
  } else {
    hipLaunchKernelGGL(k_transfer_rows<false>, dim3(batch), dim3(1024), 0, st, in_vertices, out_vertices, t->d_indptr,
                       t->d_indices, t->d_values, t->v_in, t->v_out);
  }
  return post_launch_check();
}

int smplfit_convert_plan_create(const smplfit_handle* in, const smplfit_handle* out, const smplfit_transfer* transfer,
                                smplfit_convert_plan** plan) {
  if (!in || !out || !plan) return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_convert_plan_create: null argument");
  *plan = nullptr;
  if (!in->has_device || !out->has_device)
    return fail(SMPLFIT_ERR_HIP, "smplfit_convert_plan_create: both handles need a device");
  if (transfer ? (transfer->v_in != in->t.V || transfer->v_out != out->t.V) : (in->t.V != out->t.V))
    return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_convert_plan_create: vertex counts of the models and the matrix disagree");
  if (!bm_applies(in) || !bm_applies(out) || !out->t.has_regressor)
    return fail(SMPLFIT_ERR_UNSUPPORTED,
                "smplfit_convert_plan_create: the fused conversion needs the batch-major kernels on both models "
                "(<= 4 skinning weights per vertex, 10 betas, >= 1024 vertices) and the output model's joint "
                "regressor; use forward + smplfit_transfer_f32 + fit");
  const int Vo = out->t.V;
  std::vector<int32_t> inv_in(in->t.V, 0), oslot(Vo, 0), start(Vo + 1, 0), islot;
  std::vector<float> w;
  for (int i = 0; i < in->t.Vp; ++i)
    if (in->t.perm[i] >= 0) inv_in[in->t.perm[i]] = i;
  for (int i = 0; i < out->t.Vp; ++i)
    if (out->t.perm[i] >= 0) oslot[out->t.perm[i]] = i;
  for (int r = 0; r < Vo; ++r) {
    if (transfer) {
      for (int e = transfer->indptr[r]; e < transfer->indptr[r + 1]; ++e) {
        islot.push_back(inv_in[transfer->indices[e]]);
        w.push_back(transfer->values[e]);
      }
    } else {  // same topology: the identity
      islot.push_back(inv_in[r]);
      w.push_back(1.f);
    }
    start[r + 1] = (int32_t)islot.size();
  }
  auto* p = new smplfit_convert_plan();
  p->in = in;
  p->out = out;
  p->nslab = (Vo + kSlabV - 1) / kSlabV;
  auto up = [&](const void* src, size_t bytes, void** dst) {
    if (hipMalloc(dst, std::max<size_t>(bytes, 16)) != hipSuccess) return false;
    return bytes == 0 || hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess;
  };
  if (!up(oslot.data(), oslot.size() * 4, (void**)&p->d_oslot) || !up(start.data(), start.size() * 4, (void**)&p->d_start) ||
      !up(islot.data(), islot.size() * 4, (void**)&p->d_islot) || !up(w.data(), w.size() * 4, (void**)&p->d_w)) {
    smplfit_convert_plan_destroy(p);
    return fail(SMPLFIT_ERR_HIP, "smplfit_convert_plan_create: device upload failed");
  }
  *plan = p;
  return SMPLFIT_OK;
}

void smplfit_convert_plan_destroy(smplfit_convert_plan* p) {
  if (!p) return;
  for (void* q : {(void*)p->d_oslot, (void*)p->d_start, (void*)p->d_islot, (void*)p->d_w})
    if (q) (void)hipFree(q);
  delete p;
}

size_t smplfit_convert_workspace_bytes(const smplfit_convert_plan* p, int batch) {
  if (!p || batch <= 0) return 0;
  return chunked_workspace_bytes(p->out->t, batch, &p->in->t);
}

int smplfit_convert_f32(const smplfit_convert_plan* p, const smplfit_convert_args* a) {
  if (!p || !a) return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_convert_f32: null argument");
  if (!a->pose_rotvecs) return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_convert_f32: pose_rotvecs is required");
  if (a->shape_betas && (a->num_betas_given < 0 || a->num_betas_given > p->in->t.num_betas()))
    return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_convert_f32: more betas than the input model holds; slice first");
  // the plan was made while the batch-major path applied; a later smplfit_reload_options may have switched it off
  if (!bm_applies(p->in) || !bm_applies(p->out))
    return fail(SMPLFIT_ERR_UNSUPPORTED, "smplfit_convert_f32: the batch-major path is switched off");
  smplfit_fit_args f{};
  f.batch = a->batch;
  f.num_iter = a->num_iter;
  f.beta_regularizer = a->beta_regularizer;
  f.beta_regularizer2 = a->beta_regularizer2;
  f.kid_regularizer = a->kid_regularizer;
  f.final_adjust_rots = a->final_adjust_rots;
  f.pose_rotvecs = a->out_pose_rotvecs;
  f.shape_betas = a->out_shape_betas;
  f.trans = a->out_trans;
  f.kid_factor = a->out_kid_factor;
  f.orientations = a->out_orientations;
  f.relative_orientations = a->out_relative_orientations;
  f.workspace = a->workspace;
  f.workspace_bytes = a->workspace_bytes;
  f.hip_stream = a->hip_stream;
  ConvertJob job{p, a->pose_rotvecs, a->shape_betas, a->trans, a->shape_betas ? a->num_betas_given : 0};
  return fit_impl(p->out, &f, &job);
}

int smplfit_reload_options(void) {
  (void)tune();  // make sure the first-use load has happened, then replace it
  load_tuning();
  return SMPLFIT_OK;
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
  const bool bm = bm_applies(h);  // time the kernels the default fit runs
  const int Mp = (int)align_up((size_t)batch, 128);
  auto once = [&]() -> int {
    switch (kernel_id) {
      case SMPLFIT_KERNEL_POSEDIRS_GEMM: return launch_gemm(d, ws, batch, st, bm);
      case SMPLFIT_KERNEL_PAIR_GRAM:
        if (!bm) return fail(SMPLFIT_ERR_UNSUPPORTED, "pair-Gram kernel: batch-major path not active");
        launch_residual_bm(h, ws, batch, st, 2);
        return 0;
      case SMPLFIT_KERNEL_TRANSPOSE:
        if (!bm) return fail(SMPLFIT_ERR_UNSUPPORTED, "layout kernel: batch-major path not active");
        // the hook has no target pointer: ws.tvs (unused on this path, >= B*V*3 floats) stands in for the rows
        hipLaunchKernelGGL(k_layout_targets, dim3((d.V + kSlabV - 1) / kSlabV, Mp / 64), dim3(256),
                           (size_t)64 * kSlabRow * 4, st, d, ws.tvs, ws.tT, ws.resP, batch, Mp);
        return 0;
      case SMPLFIT_KERNEL_TEMPLATE_PARTSUM:
        if (!bm) return fail(SMPLFIT_ERR_UNSUPPORTED, "template part sums: batch-major path not active");
        {
          const ShareView sv = share_view(h, sf::kShareLbsUsed, batch);
          hipLaunchKernelGGL(k_template_partsum_bm<false>, share_grid(sv, Mp), dim3(64 * kBW), 0, st, d, sv, ws, batch, Mp);
        }
        return 0;
      case SMPLFIT_KERNEL_SHAPE_ACCUM: {
        if (bm) {
          launch_residual_bm(h, ws, batch, st, 1);
          return 0;
        }
        if (int rc = launch_accum_any(d, ws, batch, false, st, gen_joint_rows(d) ? ws.tjc : nullptr)) return rc;
        return 0;
      }
      case SMPLFIT_KERNEL_SHAPE_SOLVE:
        // (the default fit's solve: k_solve_bm on the partial sums the last fit left, else the wave-per-instance stage)
        if (bm && solve_bm_applies(h, batch)) launch_solve_bm(h, ws, batch, st, 1.0f, 0.0f, 1.0f, 0, prologue_bm_applies(h, batch));
        else launch_shape_solve(d, ws, batch, st, 1.0f, 0.0f, 1.0f, (bm || use_pair_form()) ? 1 : 0, 0);
        return 0;
      case SMPLFIT_KERNEL_LBS_PARTSUM: {
        if (bm) {
#define SF_CALL_LBS(S_, KW_) \
  launch_lbs_bm<S_, KW_>(h, ws, batch, st, false, false, false, false, -1, !rot_bm_applies(h, prologue_bm_applies(h, batch)))
          SF_DISPATCH_SKW(d, SF_CALL_LBS);
#undef SF_CALL_LBS
          return 0;
        }
        if (int rc = launch_lbs_any<0>(d, ws, batch, false, d.S, ws.beta, ws.trans, nullptr, st)) return rc;
        return 0;
      }
      case SMPLFIT_KERNEL_JOINT_STAGE: {  // a rotation pass + prologue on the state of the last fit
        JointStageArgs ja{};
        ja.tj = ws.tjc;
        ja.rj = ws.rjoints;
        ja.rj_shared = 0;
        ja.Gprev = ws.G;
        ja.jw = nullptr;
        ja.fit_rotations = 1;
        ja.do_prologue = 1;
        ja.joint_block = gen_joint_rows(d) ? 0 : 1;
        ja.joint_block_weighted = 0;
        ja.vertex_sa_closed_form = d.general ? 0 : 1;
        // (what a default fit runs: the rotations + k_prologue_bm where that applies)
        {
          const bool pro_h = bm && prologue_bm_applies(h, batch);
          launch_joint_stage_fit(h, ja, ws, batch, st, pro_h, rot_bm_applies(h, pro_h) ? sf::kShareLbsUsed : -1, 1);
        }
        return 0;
      }
      case SMPLFIT_KERNEL_REFINE: {  // (outputs into the workspace: ws.tvs is unused between fits)
        RefineArgs ra{};
        ra.tj = ws.tjc;
        ra.rj_term = ws.rjoints;
        ra.jw = nullptr;
        ra.final_adjust = 1;
        float* scratch = bm ? ws.tvs : ws.rverts;
        ra.pose = scratch;
        ra.betas = scratch + (size_t)batch * d.J * 3;
        ra.trans = ra.betas + (size_t)batch * d.S;
        ra.kid = nullptr;
        ra.orient = ra.trans + (size_t)batch * 3;  // (a fit always writes the orientations and the relative rotations)
        ra.rel = ra.orient + (size_t)batch * d.J * 9;
        // (what a default fit runs: k_refine_bm on the rows of the last LBS pass where that applies)
        if (refine_bm_applies(h, bm && prologue_bm_applies(h, batch))) launch_refine_bm(h, ra, share_view(h, sf::kShareLbsAdj, batch), ws, batch, st);
        else launch_refine(d, ra, ws, batch, st);
        return 0;
      }
      case SMPLFIT_KERNEL_GRAM_COMBINE:
        if (!bm) return fail(SMPLFIT_ERR_UNSUPPORTED, "normal-equation combine: batch-major path not active");
        if (solve_bm_applies(h, batch)) return fail(SMPLFIT_ERR_UNSUPPORTED, "normal-equation combine: part of k_solve_bm (SMPLFIT_KERNEL_SHAPE_SOLVE)");
        launch_residual_bm(h, ws, batch, st, 4);
        return 0;
      case SMPLFIT_KERNEL_PSUM_COMBINE:
        if (!bm) return fail(SMPLFIT_ERR_UNSUPPORTED, "part-sum combine: batch-major path not active");
        if (rot_bm_applies(h, prologue_bm_applies(h, batch))) return fail(SMPLFIT_ERR_UNSUPPORTED, "part-sum combine: k_rotations_bm / k_refine_bm add the rows themselves");
        launch_psum_combine(d, share_view(h, sf::kShareLbsUsed, batch), ws, batch, Mp, st);
        return 0;
      case SMPLFIT_KERNEL_JD_TRANSPOSE:
        if (!bm) return fail(SMPLFIT_ERR_UNSUPPORTED, "joint-row transpose: batch-major path not active");
        if (prologue_bm_applies(h, batch)) return fail(SMPLFIT_ERR_UNSUPPORTED, "joint-row transpose: k_prologue_bm writes ws.jdT itself");
        launch_jd_transpose(d, ws, batch, st);
        return 0;
      case SMPLFIT_KERNEL_MEAN_FINISH:
        if (!bm) return fail(SMPLFIT_ERR_UNSUPPORTED, "mean pass: batch-major path not active");
        // (the slab sums of the layout pass are gone from ws.resP by now: the values are arbitrary, the work is the same;
        // the outputs go where the fit left them)
        hipLaunchKernelGGL(k_mean_finish, dim3(Mp / 64), dim3(64 * kMeanWaves), 0, st, d, ws.rjoints, ws.resP, ws, batch, Mp,
                           (d.V + kSlabV - 1) / kSlabV);
        return 0;
      case SMPLFIT_KERNEL_LBS_LAST:
        if (!bm) return fail(SMPLFIT_ERR_UNSUPPORTED, "last LBS pass: batch-major path not active");
#define SF_CALL_LBS(S_, KW_) \
  launch_lbs_bm<S_, KW_>(h, ws, batch, st, false, true, false, false, -1, !refine_bm_applies(h, prologue_bm_applies(h, batch)))
        SF_DISPATCH_SKW(d, SF_CALL_LBS);
#undef SF_CALL_LBS
        return 0;
      default: return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_time_kernel_f32: unknown kernel id");
    }
  };
  // Each timed launch runs right after the kernel that precedes it inside a fit (K2 before K3, K3
  // before K5), so caches are in the state the kernel sees in situ; only the target kernel is
  // bracketed by the two events.
  const bool nopre = getenv("SMPLFIT_TIME_NOPRE") != nullptr;  // (measurement hook only: the kernel without its producer in front)
  auto pre = [&]() {
    if (nopre) return 0;
    if (kernel_id == SMPLFIT_KERNEL_SHAPE_ACCUM) launch_gemm(d, ws, batch, st, bm);
    if ((kernel_id == SMPLFIT_KERNEL_LBS_PARTSUM || kernel_id == SMPLFIT_KERNEL_LBS_LAST) && bm) {
      if (solve_bm_applies(h, batch)) launch_solve_bm(h, ws, batch, st, 1.0f, 0.0f, 1.0f, 0, prologue_bm_applies(h, batch));
      else launch_shape_solve(d, ws, batch, st, 1.0f, 0.0f, 1.0f, 1, 0);
    }
    if (kernel_id == SMPLFIT_KERNEL_LBS_PARTSUM && !bm) {
      if (int rc = launch_accum_any(d, ws, batch, false, st)) return rc;
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

#ifdef SMPLFIT_GEMM_SHARED_CU
// Debug builds of the neighbour-interaction probe (tools/lds_probe.py): a victim that fills its LDS once and then only
// READS it — uniform-address and per-lane addresses, 4- and 16-byte reads — checking every value; a wrong value is
// logged with its lane, address and a second read of the same cell (transient read fault or changed LDS contents?),
// and the whole table is verified once more at the end.
__global__ __launch_bounds__(64) void k_lds_victim(uint32_t* __restrict__ log, int* __restrict__ nlog, int iters, int maxlog) {
  __shared__ __attribute__((aligned(16))) uint32_t tab[2048];
  const int lane = threadIdx.x;
  const uint32_t salt = blockIdx.x * 977u;
  auto val = [&](uint32_t i) { return (i * 2654435761u) ^ salt; };
  for (int i = lane; i < 2048; i += 64) tab[i] = val(i);
  __syncthreads();
  volatile uint32_t* vt = tab;
  auto report = [&](int it, int kind, uint32_t addr, uint32_t got) {
    const uint32_t again = vt[addr];
    const int k = atomicAdd(nlog, 1);
    if (k < maxlog) {
      uint32_t* r = log + (size_t)k * 8;
      r[0] = blockIdx.x; r[1] = (uint32_t)it; r[2] = (uint32_t)lane; r[3] = (uint32_t)kind; r[4] = addr; r[5] = got;
      r[6] = val(addr); r[7] = again;
    }
  };
  for (int it = 0; it < iters; ++it) {
    const uint32_t au = (uint32_t)(it * 7) & 2047u, al = (uint32_t)(it * 13 + lane * 5) & 2047u;
    const uint32_t u = vt[au];                                   // uniform address, 4 bytes
    if (u != val(au)) report(it, 0, au, u);
    const uint32_t l = vt[al];                                   // per-lane address, 4 bytes
    if (l != val(al)) report(it, 1, al, l);
    const uint32_t a4 = (uint32_t)(it * 28) & 2044u;             // uniform address, 16 bytes
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 q = *reinterpret_cast<volatile u32x4*>(tab + a4);
    if (q.x != val(a4)) report(it, 2, a4, q.x);
    if (q.y != val(a4 + 1)) report(it, 2, a4 + 1, q.y);
    if (q.z != val(a4 + 2)) report(it, 2, a4 + 2, q.z);
    if (q.w != val(a4 + 3)) report(it, 2, a4 + 3, q.w);
    const uint32_t b4 = ((uint32_t)(it * 4 + lane * 12)) & 2044u;  // per-lane address, 16 bytes
    const u32x4 p = *reinterpret_cast<volatile u32x4*>(tab + b4);
    if (p.x != val(b4)) report(it, 3, b4, p.x);
    if (p.w != val(b4 + 3)) report(it, 3, b4 + 3, p.w);
  }
  for (int i = lane; i < 2048; i += 64)
    if (vt[i] != val(i)) report(-1, 4, (uint32_t)i, vt[i]);     // the contents at the end
}

// Second victim: packed-fp32 FMAs whose operands are uniform-address 16-byte LDS reads (the form of the round-2 / 3
// victims); every value is a small integer, so the sums are exact and must equal the solo run bit for bit.
__global__ __launch_bounds__(64) void k_lds_victim_fma(float* __restrict__ out, int iters, int rewrite) {
  __shared__ __attribute__((aligned(16))) float tab[2048];
  const int lane = threadIdx.x;
  for (int i = lane; i < 2048; i += 64) tab[i] = (float)((i * 5 + blockIdx.x) % 7);
  __syncthreads();
  f2 acc[16];
  for (int k = 0; k < 16; ++k) acc[k] = mk2(0.f, 0.f);
  const f2 v = mk2((float)(lane % 3), (float)((lane + 1) % 3));
  for (int it = 0; it < iters; ++it) {
    const float4* row = reinterpret_cast<const float4*>(tab) + ((it * 8) & 511);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float4 c = row[u];  // uniform address: broadcast read
      acc[2 * u] += mk2(c.x, c.y) * v;
      acc[2 * u + 1] += mk2(c.z, c.w) * v;
    }
    if (rewrite && (it & 15) == 15) {  // the victim also rewrites its table (same values), as a double-buffered kernel does
      __syncthreads();
      for (int i = lane; i < 2048; i += 64) tab[i] = (float)((i * 5 + blockIdx.x) % 7);
      __syncthreads();
    }
    if ((it & 63) == 63)
      for (int k = 0; k < 16; ++k) {
        acc[k].x = acc[k].x > 4.0e5f ? acc[k].x - 4.0e5f : acc[k].x;
        acc[k].y = acc[k].y > 4.0e5f ? acc[k].y - 4.0e5f : acc[k].y;
      }
  }
  for (int k = 0; k < 16; ++k) {
    out[((size_t)blockIdx.x * 32 + 2 * k) * 64 + lane] = acc[k].x;
    out[((size_t)blockIdx.x * 32 + 2 * k + 1) * 64 + lane] = acc[k].y;
  }
}

int smplfit_debug_lds_victim_fma(void* stream, int nwg, int iters, int rewrite, float* out) {
  hipLaunchKernelGGL(k_lds_victim_fma, dim3(nwg), dim3(64), 0, (hipStream_t)stream, out, iters, rewrite);
  return post_launch_check();
}

int smplfit_debug_lds_victim(void* stream, int nwg, int iters, uint32_t* log, int* nlog, int maxlog) {
  hipLaunchKernelGGL(k_lds_victim, dim3(nwg), dim3(64), 0, (hipStream_t)stream, log, nlog, iters, maxlog);
  return post_launch_check();
}
#endif

#ifdef SMPLFIT_WAVE_STAMPS
// debug builds: the stamps of the last k_residual_bm launch (n waves x 5 values), see kernels_bm.inc
int smplfit_debug_wave_stamps(unsigned long long* dst, int n) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_wave_stamps), (size_t)n * 5 * 8) == hipSuccess ? 0 : -1;
}
#endif

#ifdef SMPLFIT_SOLVE_STAMPS
// debug builds: the phase stamps of the last k_solve_bm launch (n workgroups x 8 values), see kernels_bm.inc
int smplfit_debug_solve_stamps(unsigned long long* dst, int n) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_solve_stamps), (size_t)n * 8 * 8) == hipSuccess ? 0 : -1;
}
#endif

int smplfit_primitives_f32(int primitive_id, const float* a, const float* b, float* out, int n,
                           void* hip_stream) {
  if (!a || !out || n < 0) return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_primitives_f32: null pointer / negative n");
  if (primitive_id < SMPLFIT_PRIM_PROJ_SO3 || primitive_id > SMPLFIT_PRIM_SWING_TWIST)
    return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_primitives_f32: unknown primitive id");
  if ((primitive_id == SMPLFIT_PRIM_ALIGN_UNIT || primitive_id == SMPLFIT_PRIM_SWING_TWIST) && !b)
    return fail(SMPLFIT_ERR_BAD_ARG, "smplfit_primitives_f32: this primitive takes two inputs");
  if (n == 0) return SMPLFIT_OK;
  hipLaunchKernelGGL(k_primitives, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)hip_stream, primitive_id, a, b,
                     out, n);
  return post_launch_check();
}

}  // extern "C"
