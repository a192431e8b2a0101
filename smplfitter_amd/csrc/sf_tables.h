// Host-side tables derived from a body model: the native counterpart of BodyFitter.__init__
// (reference src/smplfitter/pt/bodyfitter.py:25-233), emitted in the layouts the HIP kernels read.
//
// Layout decisions:
//  * vertices are re-ordered ("sorted slots") by body part — used parts first, in part order — so a
//    64-lane wave tile belongs to ONE part (wave-level segmented sums, LDS broadcasts of joint data)
//    and every per-vertex constant is stored SoA over sorted slots, padded to Vp (multiple of 128);
//  * skinning weights are kept sparse: KW (4 or 8) (joint, weight) pairs per vertex;
//  * posedirs is stored K-major [Kp][3*Vp] (n = c*Vp + slot) for the fp32 MFMA GEMM.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/smplfit.h"

namespace sf {

constexpr int kMaxJoints = 64;
constexpr int kTile = 64;        // wave width
constexpr int kVertexPad = 128;  // Vp granularity (GEMM N tile = 128 divides 3*Vp)
constexpr int kGemmKPad = 16;    // posedirs K padded to the GEMM K step
// Products per k-step of the split-bf16 posedirs GEMMs (see kernels_wave.inc, kGemm3): 3 by default.  The host images
// of posedirs depend on it (with three products the third bf16 plane is only kept for the k-step of the bias row).
#ifndef SMPLFIT_GEMM_PRODUCTS
#define SMPLFIT_GEMM_PRODUCTS 3
#endif
constexpr bool kGemm3 = SMPLFIT_GEMM_PRODUCTS == 3;
constexpr int kJdStride = 52;    // floats per joint in the per-instance joint block (see sf_stages.h)

enum PartType : int32_t { kPartNone = 0, kPartMulti = 1, kPartBone = 2, kPartLeaf = 3 };

struct Segment {
  int32_t start, count, part;
};

// Vertex group of the batch-major ("lane = instance") vertex kernels: a run of sorted slots of ONE part
// whose skinning joints number at most kGroupJoints; the group's joint list is staged once per
// workgroup and the per-vertex records address it by local slot.
constexpr int kGroupJoints = 12;
constexpr int kBmWaves = 4;   // waves of a batch-major vertex workgroup: they split the group's vertices evenly (measured in
                              // round 3 with the piece kernels: 2 waves 1.89, 4 waves 1.95, 8 waves 1.90 M fits/s)
// A wave's share of a group is cut into PIECES: maximal runs of slots whose skinning joints number at most four
// together.  Inside a piece the vertex loops keep those four joints' records in registers (no LDS read per vertex)
// and the per-joint sums in four accumulators; a vertex's record holds its weights in the piece's joint order.
// Record: the piece's vertex count (a wave's pieces are contiguous from its first slot), joints[4] (model joint ids,
// ascending, padded with the first), local[4] (their slots in the group's joint list), 3 unused.  The table ends
// with one all-zero record, so that reading one record past a wave's last piece is always valid.
constexpr int kPieceRec = 12;
struct VertexGroup {
  int32_t start, count, part, used, nq;
  int32_t joints[kGroupJoints];  // padded with joints[0]
};

struct HostTables {
  int V = 0, J = 0, S = 0, P = 0;  // S counts every shape unknown: betas + kid
  int n_kid = 0;                   // 1 if the last unknown is the kid blend shape
  int n_pad = 0;                   // zero shape directions between the betas and the kid unknown: the kernels
                                   // are built for 10 / 16 betas, a model with fewer is padded up (unit ridge)
  int num_betas() const { return S - n_kid - n_pad; }  // the caller's betas
  int Vp = 0, Kp = 0, KW = 4;
  // max over vertices of |sum of skinning weights - 1|: the batch-major residual kernel derives the residual sum
  // from the per-joint moments, which is exact only for normalised weights (the other kernels keep the sum explicitly)
  float wsum_dev = 0.f;
  bool smpl_family = false;
  bool has_regressor = false;

  // kinematic tree
  std::vector<int32_t> parents;         // (J) parents[0] = 0 here ("parents_with_root")
  std::vector<int32_t> fk_js;           // joints of levels 1.. concatenated
  std::vector<int32_t> fk_level_start;  // (L+1) offsets into fk_js
  std::vector<int32_t> cas_start, cas_flat;  // children-and-self lists
  std::vector<int32_t> part_type;       // (J)
  std::vector<int32_t> toe_src;         // (J) copy-from part or -1
  std::vector<int32_t> adj_flag;        // (J)
  std::vector<int32_t> adj_level_start, adj_parts;  // adjustable parts per level (levels as fk)
  int adj_last_level = -1;
  std::vector<int32_t> used_part;       // (J)

  // vertex ordering
  std::vector<int32_t> part_assignment;  // (V) original order
  std::vector<int32_t> perm;             // (Vp) original index of sorted slot, -1 padding
  std::vector<int32_t> slot_part;        // (Vp) part of sorted slot, -1 padding
  int n_used = 0;                        // slots [0, n_used) belong to used parts
  std::vector<Segment> segments;         // part-aligned tiles over [0, n_used)

  // per-slot constants (SoA over Vp)
  std::vector<float> vt;        // (3,Vp)  v_template
  std::vector<float> dm;        // (3,Vp)  default mesh = forward(zero pose, zero betas)
  std::vector<float> sd;        // (3*S,Vp) shapedirs, row index c*S+s
  std::vector<uint32_t> widx;   // (KW/4, Vp) 4 joint ids per word, byte k = k-th pair
  std::vector<float> wval;      // (KW, Vp)
  std::vector<float> pdT;       // (Kp, 3*Vp) posedirs, K-major, rows in rp_pos() (parity-major) order; row rp_pos(P) = v_template
  std::vector<float> pdSw;      // (3*Vp/32, 32, Kp) the same, transposed per 32-column tile (A-stationary GEMM)
  // Split-bf16 image of pdSw for the matrix-core GEMM (Kp == 208 only): per 32-column tile the planes hi, mid (, lo;
  // hi + mid + lo == the fp32 value, each rounded to nearest bf16) of [32 n][26 slots][8 k], slot = k / 8 with bit 0
  // flipped for rows with (n >> 3) & 1 (bank-conflict-free 16-byte LDS reads from unpadded 416-byte rows).  With
  // three products per k-step (kGemm3) the lo plane is kept for the LAST k-step only (the bias row's: [32 n][2 slots][8 k],
  // slot = (k / 8) & 1), 27 KB per tile instead of 39.  The tile image is copied to LDS verbatim.
  std::vector<uint16_t> pdB;
  // split-bf16 stage images of posedirs for the tiled GEMM (Kp != 208, e.g. SMPL-X): per 128-column tile and
  // 32-k stage three planes [128 n][4 slots][8 k] (24 KB), slot = (k >> 3) ^ ((n >> 2) & 3); K padded to kc32 * 32
  // (with three products per k-step the kernel copies the third plane for the bias row's stage only)
  std::vector<uint16_t> pdB2;
  int kc32 = 0;  // stages of 32 k
  // per-vertex constants packed per 64-vertex tile for cooperative staging through LDS:
  // cstride() floats per vertex = [shapedirs s-major (s*3+c), 3*S | KW weights | KW/4 index words | pad]
  std::vector<float> cpackA;    // (Vp/64, 64, cstride) dense tiles of sorted slots  (shape accumulate)
  std::vector<float> cpackB;    // (nseg, 64, cstride)  part-aligned segments        (LBS + part sums)

  // per-joint constants
  std::vector<float> j_ext;     // (J,3,S+1)  [J_template | J_shapedirs]
  std::vector<float> bone_ext;  // (J,3,S+1)  j_ext - j_ext[parent] (root: j_ext - j_ext[0] = 0)
  // the level FK of the joint stage without dependent table reads: per position of fk_js the packed (joint | parent << 16)
  // and a copy of the joint's bone_ext rows in that order
  std::vector<int32_t> fk_jp;
  std::vector<float> bone_lv;   // (len(fk_js),3,S+1)
  std::vector<float> sa0;       // (J,3) sum of default-mesh vertices per part (template pass)
  std::vector<float> sw0;       // (J)   vertex count per part
  std::vector<float> cs_joint;  // (J,3,S) sum_v w_vj shapedirs_v  (closed-form SA of the vertex block)
  std::vector<float> cw_joint;  // (J)     sum_v w_vj

  // ---- "pair-Gram" form of the shape solve (unit weights): the Gramian of the vertex block
  // depends on the rotations only, G = sum over joint pairs of small contractions with these
  // constants; off-diagonal pairs j < j' that share at least one vertex:
  std::vector<int32_t> pair_j;   // (np, 2)
  std::vector<float> pair_c1;    // (np, 9, S, S)  sum_v w_vj w_vj' S_v[a][i] S_v[a'][i'],  [a*3+a'][i][i']
  std::vector<float> pair_c2;    // (np, 3, S)     sum_v w_vj w_vj' S_v[a][i]
  std::vector<float> pair_c3;    // (np)           sum_v w_vj w_vj'
  std::vector<float> diag_g0;    // (S, S)  sum_j sum_a C1_jj[a][a][i][i']   (R_j^T R_j = I)
  std::vector<float> diag_c2;    // (J, 3, S)
  std::vector<float> diag_c3;    // (J)
  // tiles of the residual kernel: part-aligned, <= 64 vertices, <= 16 distinct joints, over ALL
  // slots; blob per tile = [64 x cstride() vertex records | 16 x 64 MFMA A-operand weights
  // (step t, lane l -> weight of vertex 4t + l/16 for joint slot l%16) | 16 joint ids (pad = J)]
  std::vector<Segment> gtiles;
  // batch-major kernels: vertex groups and per-slot records = cpackA rows whose index words hold the
  // LOCAL joint slots of the group (byte k = slot of the k-th skinning pair), and the dense weights over
  // the group's joint list
  std::vector<VertexGroup> groups;
  // the batch-major pair-Gram kernel reads rows of shape values as aligned register PAIRS: its copies of the
  // constants have the y axis padded to an even length SE = S rounded up to 2 (the padding is zero)
  std::vector<float> pair_c1x;   // pair_c1 re-laid out as (np, S [x], 3 [a], 3 [a'], SE [y])
  std::vector<float> pair_c2e;   // pair_c2 as (np, 3, SE)
  std::vector<float> diag_c2e;   // diag_c2 as (J, 3, SE)
  int s_even() const { return (S + 1) & ~1; }
  // brec row (brec_stride() floats, fetched with scalar loads): [sd_x : S][sd_y : S][sd_z : S][pad to a
  // multiple of 4][4 weights in the joint order of the vertex's piece]
  std::vector<float> brec;       // (Vp, brec_stride())
  int brec_w() const { return (3 * S + 3) / 4 * 4; }  // offset of the weights
  int brec_stride() const { return brec_w() + 4; }
  std::vector<int32_t> pieces;       // (npieces, kPieceRec)
  std::vector<int32_t> piece_start;  // (ngroups * kBmWaves + 1) first piece of every (group, wave)
  std::vector<float> gblob;      // (ngt, gblob_stride())
  int gblob_stride() const { return 64 * cstride() + 16 * 64 + 16; }

  // sparse post-LBS joint regressor, CSR over sorted slots (joints-omitted path)
  std::vector<int32_t> reg_start, reg_slot;
  std::vector<float> reg_val;
  std::vector<float> reg_rowsum;  // (J) sum of each regressor row (regressed joints of a translated mesh)

  int num_levels() const { return (int)fk_level_start.size() - 1; }
  // floats per vertex in cpack: multiple of 4 (16-B rows) and = 4 (mod 8) so that 16 consecutive
  // lanes reading 16 B at this stride hit distinct LDS banks
  int cstride() const {
    int n = (3 * S + KW + KW / 4 + 3) / 4 * 4;
    return n % 8 == 4 ? n : n + 4;
  }
  int ne() const { return S * (S + 1) / 2 + S + 3 * S + 3; }  // normal-equation entries (+1 for W)
};

// Returns "" on success, else an error message (and `unsupported` tells which status to use).
void build_tiled_gemm_images(HostTables& t);
std::string build_tables(const smplfit_model_desc& d, HostTables& t, bool* unsupported);

}  // namespace sf
