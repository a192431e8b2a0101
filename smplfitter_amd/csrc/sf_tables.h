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

// shape-unknown counts (betas + kid) the batch-major vertex kernels are instantiated for
constexpr bool bm_shape_count(int S) { return S == 10 || S == 11 || S == 16 || S == 17; }

enum PartType : int32_t { kPartNone = 0, kPartMulti = 1, kPartBone = 2, kPartLeaf = 3 };

struct Segment {
  int32_t start, count, part;
};

// Batch-major ("lane = instance") vertex kernels: the sorted slots are cut once into PIECES — maximal runs of slots
// of ONE part whose skinning joints number at most four together (within a part the slots are sorted by joint set, so
// the runs are long).  Inside a piece the vertex loops keep those four joints' records in registers (no LDS read per
// vertex) and the per-joint sums in four accumulators; a vertex's record (HostTables::brec) holds its weights in the
// piece's joint order (ascending joint id).
struct VertexPiece {
  int32_t start, count, part, used, nj;
  int32_t joints[8];  // ascending, padded with joints[0] (padding joints carry weight 0); HostTables::NJ of them are used
};
// The work of one instance block (64 instances) is cut into NC CELLS: runs of whole or split pieces of (nearly) equal
// cost — cost = steps (a piece of odd length takes one padding step) + kPieceCost per piece (the exposed load of its
// joints); NC = the largest power of two that leaves a cell ~75+ steps (SMPL 64, SMPL-X 128, a 1024-vertex subset 16).
// A wave (a SHARE) walks `mult` consecutive cells; a launch picks the multiplier whose (rounds x share length) is
// smallest for its number of instance blocks (pick_share_mult): at B = 4096 that is ONE round of 4096 equally long
// waves where the part-aligned vertex groups of round 3 took 1.5 uneven rounds of workgroups.  No wave waits for
// another one: every wave writes its own partial sums, restarting them at every cell boundary, so the ROWS of partial
// sums (ws.psumP / ws.resP) and the order in which the combine kernels add them are the same whatever the multiplier:
// an instance's result does not depend on the batch it is fitted in.
constexpr int kGroupJoints = 12;  // local joint slots of a residual SEGMENT (moment accumulators of a wave in LDS)
#ifndef SMPLFIT_BM_WAVES
#define SMPLFIT_BM_WAVES 4
#endif
#ifndef SMPLFIT_CELL_CAP
#define SMPLFIT_CELL_CAP 256  // at most this many cells
#endif
#ifndef SMPLFIT_CELL_STEPS
#define SMPLFIT_CELL_STEPS 75  // a cell holds at least this many steps (and fewer than twice as many)
#endif
// SMALL BATCHES take a second, FINE set of the same tables.  Up to a few hundred instances a vertex pass is one round
// of lonely waves — one per SIMD at most — and its duration is the WALK of a cell: ~110 steps at the issue rate of a
// single wave (0.37 us a step: 40 us of residual pass at B = 32, where 0.2 us would move the bytes).  The fine tables
// deal the same domains to cells of kFineCellSteps+ steps (512 cells for the SMPL-shaped model: a 10 us walk); their
// eight times as many rows of partial sums — more bytes than the streams themselves — are why the large batches keep
// the coarse ones.  The two regimes add the partial sums in different orders: an instance's result is independent of
// its batch WITHIN a regime (batches up to kFineMaxBatch / above), and agrees to rounding across.  Measured (ms per
// default fit, coarse -> fine): B = 32 0.57 -> 0.36, 256 0.61 -> 0.40, 512 0.65 -> 0.50, 768 0.70 -> 0.60, 1024 0.74 -> 0.72
// (SMPL-X-shaped: 0.93 -> 0.77, 1.04 -> 0.87, 1.09 -> 0.99, 1.23 -> 1.20, 1.28 -> 1.34).
#ifndef SMPLFIT_FINE_CELL_STEPS
#define SMPLFIT_FINE_CELL_STEPS 12
#endif
constexpr int kFineCellSteps = SMPLFIT_FINE_CELL_STEPS;
constexpr int kFineCellCap = 1024;
#ifndef SMPLFIT_FINE_MAX_B
#define SMPLFIT_FINE_MAX_B 768
#endif
constexpr int kFineMaxBatch = SMPLFIT_FINE_MAX_B;  // (workspace sizes up to this batch cover the fine rows)
#ifndef SMPLFIT_PIECE_COST
#define SMPLFIT_PIECE_COST 3
#endif
constexpr int kBmWaves = SMPLFIT_BM_WAVES;  // waves (= shares) per workgroup of the batch-major vertex kernels (1, 2 or 4)
constexpr int kPieceCost = SMPLFIT_PIECE_COST;
// Cell piece record (scalar loads): [0] vertex count, [1..4] joints, [5..8] their local slots in the current residual
// segment, [9] first slot, [10] the output row block written AFTER this piece (-1: none), [11] residual tables: the
// row's joint count | (cell index + 1) << 8 behind the LAST piece of a cell (there the wave also writes the cell's
// r1 | Sb record).  Every table ends with one all-zero record: reading one record past the last piece is valid.
// Models with 5-8 skinning weights per vertex (HostTables::NJ == 8, round 5): a piece holds up to EIGHT joints and its
// record is a PAIR of such records — the second one carries joints 4..7 in [1..4] and their local slots in [5..8] (its
// other fields repeat the first one's): piece_rec() = 24.
constexpr int kPieceRec = 12;
enum ShareKind : int {
  kShareResidual = 0,  // all slots; a row = a SEGMENT (run of pieces of one cell whose joints number <= kGroupJoints): moments
  kShareLbsAll = 1,    // all slots; a row = the part sums of a run of one part inside a cell (joints-omitted fits, forward)
  kShareLbsUsed = 2,   // the slots of the used parts (bodyfitter.py:109-114)
  kShareLbsAdj = 3,    // the slots of the adjustable parts only: the LAST part sums of a fit feed the dependent
                       // refinement alone, which reads them at the adjustable parts (bodyfitter.py:1505-1517)
  kShareKinds = 4,
  kShareFine = 4  // HostTables::shares[kShareFine + kind]: the fine table of a kind
};
struct ShareTable {
  int ncells = 0, nrows = 0, max_cost = 0;
  int rec = kPieceRec;               // ints per piece record (HostTables::piece_rec())
  std::vector<int32_t> piece_start;  // (ncells + 1)
  std::vector<int32_t> pieces;       // (npieces + 1, kPieceRec)
  std::vector<int32_t> row_part;     // (nrows) LBS tables: the part of a row
  std::vector<int32_t> row_joints;   // (nrows, kGroupJoints) residual tables: joint of every local slot, -1 = unused
};

struct HostTables {
  int V = 0, J = 0, S = 0, P = 0;  // S counts every shape unknown: betas + kid
  int n_kid = 0;                   // 1 if the last unknown is the kid blend shape
  int n_pad = 0;                   // zero shape directions between the betas and the kid unknown: the kernels
                                   // are built for 10 / 16 betas, a model with fewer is padded up (unit ridge)
  int num_betas() const { return S - n_kid - n_pad; }  // the caller's betas
  int Vp = 0, Kp = 0, KW = 4;
  int NJ = 4;                      // joints of a vertex piece of the batch-major kernels: 4 (KW == 4) or 8 (KW == 8)
  int piece_rec() const { return NJ == 8 ? 2 * kPieceRec : kPieceRec; }
  // max over vertices of |sum of skinning weights - 1|: the batch-major residual kernel derives the residual sum
  // from the per-joint moments, which is exact only for normalised weights (the other kernels keep the sum explicitly)
  float wsum_dev = 0.f;
  bool smpl_family = false;
  bool has_regressor = false;
  // more than 16 betas or more than 8 skinning weights per vertex: the GENERAL path (kernels_gen.inc) — run-time loops
  // over the unknowns / the weights, stage scratch in global memory; none of the S-templated tables below is built
  bool general = false;

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
  std::vector<float> sdg;       // (Vp,3,S4) the same vertex-major, rows padded to S4 = S rounded up to 4 (general path only)
  std::vector<Segment> segments_all;  // part-aligned tiles over [0, V)
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
  // batch-major kernels: the pieces of the sorted slots and the cell tables cut from them, one per ShareKind
  std::vector<VertexPiece> vpieces;
  std::vector<ShareTable> shares;  // (2 kShareKinds: coarse, fine), empty when the model has no batch-major tables
  int share_fallback = 0;  // bit k: shares[k] is a copy of a wider / coarser table (build_share_tables); 0xffff: no tables
  // the batch-major pair-Gram kernel reads rows of shape values as aligned register PAIRS: its copies of the
  // constants have the y axis padded to an even length SE = S rounded up to 2 (the padding is zero)
  std::vector<float> pair_E;     // (np, 9 [a a'], ng_pad()) symmetrised pair_c1 over the upper triangle (i <= i2, row-major)
  std::vector<int32_t> jn_start, jn;  // per joint: its (other joint, pair) neighbours, CSR (J + 1) / (n, 2)
  int ng_pad() const { return (S * (S + 1) / 2 + 3) / 4 * 4; }
  std::vector<float> pair_c2e;   // pair_c2 as (np, 3, SE)
  std::vector<float> diag_c2e;   // diag_c2 as (J, 3, SE)
  int s_even() const { return (S + 1) & ~1; }
  // brec row (brec_stride() floats, fetched with scalar loads): [sd_x : S][sd_y : S][sd_z : S][pad to a
  // multiple of 4][NJ weights in the joint order of the vertex's piece]
  std::vector<float> brec;       // (Vp, brec_stride())
  int brec_w() const { return (3 * S + 3) / 4 * 4; }  // offset of the weights
  int brec_stride() const { return brec_w() + NJ; }
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
// Cells per share (1, 2, 4, ...) of a batch-major vertex pass of `kind` over `nblocks` instance blocks on `slots`
// resident waves: the multiplier with the smallest rounds x (share length + prologue) estimate.
int pick_share_mult(const HostTables& t, int kind, int nblocks, int slots, int wg_waves = kBmWaves);
void build_share_tables(HostTables& t);

}  // namespace sf
