// Host table builder — see sf_tables.h.  Reference: src/smplfitter/pt/bodyfitter.py:25-233.
#include "sf_tables.h"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <numeric>

namespace sf {

static int round_up(int x, int m) { return (x + m - 1) / m * m; }

// ---- share tables of the batch-major vertex kernels (see sf_tables.h) ----
namespace {
struct PieceRun {  // a whole or split piece on its way into a share
  int start, count, part, nj;
  const int32_t* joints;
};
int piece_cost(int count) { return count + (count & 1) + kPieceCost; }
// one piece record: 12 ints, or — pieces of up to eight joints — a pair of them, the second with joints / slots 4..7
void emit_piece(std::vector<int32_t>& out, int NJ, int count, const int32_t* joints, const int* slots, int start, int row, int info) {
  for (int half = 0; half < NJ / 4; ++half) {
    int32_t rec[kPieceRec] = {0};
    rec[0] = count;
    for (int q = 0; q < 4; ++q) {
      rec[1 + q] = joints[4 * half + q];
      rec[5 + q] = slots ? slots[joints[4 * half + q]] : 0;
    }
    rec[9] = start;
    rec[10] = row;
    rec[11] = info;
    out.insert(out.end(), rec, rec + kPieceRec);
  }
}

void build_share_table(const HostTables& t, int kind, ShareTable& out, int cell_steps, int cell_cap) {
  out = ShareTable();
  const int NJ = t.NJ, RS = t.piece_rec();
  out.rec = RS;
  std::vector<PieceRun> dom;
  long total = 0;
  for (const auto& p : t.vpieces) {
    const bool in = kind == kShareLbsUsed ? p.used != 0 : kind == kShareLbsAdj ? t.adj_flag[p.part] != 0 : true;
    if (!in) continue;
    dom.push_back({p.start, p.count, p.part, p.nj, p.joints});
    total += piece_cost(p.count);
  }
  // cells of ~75+ steps, a power of two of them (so that the share counts of power-of-two batches fill whole rounds)
  int ns = 8;
  while (ns < cell_cap && (long)ns * 2 * cell_steps <= total) ns *= 2;
  // deal the domain to ns cells of equal cost: whole pieces while they fit, then the head of the next one (an even
  // number of vertices, so that a split adds no padding step).  Every boundary that splits a piece costs the piece
  // overhead once more: the budget of a cell counts it for the boundaries still ahead.
  std::vector<std::vector<PieceRun>> sh;
  for (;; ns /= 2) {
    sh.assign(ns, {});
    size_t pi = 0;
    PieceRun cur{};
    bool have = false;
    long work = total;  // cost of the pieces (and the remainder of a split one) not dealt yet
    for (int k = 0; k < ns; ++k) {
      long budget = (work + (long)kPieceCost * (ns - 1 - k) + (ns - k) - 1) / (ns - k);
      while (have || pi < dom.size()) {
        if (!have) {
          cur = dom[pi++];
          have = true;
        }
        const int c = piece_cost(cur.count);
        if (c <= budget || k + 1 == ns) {
          sh[k].push_back(cur);
          budget -= c;
          work -= c;
          have = false;
          continue;
        }
        const int x = (int)((budget - kPieceCost) & ~1L);  // vertices of the head
        if (x >= 2 && x < cur.count) {
          PieceRun head = cur;
          head.count = x;
          sh[k].push_back(head);
          work -= x;
          cur.start += x;
          cur.count -= x;
        } else if (sh[k].empty()) {  // (a cell is never left empty)
          sh[k].push_back(cur);
          work -= c;
          have = false;
        }
        break;
      }
    }
    bool empty = false;
    for (int k = 0; k < ns; ++k) empty |= sh[k].empty();
    if (!empty || ns <= 8) break;
  }
  out.ncells = ns;
  // records, rows
  out.piece_start.assign(1, 0);
  int row = 0;
  for (int k = 0; k < ns; ++k) {
    const auto& ps = sh[k];
    int cost = 0;
    if (kind == kShareResidual) {
      // segments: runs of pieces whose joints number at most kGroupJoints together
      size_t a = 0;
      while (a < ps.size()) {
        int slots[kMaxJoints];
        for (int j = 0; j < kMaxJoints; ++j) slots[j] = -1;
        int nq = 0;
        size_t b = a;
        std::vector<int> order;
        while (b < ps.size()) {
          int add = 0;
          for (int q = 0; q < ps[b].nj; ++q)
            if (slots[ps[b].joints[q]] < 0) {
              bool dup = false;
              for (int q2 = 0; q2 < q; ++q2) dup |= ps[b].joints[q2] == ps[b].joints[q];
              if (!dup) ++add;
            }
          if (nq + add > kGroupJoints) break;
          for (int q = 0; q < ps[b].nj; ++q)
            if (slots[ps[b].joints[q]] < 0) {
              slots[ps[b].joints[q]] = nq++;
              order.push_back(ps[b].joints[q]);
            }
          ++b;
        }
        for (size_t i = a; i < b; ++i) {
          emit_piece(out.pieces, NJ, ps[i].count, ps[i].joints, slots, ps[i].start, i + 1 == b ? row : -1,
                     (i + 1 == b ? nq : 0) | (i + 1 == ps.size() ? (k + 1) << 8 : 0));
          cost += piece_cost(ps[i].count);
        }
        for (int q = 0; q < kGroupJoints; ++q) out.row_joints.push_back(q < nq ? order[q] : -1);
        ++row;
        a = b;
      }
    } else {
      for (size_t i = 0; i < ps.size(); ++i) {
        const bool last = i + 1 == ps.size() || ps[i + 1].part != ps[i].part;
        emit_piece(out.pieces, NJ, ps[i].count, ps[i].joints, nullptr, ps[i].start, last ? row : -1, 0);
        cost += piece_cost(ps[i].count);
        if (last) {
          out.row_part.push_back(ps[i].part);
          ++row;
        }
      }
    }
    out.max_cost = std::max(out.max_cost, cost);
    out.piece_start.push_back((int32_t)(out.pieces.size() / RS));
  }
  out.nrows = row;
  out.pieces.insert(out.pieces.end(), RS, 0);  // sentinel
  for (int k = 0; k < ns; ++k)
    if (out.piece_start[k + 1] == out.piece_start[k]) {  // (an empty cell: no table, see build_share_tables)
      if (std::getenv("SMPLFIT_DUMP_SHARES")) std::fprintf(stderr, "kind %d: cell %d of %d is empty (total %ld)\n", kind, k, ns, total);
      out.ncells = 0;
    }
}
}  // namespace

void build_share_tables(HostTables& t) {
  t.shares.clear();
  if (t.vpieces.empty()) return;
  t.shares.resize(2 * kShareKinds);
  for (int k = 0; k < 2 * kShareKinds; ++k) {
    if (k < kShareFine) build_share_table(t, k, t.shares[k], SMPLFIT_CELL_STEPS, SMPLFIT_CELL_CAP);
    else build_share_table(t, k - kShareFine, t.shares[k], kFineCellSteps, kFineCellCap);
  }
  // A domain too small for its cells (ncells == 0) falls back PER KIND: the adjustable-parts table to the used-parts
  // table of the same granularity (a superset: the extra part sums are not read by the refinement), a fine table to its
  // coarse one (walked by the split combines all the same).  Only a model whose coarse residual / all-slots / used-parts
  // table is empty stays on the wave-per-instance kernels.  HostTables::share_fallback (smplfit_info) says which.
  t.share_fallback = 0;
  for (int g = 0; g < 2; ++g) {
    ShareTable& adj = t.shares[g * kShareFine + kShareLbsAdj];
    const ShareTable& used = t.shares[g * kShareFine + kShareLbsUsed];
    if (adj.ncells == 0 && used.ncells != 0) {
      adj = used;
      t.share_fallback |= 1 << (g * kShareFine + kShareLbsAdj);
    }
  }
  for (int k = 0; k < kShareKinds; ++k)
    if (t.shares[kShareFine + k].ncells == 0 && t.shares[k].ncells != 0) {
      t.shares[kShareFine + k] = t.shares[k];
      t.share_fallback |= 1 << (kShareFine + k);
    }
  for (int k = 0; k < 2 * kShareKinds; ++k)
    if (t.shares[k].ncells == 0) {
      t.share_fallback = 0xffff;  // no batch-major tables at all
      t.shares.clear();
      return;
    }
}

int pick_share_mult(const HostTables& t, int kind, int nblocks, int slots, int wg_waves) {
  // a wave's prologue (cell record, piece record, joints, first streams: dependent round trips) costs about a dozen steps
  constexpr long kPrologue = 12;
  const ShareTable& st = t.shares[kind];
  int best = 1;
  long best_cost = -1;
  for (int m = 1; m <= 16 && st.ncells % (m * wg_waves) == 0; m *= 2) {
    const long waves = (long)nblocks * (st.ncells / m), rounds = (waves + slots - 1) / slots;
    const long c = rounds * ((long)m * st.max_cost + kPrologue);
    if (best_cost < 0 || c < best_cost) {
      best_cost = c;
      best = m;
    }
  }
  return best;
}

std::string build_tables(const smplfit_model_desc& d, HostTables& t, bool* unsupported) {
  *unsupported = false;
  if (!d.v_template || !d.shapedirs || !d.posedirs || !d.weights || !d.J_template ||
      !d.J_shapedirs || !d.parents)
    return "smplfit_create: null model array";
  if (d.num_vertices <= 0 || d.num_joints <= 1 || d.num_betas <= 0)
    return "smplfit_create: bad model dimensions";
  const int V = d.num_vertices, J = d.num_joints;
  const int n_kid = d.enable_kid ? 1 : 0;
  if (n_kid && (!d.kid_shapedir || !d.kid_J_shapedir))
    return "smplfit_create: enable_kid without kid_shapedir / kid_J_shapedir";
  // The kernels are instantiated for 10 and 16 betas (+ the kid unknown): any other count is padded up with zero
  // shape directions, which the solve pins to zero with a unit ridge (sf::solve_stage); the caller sees its own
  // num_betas everywhere (reference: BodyModel(num_betas=...) accepts any count, bodymodel.py:60-76).
  // More than 16 betas (the reference takes any count, default: every column of the file, common.py:223, 381-385; its
  // _fit_shape_general, bodyfitter.py:1104-1319) or more than 8 non-zero skinning weights per vertex (the reference
  // blends with the dense (V, J) matrix, :1000-1003): the GENERAL path — the same stages with their scratch in global
  // memory and vertex kernels with run-time loops over the unknowns and the weights (kernels_gen.inc).  No padding there.
  const int nb = d.num_betas, nb_pad = nb <= 10 ? 10 : (nb <= 16 ? 16 : nb);
  bool general = nb > 16;
  const int n_pad = nb_pad - nb;
  if (nb_pad + n_kid > 1023) {  // (kGenMaxS of kernels_gen.inc)
    *unsupported = true;
    return "smplfit_create: more than 1023 shape unknowns";
  }
  const int S = nb_pad + n_kid;  // the kid blend shape is one more shape direction (the last)
  // shape directions with the kid column appended (bodyfitter.py:52-58, :1139-1149)
  std::vector<float> shapedirs_ext((size_t)V * 3 * S, 0.f), jshapedirs_ext((size_t)J * 3 * S, 0.f);
  for (size_t r = 0; r < (size_t)V * 3; ++r) {
    for (int s2 = 0; s2 < nb; ++s2) shapedirs_ext[r * S + s2] = d.shapedirs[r * nb + s2];
    if (n_kid) shapedirs_ext[r * S + S - 1] = d.kid_shapedir[r];
  }
  for (size_t r = 0; r < (size_t)J * 3; ++r) {
    for (int s2 = 0; s2 < nb; ++s2) jshapedirs_ext[r * S + s2] = d.J_shapedirs[r * nb + s2];
    if (n_kid) jshapedirs_ext[r * S + S - 1] = d.kid_J_shapedir[r];
  }
  const float* const shapedirs = shapedirs_ext.data();
  const float* const J_shapedirs = jshapedirs_ext.data();
  if (J > kMaxJoints) {
    *unsupported = true;
    return "smplfit_create: more than 64 joints is not supported";
  }
  if (nb < 2) {
    *unsupported = true;
    return "smplfit_create: num_betas < 2 is not supported";
  }
  t.V = V; t.J = J; t.S = S; t.P = 9 * (J - 1);
  t.n_kid = n_kid;
  t.n_pad = n_pad;
  t.Vp = round_up(V, kVertexPad);
  // The batch-major vertex kernels need one all-zero padding slot (their out-of-range steps run on it).
  // A vertex count that is a multiple of the padding (vertex subsets of 1024, 2048, ...) gets one more
  // tile when those kernels apply: measured at B = 16384, 1024 vertices 4.58 -> 4.79 M fits/s, 2048
  // vertices 3.10 -> 3.69 M (512: the wave-per-instance kernels stay ahead, 5.85 vs 5.65 M).
  if (t.Vp == V && bm_shape_count(S) && V >= 1024) t.Vp += kVertexPad;
  // models whose pose features do not fit the A-stationary GEMM (more than 24 joints): the tiled split-bf16 GEMM works on
  // 256-column tiles, so 3 Vp has to be a multiple of 256 (a vertex subset with an odd number of 128-vertex tiles would
  // otherwise fall back to the fp32 GEMM, three times slower)
  if (t.P + 1 > 208) t.Vp = round_up(t.Vp, 2 * kVertexPad);
  // at least one padding row: row P of posedirs holds v_template and the matching pose feature is 1, so
  // the GEMM needs no bias operand (and adds the template last, as the reference does, bodyfitter.py:913-916)
  t.Kp = round_up(t.P + 1, kGemmKPad);
  t.smpl_family = d.is_smpl_family != 0;
  if (t.smpl_family && J < 12) return "smplfit_create: smpl-family model with < 12 joints";

  // ---- kinematic tree: parents, levels (bodyfitter.py:181-192), children-and-self (:61-64) ----
  t.parents.assign(J, 0);
  for (int i = 1; i < J; ++i) {
    int p = d.parents[i];
    if (p < 0 || p >= i) return "smplfit_create: parents must satisfy 0 <= parents[i] < i";
    t.parents[i] = p;
  }
  std::vector<int> depth(J, 0);
  int max_depth = 0;
  for (int i = 1; i < J; ++i) {
    depth[i] = depth[t.parents[i]] + 1;
    max_depth = std::max(max_depth, depth[i]);
  }
  t.fk_js.clear();
  t.fk_level_start.assign(1, 0);
  for (int lv = 1; lv <= max_depth; ++lv) {
    for (int i = 0; i < J; ++i)
      if (depth[i] == lv) t.fk_js.push_back(i);
    t.fk_level_start.push_back((int)t.fk_js.size());
  }
  std::vector<std::vector<int>> cas(J);
  for (int i = 0; i < J; ++i) cas[i].push_back(i);
  for (int i = 1; i < J; ++i) cas[t.parents[i]].push_back(i);
  t.cas_start.assign(1, 0);
  t.cas_flat.clear();
  for (int i = 0; i < J; ++i) {
    for (int j : cas[i]) t.cas_flat.push_back(j);
    t.cas_start.push_back((int)t.cas_flat.size());
  }

  // ---- part buckets (:81-97), toe copies (:147-156), adjustable parts (:101-104) ----
  t.part_type.assign(J, kPartNone);
  t.toe_src.assign(J, -1);
  for (int i = 0; i < J; ++i) {
    if (t.smpl_family && (i == 10 || i == 11)) {
      t.toe_src[i] = i - 3;  // 10 <- 7, 11 <- 8
      continue;
    }
    int n = (int)cas[i].size();
    t.part_type[i] = n >= 3 ? kPartMulti : (n == 2 ? kPartBone : kPartLeaf);
  }
  t.adj_flag.assign(J, 0);
  if (t.smpl_family) {
    static const int adj[] = {1, 2, 4, 5, 7, 8, 16, 17, 18, 19};
    for (int a : adj)
      if (a < J) t.adj_flag[a] = 1;
    // level-batched refinement requires every adjustable part to hold the same number of joints
    // (bodyfitter.py:219-221); true for SMPL / SMPL-X / SMPL+H.
    int n0 = -1;
    for (int i = 0; i < J; ++i)
      if (t.adj_flag[i]) {
        if (n0 < 0) n0 = (int)cas[i].size();
        if ((int)cas[i].size() != n0) {
          *unsupported = true;
          return "smplfit_create: adjustable parts with differing joint counts (sequential "
                 "fallback of the reference, bodyfitter.py:1546-1595, is not implemented)";
        }
      }
  } else {
    *unsupported = true;
    return "smplfit_create: non-SMPL-family models (MANO/FLAME) use the reference's sequential "
           "refinement, which is not implemented";
  }
  t.adj_level_start.assign(1, 0);
  t.adj_parts.clear();
  t.adj_last_level = -1;
  for (int lv = 0; lv < t.num_levels(); ++lv) {
    for (int k = t.fk_level_start[lv]; k < t.fk_level_start[lv + 1]; ++k)
      if (t.adj_flag[t.fk_js[k]]) {
        t.adj_parts.push_back(t.fk_js[k]);
        t.adj_last_level = lv;
      }
    t.adj_level_start.push_back((int)t.adj_parts.size());
  }
  t.used_part.assign(J, 0);
  for (int i = 0; i < J; ++i)
    if (t.part_type[i] == kPartBone || t.part_type[i] == kPartLeaf || t.adj_flag[i])
      t.used_part[i] = 1;

  // ---- part assignment = argmax skinning weight, toes -> feet (:36-44) ----
  t.part_assignment.assign(V, 0);
  int max_nnz = 0;
  for (int v = 0; v < V; ++v) {
    const float* w = d.weights + (size_t)v * J;
    int best = 0, nnz = 0;
    for (int j = 0; j < J; ++j) {
      if (w[j] > w[best]) best = j;  // first maximum, like argmax
      if (w[j] != 0.f) ++nnz;
    }
    if (t.smpl_family && (best == 10 || best == 11)) best -= 3;
    t.part_assignment[v] = best;
    max_nnz = std::max(max_nnz, nnz);
  }
  if (max_nnz > 8) general = true;
  // (16 betas + the kid unknown + 5-8 weights per vertex: the one combination the wave-per-instance kernels are not
  // built for — their LDS tiles would not fit; before round 5 such a model was refused)
  if (max_nnz > 4 && S == 17) general = true;
  if (max_nnz > 64) {
    *unsupported = true;
    return "smplfit_create: more than 64 non-zero skinning weights per vertex";
  }
  t.general = general;
  t.KW = max_nnz <= 4 ? 4 : (max_nnz <= 8 ? 8 : round_up(max_nnz, 4));
  t.NJ = t.KW == 8 ? 8 : 4;

  // ---- sorted slots: used parts first (by part id, stable), then the rest ----
  std::vector<int> order(V);
  std::iota(order.begin(), order.end(), 0);
  auto key = [&](int v) {
    int p = t.part_assignment[v];
    return (t.used_part[p] ? 0 : kMaxJoints) + p;
  };
  // within a part: by the set of skinning joints, so that runs with few distinct joints are long
  // (vertex groups of the batch-major kernels)
  std::vector<uint64_t> jmask(V, 0);
  for (int v = 0; v < V; ++v)
    for (int j = 0; j < J; ++j)
      if (d.weights[(size_t)v * J + j] != 0.f) jmask[v] |= (uint64_t)1 << j;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
    const int ka = key(a), kb = key(b);
    return ka != kb ? ka < kb : jmask[a] < jmask[b];
  });
  const int Vp = t.Vp;
  t.perm.assign(Vp, -1);
  t.slot_part.assign(Vp, -1);
  t.n_used = 0;
  for (int i = 0; i < V; ++i) {
    t.perm[i] = order[i];
    t.slot_part[i] = t.part_assignment[order[i]];
    if (t.used_part[t.slot_part[i]]) t.n_used = i + 1;
  }
  t.segments.clear();
  for (int i = 0; i < t.n_used;) {
    int p = t.slot_part[i], e = i;
    while (e < t.n_used && t.slot_part[e] == p) ++e;
    for (int s = i; s < e; s += kTile) t.segments.push_back({s, std::min(kTile, e - s), p});
    i = e;
  }

  // ---- per-slot constants ----
  const int P = t.P;
  t.vt.assign((size_t)3 * Vp, 0.f);
  t.dm.assign((size_t)3 * Vp, 0.f);
  t.sd.assign((size_t)3 * S * Vp, 0.f);
  const int S4g = (S + 3) & ~3;  // (rows padded to a multiple of four floats: 16-byte loads of four unknowns)
  t.sdg.assign(general ? (size_t)Vp * 3 * S4g : 0, 0.f);  // general path: vertex-major (Vp, 3, S4) rows
  t.widx.assign((size_t)(t.KW / 4) * Vp, 0u);
  t.wval.assign((size_t)t.KW * Vp, 0.f);
  t.pdT.assign((size_t)t.Kp * 3 * Vp, 0.f);
  const int half_k = t.Kp / 2;
  auto kpos = [&](int p) { return (p & 1) * half_k + (p >> 1); };  // == sf::rp_pos
  t.wsum_dev = 0.f;
  for (int i = 0; i < V; ++i) {
    const int v = order[i];
    const float* w = d.weights + (size_t)v * J;
    float wsum = 0.f;  // blended identity rotation = sum of weights (bodymodel.py:297-306)
    int k = 0;
    for (int j = 0; j < J; ++j) {
      wsum += w[j];
      if (w[j] != 0.f) {
        t.widx[(size_t)(k / 4) * Vp + i] |= (uint32_t)j << (8 * (k % 4));
        t.wval[(size_t)k * Vp + i] = w[j];
        ++k;
      }
    }
    t.wsum_dev = std::max(t.wsum_dev, std::fabs(wsum - 1.f));
    // padded pairs keep weight 0 and point at the vertex's own part (an LDS address other lanes
    // of the wave already read -> broadcast)
    for (; k < t.KW; ++k)
      t.widx[(size_t)(k / 4) * Vp + i] |= (uint32_t)t.part_assignment[v] << (8 * (k % 4));
    for (int c = 0; c < 3; ++c) {
      const float vtc = d.v_template[(size_t)v * 3 + c];
      t.vt[(size_t)c * Vp + i] = vtc;
      const float* pd = d.posedirs + ((size_t)v * 3 + c) * P;
      // default mesh: v_posed at identity rotations (feature = vec(I) per joint), times sum(w)
      float acc = vtc;
      for (int p = 0; p < P; ++p) {
        t.pdT[(size_t)kpos(p) * 3 * Vp + (size_t)c * Vp + i] = pd[p];
        if (p % 9 == 0 || p % 9 == 4 || p % 9 == 8) acc += pd[p];
      }
      t.pdT[(size_t)kpos(P) * 3 * Vp + (size_t)c * Vp + i] = vtc;  // bias row (feature P == 1)
      t.dm[(size_t)c * Vp + i] = wsum * acc;
      for (int s = 0; s < S; ++s) {
        t.sd[(size_t)(c * S + s) * Vp + i] = shapedirs[((size_t)v * 3 + c) * S + s];
        if (general) t.sdg[((size_t)i * 3 + c) * S4g + s] = shapedirs[((size_t)v * 3 + c) * S + s];
      }
    }
  }
  // part-aligned tiles over ALL slots (the general path's vertex passes that visit every vertex)
  t.segments_all.clear();
  for (int i = 0; i < V;) {
    int p = t.slot_part[i], e = i;
    while (e < V && t.slot_part[e] == p) ++e;
    for (int s0 = i; s0 < e; s0 += kTile) t.segments_all.push_back({s0, std::min(kTile, e - s0), p});
    i = e;
  }
  {
    const int N = 3 * Vp, Kp = t.Kp, ntile = N / 32;
    t.pdSw.assign((size_t)ntile * 32 * Kp, 0.f);
    for (int nt = 0; nt < ntile; ++nt)
      for (int n = 0; n < 32; ++n)
        for (int k = 0; k < Kp; ++k)
          t.pdSw[((size_t)nt * 32 + n) * Kp + k] = t.pdT[(size_t)k * N + nt * 32 + n];
  }
  t.pdB.clear();
  if (t.Kp == 208) {  // split-bf16 planes for k_posedirs_gemm_bf16x3 (error-free 3-way split of every fp32)
    const int N = 3 * Vp, Kp = t.Kp, ntile = N / 32, nslot = Kp / 8;
    auto bf16_rne = [](float x) -> uint16_t {
      uint32_t u;
      std::memcpy(&u, &x, 4);
      u += 0x7fffu + ((u >> 16) & 1u);  // round to nearest even (finite inputs)
      return (uint16_t)(u >> 16);
    };
    auto bf16_f32 = [](uint16_t h) -> float {
      const uint32_t u = (uint32_t)h << 16;
      float f;
      std::memcpy(&f, &u, 4);
      return f;
    };
    // elements of one tile image: three full planes, or (three products) two full planes + the lo plane of the last
    // k-step, [32 n][2 slots][8 k]
    const size_t plane = (size_t)32 * Kp, tile = kGemm3 ? 2 * plane + 32 * 16 : 3 * plane;
    t.pdB.assign((size_t)ntile * tile, 0);
    for (int nt = 0; nt < ntile; ++nt)
      for (int n = 0; n < 32; ++n)
        for (int k = 0; k < Kp; ++k) {
          const float x = t.pdSw[((size_t)nt * 32 + n) * Kp + k];
          const uint16_t h = bf16_rne(x);
          const float r1 = x - bf16_f32(h);
          const uint16_t m = bf16_rne(r1);
          const uint16_t l = bf16_rne(r1 - bf16_f32(m));
          const int slot = (k >> 3) ^ ((n >> 3) & 1);
          const size_t base = (size_t)nt * tile + (size_t)n * Kp + (size_t)slot * 8 + (k & 7);
          t.pdB[base] = h;
          t.pdB[base + plane] = m;
          if (!kGemm3) t.pdB[base + 2 * plane] = l;
          else if (k >= Kp - 16) t.pdB[(size_t)nt * tile + 2 * plane + (size_t)n * 16 + ((k >> 3) & 1) * 8 + (k & 7)] = l;
        }
    (void)nslot;
  }
  t.pdB2.clear();  // built on demand for the device (build_tiled_gemm_images)
  t.kc32 = 0;
  t.cpackA.clear(); t.cpackB.clear(); t.brec.clear(); t.vpieces.clear(); t.shares.clear();
  if (!general) {
    const int cs = t.cstride();
    auto pack = [&](float* dst, int slot) {  // one vertex record
      for (int c = 0; c < 3; ++c)
        for (int s2 = 0; s2 < S; ++s2) dst[s2 * 3 + c] = t.sd[(size_t)(c * S + s2) * Vp + slot];
      for (int k = 0; k < t.KW; ++k) dst[3 * S + k] = t.wval[(size_t)k * Vp + slot];
      for (int q = 0; q < t.KW / 4; ++q) {
        const uint32_t u = t.widx[(size_t)q * Vp + slot];
        std::memcpy(dst + 3 * S + t.KW + q, &u, 4);
      }
    };
    t.cpackA.assign((size_t)Vp * cs, 0.f);
    for (int i = 0; i < Vp; ++i) pack(t.cpackA.data() + (size_t)i * cs, i);
    // batch-major tables: the pieces of the sorted slots, the per-slot records, the share tables
    const int bs = t.brec_stride(), bw = t.brec_w();
    t.brec.assign((size_t)Vp * bs, 0.f);
    t.vpieces.clear();
    for (int i = 0; i < V;) {  // pieces of at most NJ joints (4, or 8 for models with 5-8 weights per vertex)
      const int p = t.slot_part[i];
      uint64_t u = 0;
      int e = i;
      while (e < V && t.slot_part[e] == p) {
        const uint64_t u2 = u | jmask[t.perm[e]];
        if (__builtin_popcountll(u2) > t.NJ) break;
        u = u2;
        ++e;
      }
      if (e == i) return "smplfit_create: a vertex of the batch-major tables has more skinning joints than a piece holds";
      VertexPiece pc{};
      pc.start = i;
      pc.count = e - i;
      pc.part = p;
      pc.used = t.used_part[p] ? 1 : 0;
      for (int j = 0; j < J; ++j)
        if ((u >> j) & 1) pc.joints[pc.nj++] = j;
      if (pc.nj == 0) pc.joints[pc.nj++] = p;  // (a vertex without weights: cannot happen for a valid model)
      for (int k = pc.nj; k < 8; ++k) pc.joints[k] = pc.joints[0];  // padding joints carry weight 0
      for (int v = i; v < e; ++v) {  // the vertex's weights in the piece's joint order
        float* r = t.brec.data() + (size_t)v * bs;
        for (int k = 0; k < pc.nj; ++k) r[bw + k] = d.weights[(size_t)t.perm[v] * J + pc.joints[k]];
      }
      t.vpieces.push_back(pc);
      i = e;
    }
    for (int s = 0; s < V; ++s) {
      float* rec = t.brec.data() + (size_t)s * bs;
      for (int s2 = 0; s2 < S; ++s2)
        for (int c = 0; c < 3; ++c) rec[c * S + s2] = t.sd[(size_t)(c * S + s2) * Vp + s];
    }
    build_share_tables(t);
    if (std::getenv("SMPLFIT_DUMP_SHARES"))
      for (size_t k = 0; k < t.shares.size(); ++k)
        std::fprintf(stderr, "cells %d kind %zu: pieces %zu rows %d max_cost %d\n", t.shares[k].ncells, k,
                     t.shares[k].pieces.size() / kPieceRec - 1, t.shares[k].nrows, t.shares[k].max_cost);
    t.cpackB.assign(t.segments.size() * 64 * cs, 0.f);
    for (size_t sgi = 0; sgi < t.segments.size(); ++sgi)
      for (int l = 0; l < t.segments[sgi].count; ++l)
        pack(t.cpackB.data() + (sgi * 64 + l) * cs, t.segments[sgi].start + l);
  }

  // ---- per-joint constants (:52-58, :191-192) ----
  const int S1 = S + 1;
  t.j_ext.assign((size_t)J * 3 * S1, 0.f);
  t.bone_ext.assign((size_t)J * 3 * S1, 0.f);
  for (int j = 0; j < J; ++j)
    for (int c = 0; c < 3; ++c) {
      t.j_ext[((size_t)j * 3 + c) * S1] = d.J_template[j * 3 + c];
      for (int s = 0; s < S; ++s)
        t.j_ext[((size_t)j * 3 + c) * S1 + 1 + s] = J_shapedirs[((size_t)j * 3 + c) * S + s];
    }
  for (int j = 0; j < J; ++j)
    for (int k = 0; k < 3 * S1; ++k)
      t.bone_ext[(size_t)j * 3 * S1 + k] =
          t.j_ext[(size_t)j * 3 * S1 + k] - t.j_ext[(size_t)t.parents[j] * 3 * S1 + k];

  t.fk_jp.resize(t.fk_js.size());
  t.bone_lv.assign(t.fk_js.size() * 3 * S1, 0.f);
  for (size_t q = 0; q < t.fk_js.size(); ++q) {
    const int j = t.fk_js[q];
    t.fk_jp[q] = j | (t.parents[j] << 16);
    for (int k = 0; k < 3 * S1; ++k) t.bone_lv[q * 3 * S1 + k] = t.bone_ext[(size_t)j * 3 * S1 + k];
  }

  // template-pass part sums of the reference side (s_a, s_w of _part_sums with a = default mesh)
  t.sa0.assign((size_t)J * 3, 0.f);
  t.sw0.assign(J, 0.f);
  {
    std::vector<double> acc((size_t)J * 3, 0.0);
    for (int i = 0; i < t.n_used; ++i) {
      int p = t.slot_part[i];
      for (int c = 0; c < 3; ++c) acc[p * 3 + c] += t.dm[(size_t)c * Vp + i];
      t.sw0[p] += 1.f;
    }
    for (size_t k = 0; k < acc.size(); ++k) t.sa0[k] = (float)acc[k];
  }

  // closed-form SA constants: sum over ALL vertices (the Gramian runs over every vertex)
  t.cs_joint.clear(); t.cw_joint.clear();
  if (!general) {
    std::vector<double> cs((size_t)J * 3 * S, 0.0), cw(J, 0.0);
    for (int v = 0; v < V; ++v)
      for (int j = 0; j < J; ++j) {
        const double w = d.weights[(size_t)v * J + j];
        if (w == 0.0) continue;
        cw[j] += w;
        for (int k = 0; k < 3 * S; ++k) cs[(size_t)j * 3 * S + k] += w * shapedirs[(size_t)v * 3 * S + k];
      }
    t.cs_joint.assign(cs.size(), 0.f);
    t.cw_joint.assign(J, 0.f);
    for (size_t k = 0; k < cs.size(); ++k) t.cs_joint[k] = (float)cs[k];
    for (int j = 0; j < J; ++j) t.cw_joint[j] = (float)cw[j];
  }

  // ---- pair-Gram constants (double accumulation, stored fp32) ----
  t.pair_j.clear(); t.pair_c1.clear(); t.pair_c2.clear(); t.pair_c3.clear(); t.diag_g0.clear(); t.diag_c2.clear();
  t.diag_c3.clear(); t.pair_E.clear(); t.jn_start.clear(); t.jn.clear(); t.pair_c2e.clear(); t.diag_c2e.clear();
  if (!general) {  // (O(pairs x 9 x S^2): built for the kernels that use them only)
    std::vector<int> pid((size_t)J * J, -1);
    std::vector<std::pair<int, int>> plist;
    for (int v = 0; v < V; ++v) {
      const float* w = d.weights + (size_t)v * J;
      for (int j = 0; j < J; ++j)
        if (w[j] != 0.f)
          for (int j2 = j + 1; j2 < J; ++j2)
            if (w[j2] != 0.f && pid[(size_t)j * J + j2] < 0) {
              pid[(size_t)j * J + j2] = 0;
            }
    }
    for (int j = 0; j < J; ++j)
      for (int j2 = j + 1; j2 < J; ++j2)
        if (pid[(size_t)j * J + j2] == 0) {
          pid[(size_t)j * J + j2] = (int)plist.size();
          plist.push_back({j, j2});
        }
    const int np = (int)plist.size();
    const int SS = S * S;
    std::vector<double> c1((size_t)np * 9 * SS, 0.0), c2((size_t)np * 3 * S, 0.0), c3(np, 0.0);
    std::vector<double> g0(SS, 0.0), dc2((size_t)J * 3 * S, 0.0), dc3(J, 0.0);
    std::vector<double> outer((size_t)9 * SS);
    for (int v = 0; v < V; ++v) {
      const float* w = d.weights + (size_t)v * J;
      const float* sv = shapedirs + (size_t)v * 3 * S;  // [a][i]
      bool have_outer = false;
      for (int j = 0; j < J; ++j) {
        if (w[j] == 0.f) continue;
        if (!have_outer) {
          for (int a = 0; a < 3; ++a)
            for (int a2 = 0; a2 < 3; ++a2)
              for (int i = 0; i < S; ++i)
                for (int i2 = 0; i2 < S; ++i2)
                  outer[((size_t)(a * 3 + a2) * S + i) * S + i2] = (double)sv[a * S + i] * (double)sv[a2 * S + i2];
          have_outer = true;
        }
        const double wjj = (double)w[j] * (double)w[j];
        for (int a = 0; a < 3; ++a)
          for (int k = 0; k < SS; ++k) g0[k] += wjj * outer[(size_t)(a * 3 + a) * SS + k];
        for (int k = 0; k < 3 * S; ++k) dc2[(size_t)j * 3 * S + k] += wjj * (double)sv[k];
        dc3[j] += wjj;
        for (int j2 = j + 1; j2 < J; ++j2) {
          if (w[j2] == 0.f) continue;
          const int p = pid[(size_t)j * J + j2];
          const double ww = (double)w[j] * (double)w[j2];
          double* dst = c1.data() + (size_t)p * 9 * SS;
          for (size_t k = 0; k < (size_t)9 * SS; ++k) dst[k] += ww * outer[k];
          for (int k = 0; k < 3 * S; ++k) c2[(size_t)p * 3 * S + k] += ww * (double)sv[k];
          c3[p] += ww;
        }
      }
    }
    auto tof = [](const std::vector<double>& a, std::vector<float>& o) {
      o.resize(a.size());
      for (size_t k = 0; k < a.size(); ++k) o[k] = (float)a[k];
    };
    t.pair_j.clear();
    for (auto& pr : plist) {
      t.pair_j.push_back(pr.first);
      t.pair_j.push_back(pr.second);
    }
    tof(c1, t.pair_c1); tof(c2, t.pair_c2); tof(c3, t.pair_c3);
    const int SE = t.s_even();
    // symmetrised pair constants of the batch-major pair-Gram kernel: entry e = (i <= i2) of the upper triangle in
    // row-major order, E[p][aa'][e] = c1[aa'][i][i2] + c1[aa'][i2][i] (the Gramian collects f[i][i2] + f[i2][i])
    const int NGP = t.ng_pad();
    t.pair_E.assign((size_t)np * 9 * NGP, 0.f);
    for (int p = 0; p < np; ++p)
      for (int aa = 0; aa < 9; ++aa) {
        int e = 0;
        for (int i = 0; i < S; ++i)
          for (int i2 = i; i2 < S; ++i2, ++e)
            t.pair_E[((size_t)p * 9 + aa) * NGP + e] = t.pair_c1[(((size_t)p * 9 + aa) * S + i) * S + i2] +
                                                      t.pair_c1[(((size_t)p * 9 + aa) * S + i2) * S + i];
      }
    // neighbours of every joint: (other joint, pair) for each pair the joint is part of
    t.jn_start.assign(J + 1, 0);
    t.jn.clear();
    for (int j = 0; j < J; ++j) {
      for (int p = 0; p < np; ++p) {
        const int j1 = t.pair_j[2 * p], j2 = t.pair_j[2 * p + 1];
        if (j1 == j || j2 == j) {
          t.jn.push_back(j1 == j ? j2 : j1);
          t.jn.push_back(p);
        }
      }
      t.jn_start[j + 1] = (int32_t)(t.jn.size() / 2);
    }
    t.pair_c2e.assign((size_t)np * 3 * SE, 0.f);
    for (int p = 0; p < np; ++p)
      for (int a = 0; a < 3; ++a)
        for (int i = 0; i < S; ++i) t.pair_c2e[((size_t)p * 3 + a) * SE + i] = t.pair_c2[((size_t)p * 3 + a) * S + i];
    tof(g0, t.diag_g0); tof(dc2, t.diag_c2); tof(dc3, t.diag_c3);
    t.diag_c2e.assign((size_t)J * 3 * SE, 0.f);
    for (int j = 0; j < J; ++j)
      for (int a = 0; a < 3; ++a)
        for (int i = 0; i < S; ++i) t.diag_c2e[((size_t)j * 3 + a) * SE + i] = t.diag_c2[((size_t)j * 3 + a) * S + i];
  }

  // ---- tiles of the residual kernel: part-aligned over all slots, <= 16 distinct joints ----
  t.gtiles.clear();
  t.gblob.clear();
  if (!general) {
    for (int i = 0; i < V;) {
      const int p = t.slot_part[i];
      int e = i;
      uint64_t jset = 0;
      int njoint = 0;
      while (e < V && e - i < kTile && t.slot_part[e] == p) {
        const float* w = d.weights + (size_t)t.perm[e] * J;
        uint64_t add = 0;
        for (int j = 0; j < J; ++j)
          if (w[j] != 0.f && !((jset >> j) & 1)) add |= (uint64_t)1 << j;
        const int nadd = __builtin_popcountll(add);
        if (njoint + nadd > 16) break;
        jset |= add;
        njoint += nadd;
        ++e;
      }
      if (e == i) return "smplfit_create: a vertex has more than 16 skinning joints";
      t.gtiles.push_back({i, e - i, p});
      i = e;
    }
    const int cs = t.cstride(), gs = t.gblob_stride();
    t.gblob.assign(t.gtiles.size() * (size_t)gs, 0.f);
    for (size_t g = 0; g < t.gtiles.size(); ++g) {
      float* blob = t.gblob.data() + g * gs;
      const Segment& sg = t.gtiles[g];
      int slot_of_joint[kMaxJoints];
      for (int j = 0; j < J; ++j) slot_of_joint[j] = -1;
      int nslots = 0;
      int32_t slots[16];
      for (int k = 0; k < 16; ++k) slots[k] = J;  // padding -> dummy bin
      for (int l = 0; l < sg.count; ++l) {
        std::memcpy(blob + (size_t)l * cs, t.cpackA.data() + (size_t)(sg.start + l) * cs, sizeof(float) * cs);
        const float* w = d.weights + (size_t)t.perm[sg.start + l] * J;
        for (int j = 0; j < J; ++j)
          if (w[j] != 0.f && slot_of_joint[j] < 0) {
            slot_of_joint[j] = nslots;
            slots[nslots++] = j;
          }
      }
      // padded lanes keep zero records; give them a valid joint index word (joint of the part)
      for (int l = sg.count; l < 64; ++l) {
        uint32_t u = 0;
        for (int k = 0; k < 4; ++k) u |= (uint32_t)sg.part << (8 * k);
        for (int q = 0; q < t.KW / 4; ++q) std::memcpy(blob + (size_t)l * cs + 3 * S + t.KW + q, &u, 4);
      }
      float* wA = blob + 64 * cs;  // [16 steps][64 lanes]
      for (int step = 0; step < 16; ++step)
        for (int l = 0; l < 64; ++l) {
          const int vtx = 4 * step + l / 16, slot = l % 16;
          float val = 0.f;
          if (vtx < sg.count && slots[slot] < J)
            val = d.weights[(size_t)t.perm[sg.start + vtx] * J + slots[slot]];
          wA[step * 64 + l] = val;
        }
      std::memcpy(blob + 64 * cs + 16 * 64, slots, sizeof(slots));
    }
  }

  // ---- sparse joint regressor over sorted slots ----
  t.has_regressor = false;
  t.reg_start.assign(J + 1, 0);  // (no regressor: J empty rows, so that a stray reader finds valid loop bounds)
  t.reg_slot.clear();
  t.reg_val.clear();
  t.reg_rowsum.assign(J, 0.f);
  if (d.J_regressor_post_lbs && d.regressor_num_vertices == V) {
    t.reg_start.assign(1, 0);
    std::vector<int> slot_of(V);
    for (int i = 0; i < V; ++i) slot_of[order[i]] = i;
    for (int j = 0; j < J; ++j) {
      std::vector<std::pair<int, float>> row;
      for (int v = 0; v < V; ++v) {
        float r = d.J_regressor_post_lbs[(size_t)j * V + v];
        if (r != 0.f) row.push_back({slot_of[v], r});
      }
      std::sort(row.begin(), row.end());
      for (auto& e : row) {
        t.reg_slot.push_back(e.first);
        t.reg_val.push_back(e.second);
        t.reg_rowsum[j] += e.second;
      }
      t.reg_start.push_back((int)t.reg_slot.size());
    }
    t.has_regressor = true;
    // the slots the regressor reads, flagged in the spare float behind the shapedirs of their batch-major record: a
    // joints-omitted fit writes the posed vertex back only there (k_lbs_partsum_bm, WRITE_V)
    if (!t.brec.empty() && t.brec_w() > 3 * S)
      for (int slot : t.reg_slot) t.brec[(size_t)slot * t.brec_stride() + 3 * S] = 1.f;
  }
  return "";
}


// Stage images of posedirs for k_posedirs_gemm_bf16x3_tiled (models with Kp != 208): 94 MB for SMPL-X, so they
// are built only for a handle that uploads to a device, not by the host-only table builds of the tests.
void build_tiled_gemm_images(HostTables& t) {
  const int Vp = t.Vp;
  t.pdB2.clear();
  t.kc32 = 0;
  if (kGemm3 && t.Kp != 208 && (3 * Vp) % 256 == 0) {  // (the kernel is the three-product form, on 256-column tiles)
    const int N = 3 * Vp, Kp = t.Kp, nt128 = N / 128, kc32 = (Kp + 31) / 32;
    auto bf16_rne = [](float x) -> uint16_t {
      uint32_t u;
      std::memcpy(&u, &x, 4);
      u += 0x7fffu + ((u >> 16) & 1u);
      return (uint16_t)(u >> 16);
    };
    auto bf16_f32 = [](uint16_t h) -> float {
      const uint32_t u = (uint32_t)h << 16;
      float f;
      std::memcpy(&f, &u, 4);
      return f;
    };
    t.kc32 = kc32;
    const size_t plane = (size_t)128 * 32;  // elements of one plane of a stage
    t.pdB2.assign((size_t)nt128 * kc32 * 3 * plane, 0);
    for (int nt = 0; nt < nt128; ++nt)
      for (int n = 0; n < 128; ++n)
        for (int k = 0; k < Kp; ++k) {
          const int col = nt * 128 + n;
          const float x = t.pdSw[((size_t)(col / 32) * 32 + col % 32) * Kp + k];
          const uint16_t h = bf16_rne(x);
          const float r1 = x - bf16_f32(h);
          const uint16_t m = bf16_rne(r1);
          const uint16_t l = bf16_rne(r1 - bf16_f32(m));
          const int kc = k / 32, kk = k % 32, slot = (kk >> 3) ^ ((n >> 2) & 3);
          const size_t base = ((size_t)nt * kc32 + kc) * 3 * plane + (size_t)n * 32 + slot * 8 + (kk & 7);
          t.pdB2[base] = h;
          t.pdB2[base + plane] = m;
          t.pdB2[base + 2 * plane] = l;
        }
  }
}

}  // namespace sf
