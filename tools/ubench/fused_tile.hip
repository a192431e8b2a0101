// Stage (a) of the fused posedirs-GEMM + residual pass: can a workgroup form a
// (2 instance blocks x 32 slots) tile of v_posed on the bf16 matrix cores, leave it in LDS and run the residual
// pass's per-vertex step on it — instead of the GEMM writing 340 MB that the residual pass reads back?
//
// Form measured here (the one DESIGN.md §8.1 arrives at: the A-stationary operand reuse has to survive the fusion):
// a workgroup of 12 waves = 4 PRODUCERS (a 32-instance half of one of two 64-instance blocks each: the library's
// k_posedirs_gemm_bf16x3 inner loop — features resident as split-bf16 fragments, posedirs tile images through a
// 2-slot LDS ring filled by LDS-DMA, three products per k-step + the bias step's third term) and 8 CONSUMERS (four
// per block, 8 of a tile's 32 slots each: the arithmetic of k_residual_bm's step — blend of the piece's four joints
// held in registers, residual, R~^T b, r1 += S_v^T u, four moments — with v_posed read from the LDS tile, lane =
// instance, and the targets streamed from HBM one tile ahead).  Everything is synchronised by the workgroup barrier:
// one barrier per tile image (three per 32-slot tile), consumers working on tile t - 1 while the producers form
// tile t in the other half of a double-buffered LDS tile (98 KB) — no flags, no polling.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/fused_tile.hip -o /tmp/fused_tile && /tmp/fused_tile
//
// Prints the time of the fused kernel at B = 4096, Vp = 6912, K = 208 (the c2 shapes) in its variants, checks the
// v_posed values the consumers see against a plain evaluation of the same split products, and — for calibration on
// the same box — the time of the same producer loop writing v_posed to HBM and of the same consumer loop reading it
// from there (the unfused pair: what the library's K2' + K3' do in 112 + 127 us).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int KS = 13, KP = 16 * KS;                 // 208 pose features (incl. the bias row)
constexpr int kPlane = 32 * KP * 2;                  // 13312 B: one bf16 plane of a 32-column tile image
constexpr int kImg = 2 * kPlane + 32 * 32;           // 27648 B: planes hi, mid + the lo plane of the last k-step
constexpr int NDMA = kImg / 1024;                    // 27 LDS-DMA instructions of 1 KB per image
constexpr int S = 10, BREC = 36;                     // vertex record: 30 shapedirs values, 2 pad, 4 weights (brec_stride)
constexpr int JROW = 12;                             // joint record: R (9) | T0 (3), instance-innermost rows
constexpr int NPROD = 4, NCONS = 8, NWAVE = NPROD + NCONS;
constexpr int kTileFloats = 2 * 3 * 32 * 64;         // one v_posed tile: [block 2][coord 3][slot 32][instance 64]
constexpr size_t kLds = 2 * (size_t)kImg + 2 * (size_t)kTileFloats * 4;  // 55296 + 98304 = 153600 B

__device__ __forceinline__ f2 mk2(float a, float b) { f2 r; r.x = a; r.y = b; return r; }
typedef const __attribute__((address_space(4))) float* const_f32_ptr;
__device__ __forceinline__ const_f32_ptr as_constant(const float* p) { return (const_f32_ptr)(unsigned long long)p; }

__device__ __forceinline__ void lds_dma16(const char* src, uint32_t lane_off, char* lds) {
  const uint32_t l = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(l), "v"(lane_off), "s"(src) : "memory", "m0");
}

// the split features of this lane's instance (its k-octet of every k-step): two bf16 planes (three-product form)
struct Features { bf16x8 f1[KS], f2[KS]; };
__device__ __forceinline__ void load_features(const float* A, int inst, int kg, Features& F) {
  const float* src = A + (size_t)inst * KP + kg * 8;
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const float4 lo = *reinterpret_cast<const float4*>(src + 16 * s);
    const float4 hi = *reinterpret_cast<const float4*>(src + 16 * s + 4);
    const float x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const __bf16 h = (__bf16)x[j];
      F.f1[s][j] = h;
      F.f2[s][j] = (__bf16)(x[j] - (float)h);
    }
  }
}

// one 32 x 32 tile of v_posed: rows = the image's 32 columns (slots of one coordinate), columns = 32 instances
__device__ __forceinline__ f32x16 tile_product(const char* img, int l31, int kg, const Features& F) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const char* base = img + l31 * (KP * 2) + ((kg ^ ((l31 >> 3) & 1)) * 16);
  auto frag = [&](int plane, int s) {
    if (plane == 2) return *reinterpret_cast<const bf16x8*>(img + 2 * kPlane + l31 * 32 + kg * 16);
    return *reinterpret_cast<const bf16x8*>(base + plane * kPlane + s * 32);
  };
  bf16x8 p1 = frag(0, 0), p2 = frag(1, 0), p3 = p1;
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    bf16x8 n1, n2, n3;
    if (s + 1 < KS) {
      n1 = frag(0, s + 1);
      n2 = frag(1, s + 1);
      n3 = n1;
      if (s + 1 == KS - 1) n3 = frag(2, 0);
    }
    if (s == KS - 1) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p3, F.f1[s], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p2, F.f1[s], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p1, F.f2[s], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p1, F.f1[s], acc, 0, 0, 0);
    if (s + 1 < KS) { p1 = n1; p2 = n2; p3 = n3; }
  }
  return acc;
}

// LDS column of (slot, instance): the two 32-instance halves are swapped on slots with bit 2 set, so that the two lane
// halves of a producer's store (slots s and s + 4, the same 32 instances) land on disjoint banks; a consumer (lane =
// instance, one slot) still reads 64 consecutive banks
__device__ __forceinline__ int tile_col(int slot, int inst) { return slot * 64 + (inst ^ (((slot >> 2) & 1) << 5)); }

struct Args {
  const float* A;        // (Mp, 208) pose features
  const char* img;       // (3 Vp / 32) tile images, image of (coordinate c, slot tile t) = c * (Vp / 32) + t
  const float* tT;       // (Mp / 64, 3 Vp, 64) targets, instance-innermost
  float* vpT;            // (Mp / 64, 3 Vp, 64) v_posed (unfused variants, and WRITE_VPT)
  const float* jdT;      // (Mp / 64, J * 12, 64) joint records
  const float* brec;     // (Vp, 36) vertex records
  float* out;            // per consumer wave: 22 sums x 64 lanes
  float* dump;           // CHECK: the v_posed values the consumers read, (Mp / 64, 3 Vp, 64)
  int Vp, J, tiles_per_wg, jperiod;
};

// the residual step of one vertex (k_residual_bm::step): joints of the piece in registers, record through scalar loads
struct Piece { f2 p[4][6]; };
struct Sums { f2 r1p[5], m01[4]; float m2[4]; };
struct Rec { float sd[3 * S], cw[4]; };
__device__ __forceinline__ void load_rec(const float* brec, int v, Rec& r) {
  const const_f32_ptr rec = as_constant(brec + (size_t)v * BREC);
#pragma unroll
  for (int k = 0; k < 3 * S; ++k) r.sd[k] = rec[k];
#pragma unroll
  for (int k = 0; k < 4; ++k) r.cw[k] = rec[32 + k];
}
__device__ __forceinline__ void vertex_step(const Piece& jr, const Rec& rc, float x0, float x1, float x2, float t0, float t1,
                                            float t2, Sums& a) {
  const float (&sd)[3 * S] = rc.sd;
  const float (&cw)[4] = rc.cw;
  f2 Q[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) Q[i] = mk2(0, 0);
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int i = 0; i < 6; ++i) Q[i] += cw[k] * jr.p[k][i];
  const f2 pos01 = (Q[0] * x0 + Q[1] * x1) + (Q[2] * x2 + Q[5]);
  const float pos2 = (Q[3].x * x0 + Q[3].y * x1) + (Q[4].x * x2 + Q[4].y);
  const f2 b01 = mk2(t0, t1) - pos01;
  const float b2 = t2 - pos2;
  const float u0 = (Q[0].x * b01.x + Q[0].y * b01.y) + Q[3].x * b2;
  const float u1 = (Q[1].x * b01.x + Q[1].y * b01.y) + Q[3].y * b2;
  const float u2 = (Q[2].x * b01.x + Q[2].y * b01.y) + Q[4].x * b2;
#pragma unroll
  for (int k = 0; k < 5; ++k)
    a.r1p[k] += (mk2(sd[2 * k], sd[2 * k + 1]) * u0 + mk2(sd[S + 2 * k], sd[S + 2 * k + 1]) * u1) + mk2(sd[2 * S + 2 * k], sd[2 * S + 2 * k + 1]) * u2;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    a.m01[k] += cw[k] * b01;
    a.m2[k] += cw[k] * b2;
  }
}
__device__ __forceinline__ void load_piece(const float* jdb, int j0, int J, Piece& jr) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float* row = jdb + (size_t)((j0 + k) % J) * JROW * 64;
    float r[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) r[i] = row[i * 64];
    jr.p[k][0] = mk2(r[0], r[3]); jr.p[k][1] = mk2(r[1], r[4]); jr.p[k][2] = mk2(r[2], r[5]);
    jr.p[k][3] = mk2(r[6], r[7]); jr.p[k][4] = mk2(r[8], r[11]); jr.p[k][5] = mk2(r[9], r[10]);
  }
}
__device__ __forceinline__ void store_sums(float* out, size_t wave_id, int lane, const Sums& a) {
  float* o = out + wave_id * 22 * 64 + lane;
#pragma unroll
  for (int k = 0; k < 5; ++k) { o[(2 * k) * 64] = a.r1p[k].x; o[(2 * k + 1) * 64] = a.r1p[k].y; }
#pragma unroll
  for (int k = 0; k < 4; ++k) { o[(10 + 3 * k) * 64] = a.m01[k].x; o[(11 + 3 * k) * 64] = a.m01[k].y; o[(12 + 3 * k) * 64] = a.m2[k]; }
}

// ------------------------------------------------------------------------------------------------------------------
// the fused kernel.  grid (chunks of slot tiles, Mp / 128), block 768.
// WRITE_VPT: the producers also write v_posed to HBM (what a stand-alone LBS pass would still read).
// CHECK: the consumers dump the v_posed values they read.
// ------------------------------------------------------------------------------------------------------------------
// ABL (timing only): 1 the consumers do nothing but the barriers (producer side alone), 2 the producers skip their
// products and stores (consumer side alone), 4 the consumers skip the LDS reads of v_posed
template <bool WRITE_VPT, bool CHECK, int ABL = 0>
__global__ __launch_bounds__(64 * NWAVE) void k_fused(Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ring = smem;                                            // [2][kImg]
  float* tile = reinterpret_cast<float*>(smem + 2 * kImg);      // [2][kTileFloats]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Vp = a.Vp, ntile = Vp / 32;
  const int t0 = blockIdx.x * a.tiles_per_wg, t1 = min(t0 + a.tiles_per_wg, ntile), nt = t1 - t0;
  if (nt <= 0) return;
  const int blk0 = blockIdx.y * 2;
  const bool producer = wave < NPROD;
  auto image_of = [&](int u) {  // sub-step u of this workgroup: tile t0 + u / 3, coordinate u % 3
    return a.img + (size_t)((u % 3) * ntile + t0 + u / 3) * kImg;
  };
  auto dma_image = [&](int u) {  // the producers copy the image of sub-step u into ring slot u & 1 (7 / 7 / 7 / 6 chunks)
    const char* src = image_of(u);
    char* dst = ring + (u & 1) * kImg;
#pragma unroll
    for (int i = 0; i < (NDMA + NPROD - 1) / NPROD; ++i) {
      const int c = wave + NPROD * i;
      if (c < NDMA) lds_dma16(src + c * 1024, lane * 16, dst + c * 1024);
    }
  };
  if (producer) {
    // ---------------- producers ----------------
    const int l31 = lane & 31, kg = lane >> 5, pblk = wave >> 1, half = wave & 1;
    Features F;
    load_features(a.A, (blk0 + pblk) * 64 + half * 32 + l31, kg, F);
    dma_image(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int it = 0; it <= nt; ++it) {
#pragma unroll 1
      for (int c = 0; c < 3; ++c) {
        const int u = it * 3 + c;
        if (it < nt && !(ABL & 2)) {
          if (u + 1 < 3 * nt) dma_image(u + 1);
          const f32x16 acc = tile_product(ring + (u & 1) * kImg, l31, kg, F);
          float* dst = tile + (it & 1) * kTileFloats + (pblk * 3 + c) * 32 * 64;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int slot = (r & 3) + 8 * (r >> 2) + 4 * kg;
            dst[tile_col(slot, half * 32 + l31)] = acc[r];
          }
          if constexpr (WRITE_VPT) {
            float* g = a.vpT + ((size_t)(blk0 + pblk) * 3 * Vp + (size_t)c * Vp + (size_t)(t0 + it) * 32) * 64 + half * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) __builtin_nontemporal_store(acc[r], g + (size_t)((r & 3) + 8 * (r >> 2) + 4 * kg) * 64);
          }
          // the next image has landed (this wave's chunks), the tile rows are written
          if constexpr (WRITE_VPT) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
      }
    }
    if constexpr (WRITE_VPT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }
  // ---------------- consumers ----------------
  const int ci = wave - NPROD, cblk = ci >> 2, q = ci & 3;  // block of the pair, 8-slot run of every tile
  const int blk = blk0 + cblk;
  const float* tp = a.tT + (size_t)blk * 3 * Vp * 64 + lane;
  const float* jdb = a.jdT + (size_t)blk * a.J * JROW * 64 + lane;
  const size_t cstr = (size_t)Vp * 64;
  Sums acc;
#pragma unroll
  for (int k = 0; k < 5; ++k) acc.r1p[k] = mk2(0, 0);
#pragma unroll
  for (int k = 0; k < 4; ++k) { acc.m01[k] = mk2(0, 0); acc.m2[k] = 0.f; }
  Piece jr;
  // targets of a tile's 8 slots, requested a whole tile ahead: ONE register set, a slot's three registers are refilled
  // with the same slot of the next tile as soon as the step has consumed them
  float tg[8][3];
  auto request_slot = [&](int t, int i) {
    const float* p = tp + (size_t)(t * 32 + q * 8 + i) * 64;
#pragma unroll
    for (int c = 0; c < 3; ++c) tg[i][c] = __builtin_nontemporal_load(p + (size_t)c * cstr);
  };
#pragma unroll
  for (int i = 0; i < 8; ++i) request_slot(t0, i);
  Rec recA, recB;  // the records of even / odd slots of the run, requested one slot ahead
  load_rec(a.brec, t0 * 32 + q * 8, recA);
  __builtin_amdgcn_s_barrier();  // (the producers' first image)
  for (int it = 0; it <= nt; ++it) {
    // it >= 1: tile t0 + it - 1 sits in tile buffer (it - 1) & 1
    const int t = t0 + it - 1;
    const float* src = tile + ((it - 1) & 1) * kTileFloats + cblk * 3 * 32 * 64;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      if (it >= 1 && !(ABL & 1)) {
        if (c == 0 && (it - 1) % a.jperiod == 0) load_piece(jdb, (t * 4 + q) % a.J, a.J, jr);
        const int i0 = c * 3, i1 = c == 2 ? 8 : i0 + 3;
#pragma unroll
        for (int i = i0; i < i1; ++i) {
          const int slot = q * 8 + i;
          // this slot's record was requested a step ago (scalar loads): take it, request the next slot's
          const Rec rc = i & 1 ? recB : recA;
          {
            const int nslot = i + 1 < 8 ? t * 32 + slot + 1 : (t + 1) * 32 + q * 8;
            asm volatile("" ::"s"(rc.cw[0]));
            __builtin_amdgcn_sched_barrier(0);
            if (i & 1) load_rec(a.brec, nslot < Vp ? nslot : Vp - 1, recA);
            else load_rec(a.brec, nslot < Vp ? nslot : Vp - 1, recB);
            __builtin_amdgcn_sched_barrier(0);
          }
          float x0 = 0.1f, x1 = 0.2f, x2 = 0.3f;
          if (!(ABL & 4)) {
            x0 = src[tile_col(slot, lane)]; x1 = src[32 * 64 + tile_col(slot, lane)]; x2 = src[2 * 32 * 64 + tile_col(slot, lane)];
          }
          if constexpr (CHECK) {
            float* d = a.dump + ((size_t)blk * 3 * Vp + (size_t)(t * 32 + slot)) * 64 + lane;
            d[0] = x0; d[cstr] = x1; d[2 * cstr] = x2;
          }
          vertex_step(jr, rc, x0, x1, x2, tg[i][0], tg[i][1], tg[i][2], acc);
          if (it < nt) request_slot(t + 1, i);
        }
      }
      __builtin_amdgcn_s_barrier();
    }
  }
  store_sums(a.out, ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * NCONS + ci, lane, acc);
}

// ------------------------------------------------------------------------------------------------------------------
// SYMMETRIC form: every wave plays both roles in turn.  A workgroup of 8 waves = 256 instances = four 64-instance blocks;
// per 32-slot tile: (A) every wave forms its 32 instances' three coordinate tiles (the library's GEMM loop, one barrier
// per image) into an LDS tile [4 blocks][3][32][64] (98 KB); (B) wave w runs the residual step on 16 of the tile's slots
// for block w / 2 (slots 16 (w & 1) ...), lane = instance; one more barrier closes the tile.  No MFMA / VALU overlap
// inside a workgroup (the phases alternate), but in phase B all eight waves — two per SIMD, 256 registers each: the
// occupancy of the library's eight-joint vertex kernels — do vertex work.  Targets of the next tile are requested before
// phase A (they arrive under the products), vertex records one slot ahead.  grid (chunks of tiles, Mp / 256).
// ------------------------------------------------------------------------------------------------------------------
constexpr int kSymTileFloats = 4 * 3 * 32 * 64;
constexpr size_t kSymLds = 2 * (size_t)kImg + (size_t)kSymTileFloats * 4;  // 55296 + 98304
template <bool WRITE_VPT, int ABL = 0>
__global__ __launch_bounds__(512, 1) void k_fused_sym(Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ring = smem;
  float* tile = reinterpret_cast<float*>(smem + 2 * kImg);
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kg = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Vp = a.Vp, ntile = Vp / 32;
  const int t0 = blockIdx.x * a.tiles_per_wg, t1 = min(t0 + a.tiles_per_wg, ntile), nt = t1 - t0;
  if (nt <= 0) return;
  const int blk0 = blockIdx.y * 4;
  // role A: instances (blk0 * 64 + wave * 32 ...), role B: block cblk, slots 16 h ... of every tile
  const int pblk = wave >> 1, half = wave & 1;
  const int blk = blk0 + pblk;
  Features F;
  load_features(a.A, blk0 * 64 + wave * 32 + l31, kg, F);
  auto dma_image = [&](int u) {  // image of sub-step u (tile t0 + u / 3, coordinate u % 3) into ring slot u & 1
    const char* src = a.img + (size_t)((u % 3) * ntile + t0 + u / 3) * kImg;
    char* dst = ring + (u & 1) * kImg;
#pragma unroll
    for (int i = 0; i < (NDMA + 7) / 8; ++i) {
      const int c = wave + 8 * i;
      if (c < NDMA) lds_dma16(src + c * 1024, lane * 16, dst + c * 1024);
    }
  };
  const float* tp = a.tT + (size_t)blk * 3 * Vp * 64 + lane;
  const float* jdb = a.jdT + (size_t)blk * a.J * JROW * 64 + lane;
  const size_t cstr = (size_t)Vp * 64;
  Sums acc;
#pragma unroll
  for (int k = 0; k < 5; ++k) acc.r1p[k] = mk2(0, 0);
#pragma unroll
  for (int k = 0; k < 4; ++k) { acc.m01[k] = mk2(0, 0); acc.m2[k] = 0.f; }
  Piece jr;
  float tg[16][3];  // targets of this wave's 16 slots of the CURRENT tile (requested before its phase A)
  auto request_tile = [&](int t) {
    const float* p = tp + (size_t)(t * 32 + half * 16) * 64;
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int c = 0; c < 3; ++c) tg[i][c] = __builtin_nontemporal_load(p + (size_t)c * cstr + (size_t)i * 64);
  };
  Rec recA, recB;
  dma_image(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int it = 0; it < nt; ++it) {
    const int t = t0 + it;
    if (!(ABL & 1)) request_tile(t);
    // ---- phase A: three coordinate tiles of this wave's 32 instances ----
#pragma unroll 1
    for (int c = 0; c < 3; ++c) {
      const int u = it * 3 + c;
      if (!(ABL & 2)) {
        if (u + 1 < 3 * nt) dma_image(u + 1);
        const f32x16 pr = tile_product(ring + (u & 1) * kImg, l31, kg, F);
        float* dst = tile + (pblk * 3 + c) * 32 * 64;
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[tile_col((r & 3) + 8 * (r >> 2) + 4 * kg, half * 32 + l31)] = pr[r];
        if constexpr (WRITE_VPT) {
          float* g = a.vpT + ((size_t)blk * 3 * Vp + (size_t)c * Vp + (size_t)t * 32) * 64 + half * 32 + l31;
#pragma unroll
          for (int r = 0; r < 16; ++r) __builtin_nontemporal_store(pr[r], g + (size_t)((r & 3) + 8 * (r >> 2) + 4 * kg) * 64);
        }
      }
      // this wave's chunks of the next image have landed; the tile rows are written.  (The targets requested above are
      // older than the DMA: a counted wait would have to keep them out — they have had a whole product to arrive.)
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    // ---- phase B: 16 slots of block pblk ----
    if (!(ABL & 1)) {
      if (it % a.jperiod == 0) load_piece(jdb, (t * 2 + half) % a.J, a.J, jr);
      const float* src = tile + pblk * 3 * 32 * 64;
      load_rec(a.brec, t * 32 + half * 16, recA);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int slot = half * 16 + i;
        const Rec rc = i & 1 ? recB : recA;
        if (i + 1 < 16) {
          asm volatile("" ::"s"(rc.cw[0]));
          __builtin_amdgcn_sched_barrier(0);
          if (i & 1) load_rec(a.brec, t * 32 + slot + 1, recA);
          else load_rec(a.brec, t * 32 + slot + 1, recB);
          __builtin_amdgcn_sched_barrier(0);
        }
        const float x0 = src[tile_col(slot, lane)], x1 = src[32 * 64 + tile_col(slot, lane)], x2 = src[2 * 32 * 64 + tile_col(slot, lane)];
        vertex_step(jr, rc, x0, x1, x2, tg[i][0], tg[i][1], tg[i][2], acc);
      }
    }
    __builtin_amdgcn_s_barrier();  // the tile is consumed
  }
  store_sums(a.out, ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave, lane, acc);
}

// ------------------------------------------------------------------------------------------------------------------
// calibration: the same two loops UNFUSED.  k_gemm_only: 8 producer waves per workgroup (one CU each, as the library's
// kernel), v_posed written instance-innermost with non-temporal stores.  k_residual_only: 4 waves per workgroup, a wave
// = one 64-instance block x a run of slots, six non-temporal streams, two steps of requests in flight.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 1) void k_gemm_only(Args a, int tiles_per_chunk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kg = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  asm volatile("v_mov_b32 v255, 0" ::: "v255");
  const int N = 3 * a.Vp, ntiles = N / 32;
  const int tb = blockIdx.x * tiles_per_chunk, te = min(tb + tiles_per_chunk, ntiles);
  if (tb >= te) return;
  const int m0 = (blockIdx.y * 8 + wave) * 32;
  Features F;
  load_features(a.A, m0 + l31, kg, F);
  auto dma = [&](int t, int buf) {
    const char* src = a.img + (size_t)t * kImg;
#pragma unroll
    for (int i = 0; i < (NDMA + 7) / 8; ++i) {
      const int c = wave + 8 * i;
      if (c < NDMA) lds_dma16(src + c * 1024, lane * 16, smem + buf * kImg + c * 1024);
    }
  };
  dma(tb, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int t = tb; t < te; ++t) {
    const int buf = (t - tb) & 1;
    if (t + 1 < te) dma(t + 1, buf ^ 1);
    const f32x16 acc = tile_product(smem + buf * kImg, l31, kg, F);
    float* ccol = a.vpT + ((size_t)(m0 >> 6) * N + (size_t)t * 32) * 64 + (m0 & 63) + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) __builtin_nontemporal_store(acc[r], ccol + (size_t)((r & 3) + 8 * (r >> 2) + 4 * kg) * 64);
    if (t + 1 < te) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_residual_only(Args a, int slots_per_wave) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int blk = blockIdx.x, run = blockIdx.y * 4 + wave, Vp = a.Vp;
  const int v0 = run * slots_per_wave, v1 = min(v0 + slots_per_wave, Vp);
  if (v0 >= v1) return;
  const size_t cstr = (size_t)Vp * 64;
  const float* vp = a.vpT + (size_t)blk * 3 * Vp * 64 + lane;
  const float* tp = a.tT + (size_t)blk * 3 * Vp * 64 + lane;
  const float* jdb = a.jdT + (size_t)blk * a.J * JROW * 64 + lane;
  Sums acc;
#pragma unroll
  for (int k = 0; k < 5; ++k) acc.r1p[k] = mk2(0, 0);
#pragma unroll
  for (int k = 0; k < 4; ++k) { acc.m01[k] = mk2(0, 0); acc.m2[k] = 0.f; }
  Piece jr;
  float bA[6], bB[6];
  auto request = [&](int v, float (&b)[6]) {
    const int vv = v < v1 ? v : v1 - 1;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      b[c] = __builtin_nontemporal_load(vp + (size_t)c * cstr + (size_t)vv * 64);
      b[3 + c] = __builtin_nontemporal_load(tp + (size_t)c * cstr + (size_t)vv * 64);
    }
  };
  request(v0, bA);
  request(v0 + 1, bB);
  for (int v = v0; v < v1; v += 2) {
    if (((v - v0) & 31) == 0) load_piece(jdb, (v / 8) % a.J, a.J, jr);
    Rec r0, r1;
    load_rec(a.brec, v, r0);
    load_rec(a.brec, v + 1 < v1 ? v + 1 : v, r1);
    vertex_step(jr, r0, bA[0], bA[1], bA[2], bA[3], bA[4], bA[5], acc);
    request(v + 2, bA);
    vertex_step(jr, r1, bB[0], bB[1], bB[2], bB[3], bB[4], bB[5], acc);
    request(v + 3, bB);
  }
  store_sums(a.out, (size_t)blk * gridDim.y * 4 + run, lane, acc);
}

// plain evaluation of the split products for the check: one thread per (row n of 3 Vp, instance)
__global__ void k_reference(Args a, float* ref, int Mp) {
  const int inst = blockIdx.x * 64 + (threadIdx.x & 63), n = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (inst >= Mp || n >= 3 * a.Vp) return;
  const int t = n / 32, r = n % 32;
  const char* img = a.img + (size_t)t * kImg;
  auto b16 = [](const char* p) { uint32_t u = (uint32_t)(*reinterpret_cast<const uint16_t*>(p)) << 16; return __uint_as_float(u); };
  float acc = 0.f;
  for (int k = 0; k < KP; ++k) {
    const float x = a.A[(size_t)inst * KP + k];
    const float f1 = (float)(__bf16)x, f2 = (float)(__bf16)(x - f1);
    const char* e = img + r * (KP * 2) + (((k >> 3) ^ ((r >> 3) & 1)) * 16) + (k & 7) * 2;
    const float p1 = b16(e), p2 = b16(e + kPlane);
    acc += p1 * f1 + p1 * f2 + p2 * f1;
    if (k >= KP - 16) acc += b16(img + 2 * kPlane + r * 32 + ((k >> 3) & 1) * 16 + (k & 7) * 2) * f1;
  }
  ref[((size_t)(inst >> 6) * 3 * a.Vp + n) * 64 + (inst & 63)] = acc;
}

__global__ void k_fill(float* p, size_t n, uint32_t seed, float scale) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)i * 2654435761u ^ seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    p[i] = ((float)(h & 0xffffff) / 8388608.f - 1.f) * scale;
  }
}
__global__ void k_fill_bf16(uint16_t* p, size_t n, uint32_t seed, float scale) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)i * 2654435761u ^ seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    const float x = ((float)(h & 0xffffff) / 8388608.f - 1.f) * scale;
    p[i] = (uint16_t)(__float_as_uint(x) >> 16);
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <class F>
float time_us(F&& launch, int reps = 12) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f, sum = 0.f;
  for (int r = 0; r < reps + 2; ++r) {
    CK(hipEventRecord(e0));
    launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (r >= 2) { best = ms < best ? ms : best; sum += ms; }
  }
  CK(hipGetLastError());
  printf("   (mean %.1f us)", sum / reps * 1e3f);
  return best * 1e3f;
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 4096;
  const int Vp = 6912, J = 24, Mp = (B + 127) / 128 * 128, nblk = Mp / 64, ntile = Vp / 32;
  Args a{};
  a.Vp = Vp; a.J = J;
  float *A, *tT, *vpT, *jdT, *brec, *out, *dump, *ref;
  char* img;
  const size_t nstream = (size_t)nblk * 3 * Vp * 64;
  CK(hipMalloc(&A, (size_t)Mp * KP * 4)); CK(hipMalloc(&img, (size_t)3 * ntile * kImg));
  CK(hipMalloc(&tT, nstream * 4)); CK(hipMalloc(&vpT, nstream * 4)); CK(hipMalloc(&dump, nstream * 4)); CK(hipMalloc(&ref, nstream * 4));
  CK(hipMalloc(&jdT, (size_t)nblk * J * JROW * 64 * 4)); CK(hipMalloc(&brec, (size_t)Vp * BREC * 4));
  CK(hipMalloc(&out, (size_t)8192 * 22 * 64 * 4));
  k_fill<<<1024, 256>>>(A, (size_t)Mp * KP, 1u, 0.3f);
  k_fill_bf16<<<1024, 256>>>((uint16_t*)img, (size_t)3 * ntile * kImg / 2, 2u, 0.01f);
  k_fill<<<4096, 256>>>(tT, nstream, 3u, 1.0f);
  k_fill<<<1024, 256>>>(jdT, (size_t)nblk * J * JROW * 64, 4u, 1.0f);
  k_fill<<<1024, 256>>>(brec, (size_t)Vp * BREC, 5u, 0.25f);
  CK(hipMemset(dump, 0, nstream * 4));
  CK(hipDeviceSynchronize());
  a.A = A; a.img = img; a.tT = tT; a.vpT = vpT; a.jdT = jdT; a.brec = brec; a.out = out; a.dump = dump;
  CK(hipFuncSetAttribute((const void*)k_fused<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds));
  CK(hipFuncSetAttribute((const void*)k_fused<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds));
  CK(hipFuncSetAttribute((const void*)k_fused<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds));
  CK(hipFuncSetAttribute((const void*)k_gemm_only, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kImg));
  hipFuncAttributes fa;
  CK(hipFuncGetAttributes(&fa, (const void*)k_fused<false, false>));
  printf("k_fused: %d VGPRs, %zu B scratch, %zu B static LDS; B = %d, Vp = %d\n", fa.numRegs, (size_t)fa.localSizeBytes, (size_t)fa.sharedSizeBytes, B, Vp);
  CK(hipFuncGetAttributes(&fa, (const void*)k_fused<true, false>));
  printf("k_fused<WRITE_VPT>: %d VGPRs, %zu B scratch\n", fa.numRegs, (size_t)fa.localSizeBytes);

  // ---- check: the values the consumers read = the split products ----
  {
    const int nchunk = 8;
    a.tiles_per_wg = (ntile + nchunk - 1) / nchunk; a.jperiod = 1;
    hipLaunchKernelGGL((k_fused<false, true>), dim3(nchunk, Mp / 128), dim3(64 * NWAVE), kLds, 0, a);
    hipLaunchKernelGGL(k_reference, dim3(Mp / 64, 3 * Vp / 4), dim3(256), 0, 0, a, ref, Mp);
    CK(hipDeviceSynchronize());
    std::vector<float> hd(nstream), hr(nstream);
    CK(hipMemcpy(hd.data(), dump, nstream * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hr.data(), ref, nstream * 4, hipMemcpyDeviceToHost));
    double maxd = 0, maxv = 0;
    size_t bad = 0;
    for (size_t i = 0; i < nstream; ++i) {
      const double d = std::abs((double)hd[i] - hr[i]);
      maxd = d > maxd ? d : maxd;
      maxv = std::abs(hr[i]) > maxv ? std::abs(hr[i]) : maxv;
      if (d > 1e-5) ++bad;
    }
    printf("check: v_posed seen by the consumers vs plain split products: max |diff| %.3g (max |value| %.3g), %zu of %zu beyond 1e-5\n",
           maxd, maxv, bad, nstream);
  }
  // ---- timings ----
  for (int nchunk : {8, 16, 24}) {
    a.tiles_per_wg = (ntile + nchunk - 1) / nchunk;
    for (int jp : {1, 2}) {
      a.jperiod = jp;
      printf("fused            chunks %2d (tiles / workgroup %2d) joints every %d tile(s):", nchunk, a.tiles_per_wg, jp);
      const float us = time_us([&] { hipLaunchKernelGGL((k_fused<false, false>), dim3(nchunk, Mp / 128), dim3(64 * NWAVE), kLds, 0, a); });
      printf("  %7.1f us\n", us);
    }
    a.jperiod = 2;
    printf("fused + vpT out  chunks %2d (tiles / workgroup %2d) joints every 2 tile(s):", nchunk, a.tiles_per_wg);
    const float us = time_us([&] { hipLaunchKernelGGL((k_fused<true, false>), dim3(nchunk, Mp / 128), dim3(64 * NWAVE), kLds, 0, a); });
    printf("  %7.1f us\n", us);
  }
  {
    const int nchunk = 8;
    a.tiles_per_wg = (ntile + nchunk - 1) / nchunk; a.jperiod = 2;
    CK(hipFuncSetAttribute((const void*)k_fused<false, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds));
    CK(hipFuncSetAttribute((const void*)k_fused<false, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds));
    CK(hipFuncSetAttribute((const void*)k_fused<false, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds));
    printf("ablation: producers alone (consumers only meet the barriers):");
    float us = time_us([&] { hipLaunchKernelGGL((k_fused<false, false, 1>), dim3(nchunk, Mp / 128), dim3(64 * NWAVE), kLds, 0, a); });
    printf("  %7.1f us\n", us);
    printf("ablation: consumers alone (producers only meet the barriers):");
    us = time_us([&] { hipLaunchKernelGGL((k_fused<false, false, 2>), dim3(nchunk, Mp / 128), dim3(64 * NWAVE), kLds, 0, a); });
    printf("  %7.1f us\n", us);
    printf("ablation: consumers without the LDS reads of v_posed:");
    us = time_us([&] { hipLaunchKernelGGL((k_fused<false, false, 4>), dim3(nchunk, Mp / 128), dim3(64 * NWAVE), kLds, 0, a); });
    printf("  %7.1f us\n", us);
  }
  {
    CK(hipFuncSetAttribute((const void*)k_fused_sym<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSymLds));
    CK(hipFuncSetAttribute((const void*)k_fused_sym<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSymLds));
    CK(hipFuncSetAttribute((const void*)k_fused_sym<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSymLds));
    CK(hipFuncSetAttribute((const void*)k_fused_sym<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSymLds));
    CK(hipFuncGetAttributes(&fa, (const void*)k_fused_sym<false>));
    printf("k_fused_sym: %d VGPRs, %zu B scratch\n", fa.numRegs, (size_t)fa.localSizeBytes);
    for (int nchunk : {16, 24, 32, 48}) {
      a.tiles_per_wg = (ntile + nchunk - 1) / nchunk; a.jperiod = 2;
      printf("symmetric fused            chunks %2d (tiles / workgroup %2d):", nchunk, a.tiles_per_wg);
      float us = time_us([&] { hipLaunchKernelGGL((k_fused_sym<false>), dim3(nchunk, Mp / 256), dim3(512), kSymLds, 0, a); });
      printf("  %7.1f us\n", us);
      printf("symmetric fused + vpT out  chunks %2d (tiles / workgroup %2d):", nchunk, a.tiles_per_wg);
      us = time_us([&] { hipLaunchKernelGGL((k_fused_sym<true>), dim3(nchunk, Mp / 256), dim3(512), kSymLds, 0, a); });
      printf("  %7.1f us\n", us);
    }
    a.tiles_per_wg = (ntile + 15) / 16;
    printf("symmetric ablation: products alone (no vertex phase):");
    float us = time_us([&] { hipLaunchKernelGGL((k_fused_sym<false, 1>), dim3(16, Mp / 256), dim3(512), kSymLds, 0, a); });
    printf("  %7.1f us\n", us);
    printf("symmetric ablation: vertex phase alone (no products):");
    us = time_us([&] { hipLaunchKernelGGL((k_fused_sym<false, 2>), dim3(16, Mp / 256), dim3(512), kSymLds, 0, a); });
    printf("  %7.1f us\n", us);
  }
  {
    const int ny = (Mp + 255) / 256, ntiles = 3 * Vp / 32;
    int nchunk = (2 * 256 / ny + 4) / 8 * 8;
    if (nchunk < 8) nchunk = 8;
    const int per = (ntiles + nchunk - 1) / nchunk;
    printf("unfused GEMM     (8 waves x 32 instances, %d x %d workgroups):", nchunk, ny);
    const float us = time_us([&] { hipLaunchKernelGGL(k_gemm_only, dim3(nchunk, ny), dim3(512), 2 * kImg, 0, a, per); });
    printf("  %7.1f us\n", us);
    for (int spw : {108, 216}) {
      const int runs = (Vp + spw - 1) / spw;
      printf("unfused residual (%3d slots per wave, %d x %d workgroups):", spw, nblk, (runs + 3) / 4);
      const float us2 = time_us([&] { hipLaunchKernelGGL(k_residual_only, dim3(nblk, (runs + 3) / 4), dim3(256), 0, 0, a, spw); });
      printf("  %7.1f us\n", us2);
    }
  }
  return 0;
}
