// Round-6 reproducer attempt for the cross-wave fault of DESIGN.md section 4 / 8.3 (rounds 2 - 4: a wave whose neighbour on
// the CU streams LDS fragments into bf16 MFMAs read a wrong last quarter — lanes 48-63 — of a uniform-address LDS read).
// The victim here has the FORM of the one kernel that was still hit in round 3 (the old k_pair_gram_bm, git 829c94d):
// a ONE-wave workgroup, a DOUBLE-BUFFERED table in LDS that the wave itself refills from global memory while it reads the
// other half back with uniform-address reads (no barrier: one wave), a register-heavy body.  Swept: the victim's read
// width (4 / 8 / 16 bytes), whether the aggressor refills its fragments by LDS-DMA (global_load_lds_dwordx4, as the
// library's GEMM does), s_setprio around the aggressor's MFMAs, and the LDS size of the aggressor (two or one of its
// workgroups per CU).  Every table value is a small integer: the victim's sums are exact and must equal the solo run's.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_neighbour_r6.hip -o /tmp/lds_neighbour_r6 && /tmp/lds_neighbour_r6
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void lds_dma16(const void* src_uniform, uint32_t lane_off, void* lds_base) {
  const uint32_t l = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds_base);  // LDS byte address of the 1 KB destination (lane i -> + 16 i)
  const uint64_t a = (uint64_t)(uintptr_t)src_uniform;
  const uint64_t au = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(a >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)a);
  src_uniform = (const void*)(uintptr_t)au;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(l), "v"(lane_off), "s"(src_uniform) : "memory");
}

// aggressor: the inner loop of the library's split-bf16 GEMM (3 fragment reads + 6 bf16 MFMAs per k-step)
template <bool DMA, bool PRIO>
__global__ __launch_bounds__(256, 2) void aggressor(const float* __restrict__ img, float* out, int tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 79872 / 4; i += 256) reinterpret_cast<float*>(smem)[i] = 1.0f;
  __syncthreads();
  bf16x8 f[13];
  for (int s = 0; s < 13; ++s) for (int q = 0; q < 8; ++q) f[s][q] = (__bf16)(0.25f * (q + s));
  const char* base0 = smem + (lane & 31) * 416 + (lane >> 5) * 16;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int t = 0; t < tiles; ++t) {
    const char* base = base0 + (t & 1) * 39936;
    if (DMA) {  // refill the OTHER half of the ring: 39 x 1 KB per tile image, dealt to the four waves
      char* dst = smem + ((t + 1) & 1) * 39936;
      for (int p = wave; p < 39; p += 4) lds_dma16(img + (size_t)((t * 39 + p) & 1023) * 256, (uint32_t)lane * 16u, dst + p * 1024);
    }
#pragma unroll
    for (int s = 0; s < 13; ++s) {
      const bf16x8 p1 = *reinterpret_cast<const bf16x8*>(base + s * 32);
      const bf16x8 p2 = *reinterpret_cast<const bf16x8*>(base + 13312 + s * 32);
      const bf16x8 p3 = *reinterpret_cast<const bf16x8*>(base + 26624 + s * 32);
      if (PRIO) __builtin_amdgcn_s_setprio(3);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p3, f[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p2, f[(s + 1) % 13], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p1, f[(s + 2) % 13], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p2, f[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p1, f[(s + 1) % 13], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p1, f[s], acc, 0, 0, 0);
      if (PRIO) __builtin_amdgcn_s_setprio(0);
    }
    if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  float sacc = 0;
  for (int r = 0; r < 16; ++r) sacc += acc[r];
  if (sacc == 1234.5f) out[0] = sacc;
}

// victim: one wave; table of NT floats per unit, double-buffered; W = bytes per uniform-address read
constexpr int NT = 960;
template <int W>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void victim(const float* __restrict__ tabs, float* out, int units) {
  __shared__ __attribute__((aligned(16))) float c1s[2][NT];
  const int lane = threadIdx.x;
  float nx[NT / 64];
  auto fetch = [&](int u) {
#pragma unroll
    for (int k = 0; k < NT / 64; ++k) nx[k] = tabs[(size_t)((u + blockIdx.x) & 63) * NT + k * 64 + lane];
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int k = 0; k < NT / 64; ++k) c1s[buf][k * 64 + lane] = nx[k];
  };
  f2 acc[48];
#pragma unroll
  for (int k = 0; k < 48; ++k) acc[k] = f2{0.f, 0.f};
  const f2 v = {(float)(lane % 3), (float)((lane + 1) % 3)};
  fetch(0);
  commit(0);
  for (int u = 0; u < units; ++u) {
    const int buf = u & 1;
    if (u + 1 < units) fetch(u + 1);
    const float* c1 = c1s[buf];
#pragma unroll
    for (int x = 0; x < 10; ++x) {
#pragma unroll
      for (int k = 0; k < 48; ++k) {
        f2 c;
        if (W == 4) { c.x = c1[x * 96 + 2 * k]; c.y = c1[x * 96 + 2 * k + 1]; }
        else if (W == 8) c = *reinterpret_cast<const f2*>(c1 + x * 96 + 2 * k);
        else { const f4 q = *reinterpret_cast<const f4*>(c1 + x * 96 + 4 * (k / 2)); c = (k & 1) ? f2{q.z, q.w} : f2{q.x, q.y}; }
        acc[k] += c * v;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if ((u & 15) == 15)
#pragma unroll
      for (int k = 0; k < 48; ++k) { acc[k].x = acc[k].x > 4.0e5f ? acc[k].x - 4.0e5f : acc[k].x; acc[k].y = acc[k].y > 4.0e5f ? acc[k].y - 4.0e5f : acc[k].y; }
    if (u + 1 < units) commit(buf ^ 1);
  }
#pragma unroll
  for (int k = 0; k < 48; ++k) {
    out[((size_t)blockIdx.x * 96 + 2 * k) * 64 + lane] = acc[k].x;
    out[((size_t)blockIdx.x * 96 + 2 * k + 1) * 64 + lane] = acc[k].y;
  }
}

template <int W>
void launch_victim(int nv, hipStream_t s, const float* tabs, float* dv, int units) { hipLaunchKernelGGL(victim<W>, dim3(nv), dim3(64), 0, s, tabs, dv, units); }

int main() {
  const int NV = 2048, UNITS = 600;
  const size_t n = (size_t)NV * 96 * 64;
  float *dv, *da, *tabs, *img;
  (void)hipMalloc(&dv, n * 4); (void)hipMalloc(&da, 4096); (void)hipMalloc(&tabs, 64 * NT * 4); (void)hipMalloc(&img, 1024 * 1024);
  std::vector<float> ht(64 * NT);
  for (size_t i = 0; i < ht.size(); ++i) ht[i] = (float)((i * 7 + i / NT) % 5);
  (void)hipMemcpy(tabs, ht.data(), ht.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemset(img, 0, 1024 * 1024);
  hipStream_t s1, s2; (void)hipStreamCreate(&s1); (void)hipStreamCreate(&s2);
#define SET_LDS(K) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&K), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
  SET_LDS((aggressor<false, false>)); SET_LDS((aggressor<true, false>)); SET_LDS((aggressor<false, true>)); SET_LDS((aggressor<true, true>));
  std::vector<float> ref(n), cur(n);
  size_t total_bad = 0;
  for (int W : {4, 8, 16}) {
    auto lv = [&](hipStream_t s) { if (W == 4) launch_victim<4>(NV, s, tabs, dv, UNITS); else if (W == 8) launch_victim<8>(NV, s, tabs, dv, UNITS); else launch_victim<16>(NV, s, tabs, dv, UNITS); };
    lv(s2);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(ref.data(), dv, n * 4, hipMemcpyDeviceToHost);
    for (int lds_kb : {78, 100})
      for (int dma = 0; dma < 2; ++dma)
        for (int prio = 0; prio < 2; ++prio)
          for (int rep = 0; rep < 2; ++rep) {
            (void)hipMemsetAsync(dv, 0, n * 4, s2);
            (void)hipDeviceSynchronize();
            const size_t lds = (size_t)lds_kb * 1024;
            const int tiles = 3000;
            if (!dma && !prio) hipLaunchKernelGGL((aggressor<false, false>), dim3(1024), dim3(256), lds, s1, img, da, tiles);
            if (dma && !prio) hipLaunchKernelGGL((aggressor<true, false>), dim3(1024), dim3(256), lds, s1, img, da, tiles);
            if (!dma && prio) hipLaunchKernelGGL((aggressor<false, true>), dim3(1024), dim3(256), lds, s1, img, da, tiles);
            if (dma && prio) hipLaunchKernelGGL((aggressor<true, true>), dim3(1024), dim3(256), lds, s1, img, da, tiles);
            for (int k = 0; k < 3; ++k) lv(s2);
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(cur.data(), dv, n * 4, hipMemcpyDeviceToHost);
            size_t bad = 0; int lanes[64] = {0};
            for (size_t i = 0; i < n; ++i) if (cur[i] != ref[i]) { ++bad; ++lanes[i % 64]; }
            total_bad += bad;
            printf("victim read %2d B | aggressor lds %3d KB dma %d setprio %d rep %d: %zu of %zu victim values differ", W, lds_kb, dma, prio, rep, bad, n);
            if (bad) { printf("; lanes:"); for (int l = 0; l < 64; ++l) if (lanes[l]) printf(" %d(%d)", l, lanes[l]); }
            printf("\n");
          }
  }
  printf("total differing values: %zu\n", total_bad);
  return 0;
}
