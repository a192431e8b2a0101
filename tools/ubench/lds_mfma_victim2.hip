// Reproducer 2: aggressor and victim are DIFFERENT kernels on different streams (as in the chunked fit).
// Aggressor: the inner loop of k_posedirs_gemm_bf16x3 (3 ds_read_b128 + 6 bf16 MFMAs per k-step).
// Victim: one-wave workgroups accumulating packed-fp32 FMAs whose operands are LDS broadcast reads; every
// value is a small integer, so the sums are exact and must be bit-identical to the solo run.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int AGG>  // 1 lds+mfma, 2 mfma only, 3 lds only
__global__ __launch_bounds__(256, 2) void aggressor(float* out, int tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 79872 / 4; i += 256) reinterpret_cast<float*>(smem)[i] = 1.0f;
  __syncthreads();
  bf16x8 f[13];
  for (int s = 0; s < 13; ++s) for (int q = 0; q < 8; ++q) f[s][q] = (__bf16)(0.25f * (q + s));
  const char* base0 = smem + (lane & 31) * 416 + (lane >> 5) * 16;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int t = 0; t < tiles; ++t) {
    const char* base = base0 + (t & 1) * 39936;
#pragma unroll
    for (int s = 0; s < 13; ++s) {
      bf16x8 p1 = f[s], p2 = f[(s + 1) % 13], p3 = f[(s + 2) % 13];
      if (AGG != 2) {
        p1 = *reinterpret_cast<const bf16x8*>(base + s * 32);
        p2 = *reinterpret_cast<const bf16x8*>(base + 13312 + s * 32);
        p3 = *reinterpret_cast<const bf16x8*>(base + 26624 + s * 32);
      }
      if (AGG != 3) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p3, f[s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p2, f[(s + 1) % 13], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p1, f[(s + 2) % 13], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p2, f[s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p1, f[(s + 1) % 13], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p1, f[s], acc, 0, 0, 0);
      } else {
        acc[0] += (float)p1[0] + (float)p2[1] + (float)p3[2];
      }
    }
    __syncthreads();
  }
  float sacc = 0;
  for (int r = 0; r < 16; ++r) sacc += acc[r];
  if (sacc == 1234.5f) out[0] = sacc;
}

__global__ __launch_bounds__(64) void victim(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float tab[2048];
  const int lane = threadIdx.x;
  for (int i = lane; i < 2048; i += 64) tab[i] = (float)((i * 5 + blockIdx.x) % 7);
  __syncthreads();
  f2 acc[16];
  for (int k = 0; k < 16; ++k) { acc[k].x = 0.f; acc[k].y = 0.f; }
  const f2 v = {(float)(lane % 3), (float)((lane + 1) % 3)};
  for (int it = 0; it < iters; ++it) {
    const float4* row = reinterpret_cast<const float4*>(tab) + ((it * 8) & 511);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float4 c = row[u];          // uniform address: broadcast read
      f2 c01 = {c.x, c.y}, c23 = {c.z, c.w};
      acc[2 * u] += c01 * v;            // v_pk_fma_f32 / v_pk_mul + v_pk_add
      acc[2 * u + 1] += c23 * v;
    }
    if ((it & 63) == 63)
      for (int k = 0; k < 16; ++k) { acc[k].x = acc[k].x > 4.0e5f ? acc[k].x - 4.0e5f : acc[k].x; acc[k].y = acc[k].y > 4.0e5f ? acc[k].y - 4.0e5f : acc[k].y; }
  }
  for (int k = 0; k < 16; ++k) {
    out[((size_t)blockIdx.x * 32 + 2 * k) * 64 + lane] = acc[k].x;
    out[((size_t)blockIdx.x * 32 + 2 * k + 1) * 64 + lane] = acc[k].y;
  }
}

int main() {
  const int NV = 4096;
  const size_t n = (size_t)NV * 32 * 64;
  float *dv, *da; (void)hipMalloc(&dv, n * 4); (void)hipMalloc(&da, 4096);
  hipStream_t s1, s2; (void)hipStreamCreate(&s1); (void)hipStreamCreate(&s2);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&aggressor<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&aggressor<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&aggressor<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int iters = 3000;
  hipLaunchKernelGGL(victim, dim3(NV), dim3(64), 0, s2, dv, iters);
  (void)hipDeviceSynchronize();
  std::vector<float> ref(n), cur(n);
  (void)hipMemcpy(ref.data(), dv, n * 4, hipMemcpyDeviceToHost);
  const char* names[4] = {"", "lds+mfma", "mfma only", "lds only"};
  for (int lds_kb : {78, 84})
    for (int agg = 1; agg <= 3; ++agg)
      for (int rep = 0; rep < 3; ++rep) {
        (void)hipMemsetAsync(dv, 0, n * 4, s2);
        (void)hipDeviceSynchronize();
        const size_t lds = (size_t)lds_kb * 1024;
        if (agg == 1) hipLaunchKernelGGL(aggressor<1>, dim3(1024), dim3(256), lds, s1, da, 400);
        if (agg == 2) hipLaunchKernelGGL(aggressor<2>, dim3(1024), dim3(256), lds, s1, da, 400);
        if (agg == 3) hipLaunchKernelGGL(aggressor<3>, dim3(1024), dim3(256), lds, s1, da, 400);
        for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(victim, dim3(NV), dim3(64), 0, s2, dv, iters);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(cur.data(), dv, n * 4, hipMemcpyDeviceToHost);
        size_t bad = 0; int lanes[64] = {0};
        for (size_t i = 0; i < n; ++i) if (cur[i] != ref[i]) { ++bad; ++lanes[i % 64]; }
        printf("aggressor %-9s lds %d KB rep %d: %zu of %zu victim values differ", names[agg], lds_kb, rep, bad, n);
        if (bad) { printf("; lanes:"); for (int l = 0; l < 64; ++l) if (lanes[l]) printf(" %d(%d)", l, lanes[l]); }
        printf("\n");
      }
  return 0;
}
