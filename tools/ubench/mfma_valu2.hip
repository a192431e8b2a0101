// Micro-benchmark 2: (a) does VALU work hide under a BF16 MFMA chain (matrix core) from the same wave?
// (b) do an f32-MFMA-only wave and a VALU-only wave on the same SIMD overlap (2 waves per SIMD)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// MODE 0: bf16 32x32x16 MFMA + K v_fma per MFMA (one wave per SIMD)
// MODE 1: 512 threads: waves 0-3 f32 MFMA chain only, waves 4-7 K v_fma per "slot" only (same trip count)
// MODE 2: 512 threads: waves 0-3 bf16 MFMA chain only, waves 4-7 VALU only
// MODE 3: 512 threads: all waves VALU only (K v_fma per slot)           -> VALU-alone reference at 2 waves/SIMD
// MODE 4: 512 threads: waves 0-3 f32 MFMA only, waves 4-7 idle           -> MFMA-alone reference
template <int K, int MODE>
__global__ __launch_bounds__(512, 1) void bench(float* out, int iters) {
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float a = 1.0f + tid * 1e-7f, b = 0.5f;
  bf16x8 pa, pb;
  for (int k = 0; k < 8; ++k) { pa[k] = (__bf16)(0.5f + k); pb[k] = (__bf16)(0.25f * k); }
  float v[16];
  for (int k = 0; k < 16; ++k) v[k] = (float)k + tid;
  const bool do_mfma = MODE == 0 || ((MODE == 1 || MODE == 2 || MODE == 4) && wave < 4);
  const bool do_valu = MODE == 0 || MODE == 3 || ((MODE == 1 || MODE == 2) && wave >= 4);
  if (do_mfma && do_valu) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(pa), "v"(pb));
#pragma unroll
        for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[k & 15]) : "v"(a), "v"(b));
      }
    }
  } else if (do_mfma) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (MODE == 2) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(pa), "v"(pb));
        else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
      }
    }
  } else if (do_valu) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
#pragma unroll
        for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[k & 15]) : "v"(a), "v"(b));
      }
    }
  }
  float s = 0;
  for (int r = 0; r < 16; ++r) s += acc[r];
  for (int k = 0; k < 16; ++k) s += v[k];
  out[blockIdx.x * 512 + tid] = s;
}

template <int K, int MODE>
void run(const char* name, float* d, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((bench<K, MODE>), dim3(256), dim3(MODE == 0 ? 256 : 512), 0, 0, d, 10);
  (void)hipDeviceSynchronize();
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((bench<K, MODE>), dim3(256), dim3(MODE == 0 ? 256 : 512), 0, 0, d, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double slots = (double)iters * 8;
  printf("%-34s K=%2d  %8.3f ms  %7.1f cyc/slot at 2.4 GHz\n", name, K, best, best * 1e6 / slots * 2.4);
}

int main() {
  float* d; (void)hipMalloc(&d, 256 * 512 * 4);
  const int it = 20000;
  run<0, 0>("bf16 mfma only (same wave)", d, it);
  run<2, 0>("bf16 mfma + v_fma same wave", d, it); run<4, 0>("bf16 mfma + v_fma same wave", d, it);
  run<5, 0>("bf16 mfma + v_fma same wave", d, it); run<6, 0>("bf16 mfma + v_fma same wave", d, it);
  run<8, 0>("bf16 mfma + v_fma same wave", d, it); run<12, 0>("bf16 mfma + v_fma same wave", d, it);
  run<0, 4>("f32 mfma waves alone", d, it);
  run<8, 3>("valu alone, 2 waves/SIMD", d, it); run<16, 3>("valu alone, 2 waves/SIMD", d, it);
  run<4, 1>("f32 mfma waves || valu waves", d, it); run<8, 1>("f32 mfma waves || valu waves", d, it);
  run<12, 1>("f32 mfma waves || valu waves", d, it); run<16, 1>("f32 mfma waves || valu waves", d, it);
  run<4, 2>("bf16 mfma waves || valu waves", d, it); run<8, 2>("bf16 mfma waves || valu waves", d, it);
  run<16, 2>("bf16 mfma waves || valu waves", d, it);
  return 0;
}
