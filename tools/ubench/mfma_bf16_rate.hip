// bf16 MFMA issue rate per SIMD: 1 or 2 waves per SIMD, dependent chain on one accumulator vs two
// accumulators, operands fixed vs rotating over many registers (as in the GEMM's register-stationary form).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC, bool ROT>
__global__ __launch_bounds__(512, 1) void bench(float* out, int iters) {
  const int tid = threadIdx.x;
  f32x16 acc0, acc1;
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  bf16x8 fa[13], fb[13];
  for (int s = 0; s < 13; ++s) for (int k = 0; k < 8; ++k) { fa[s][k] = (__bf16)(0.01f * (k + s) + tid * 1e-4f); fb[s][k] = (__bf16)(0.02f * k - s * 0.01f); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 13; ++s) {
      const int sa = ROT ? s : 0;
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        if (NACC == 1 || (u & 1) == 0)
          asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(fa[sa]), "v"(fb[(sa + u) % 13]));
        else
          asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(fa[sa]), "v"(fb[(sa + u) % 13]));
      }
    }
  }
  float s = 0;
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
  out[blockIdx.x * 512 + tid] = s;
}

template <int NACC, bool ROT>
void run(const char* name, float* d, int threads) {
  const int iters = 2000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((bench<NACC, ROT>), dim3(256), dim3(threads), 0, 0, d, 10);
  (void)hipDeviceSynchronize();
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((bench<NACC, ROT>), dim3(256), dim3(threads), 0, 0, d, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double per_simd = (double)iters * 78 * (threads / 256);  // MFMAs per SIMD
  printf("%-40s %d waves/SIMD: %7.3f ms, %.1f ns per MFMA per SIMD (%.1f cyc @2.4GHz), %.0f TF\n", name, threads / 256, best,
         best * 1e6 / per_simd, best * 1e6 / per_simd * 2.4, 1024.0 * per_simd * 32768 * 2 / (best * 1e-3) / 1e12);
}

int main() {
  float* d; (void)hipMalloc(&d, 256 * 512 * 4);
  run<1, false>("1 acc, fixed operands", d, 256);
  run<1, false>("1 acc, fixed operands", d, 512);
  run<1, true>("1 acc, rotating operands", d, 256);
  run<1, true>("1 acc, rotating operands", d, 512);
  run<2, true>("2 acc, rotating operands", d, 256);
  run<2, true>("2 acc, rotating operands", d, 512);
  return 0;
}
