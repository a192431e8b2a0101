// Micro-benchmark: how many independent fp32 VALU / LDS instructions fit in the shadow of a dependent chain
// of v_mfma_f32_32x32x2_f32 (64 cycles each) from ONE wave per SIMD?  Decides whether the residual pass
// can ride inside the posedirs GEMM (fused kernel) at the GEMM's MFMA-bound rate.
//   build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu tools/ubench/mfma_valu.hip ; run: ./mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int K, int MODE>  // MODE 0: v_fma_f32, 1: v_pk_fma_f32, 2: ds_read_b128 (K/4 per MFMA), 3: fma + 1 ds_read_b128
__global__ __launch_bounds__(256, 1) void bench(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[4096];
  const int tid = threadIdx.x;
  for (int i = tid; i < 4096; i += 256) lds[i] = (float)i * 1e-6f;
  __syncthreads();
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float a = 1.0f + tid * 1e-7f, b = 0.5f;
  float v[16];
  for (int k = 0; k < 16; ++k) v[k] = (float)k + tid;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p[8];
  for (int k = 0; k < 8; ++k) { p[k].x = k; p[k].y = tid; }
  float4 q = make_float4(0, 0, 0, 0);
  const float4* lp = reinterpret_cast<const float4*>(lds) + (tid & 63) * 3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
      if (MODE == 0 || MODE == 3) {
#pragma unroll
        for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[k & 15]) : "v"(a), "v"(b));
      }
      if (MODE == 1) {
#pragma unroll
        for (int k = 0; k < K; ++k) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(p[k & 7]) : "v"(p[(k + 1) & 7]));
      }
      if (MODE == 2 || MODE == 3) {
#pragma unroll
        for (int k = 0; k < (MODE == 3 ? 1 : K / 4); ++k) {
          float4 r = lp[(u * 4 + k) & 31];
          asm volatile("" : "+v"(r.x), "+v"(r.y), "+v"(r.z), "+v"(r.w));
          q.x += r.x;
        }
      }
    }
  }
  float s = q.x;
  for (int r = 0; r < 16; ++r) s += acc[r];
  for (int k = 0; k < 16; ++k) s += v[k];
  for (int k = 0; k < 8; ++k) s += p[k].x + p[k].y;
  out[blockIdx.x * 256 + tid] = s;
}

template <int K, int MODE>
void run(const char* name, float* d, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((bench<K, MODE>), dim3(256), dim3(256), 0, 0, d, 10);
  hipDeviceSynchronize();
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((bench<K, MODE>), dim3(256), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double mf = (double)iters * 8;
  printf("%-14s K=%2d  %8.3f ms  %7.1f ns/MFMA  (%.1f cyc at 2.4 GHz)  MFMA rate %.1f TF\n", name, K, best,
         best * 1e6 / mf, best * 1e6 / mf * 2.4, 256.0 * 4 * mf * 4096 / (best * 1e-3) / 1e12);
}

int main() {
  float* d; hipMalloc(&d, 256 * 256 * 4);
  const int it = 20000;
  run<0, 0>("mfma only", d, it);
  run<4, 0>("v_fma", d, it); run<8, 0>("v_fma", d, it); run<10, 0>("v_fma", d, it); run<12, 0>("v_fma", d, it);
  run<14, 0>("v_fma", d, it); run<16, 0>("v_fma", d, it); run<20, 0>("v_fma", d, it); run<24, 0>("v_fma", d, it); run<32, 0>("v_fma", d, it);
  run<2, 1>("v_pk_fma", d, it); run<4, 1>("v_pk_fma", d, it); run<6, 1>("v_pk_fma", d, it); run<8, 1>("v_pk_fma", d, it); run<12, 1>("v_pk_fma", d, it);
  run<4, 2>("ds_read_b128", d, it); run<8, 2>("ds_read_b128", d, it);
  run<8, 3>("fma+ds128", d, it); run<10, 3>("fma+ds128", d, it); run<12, 3>("fma+ds128", d, it);
  return 0;
}
