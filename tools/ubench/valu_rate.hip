// VALU issue cost on gfx950: v_fma_f32 vs v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32, 8 independent chains per wave,
// 1 / 2 / 4 waves per SIMD.  Reports cycles per instruction per SIMD (wall time x 2.4 GHz; the real clock may be lower).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(1024) void bench(float* out, int iters) {
  float a[8]; f2 p[8];
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 1e-3f + i; p[i].x = a[i]; p[i].y = a[i] * 0.5f; }
  const float c = 0.999f, d = 1e-3f; f2 c2; c2.x = c; c2.y = c; f2 d2; d2.x = d; d2.y = d;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d));
        if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(c2), "v"(d2));
        if (OP == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
        if (OP == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(d2));
        if (OP == 4) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        if (OP == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,0]" : "+v"(p[i]) : "v"(c2), "v"(d2));
      }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
  out[blockIdx.x * 1024 + threadIdx.x] = s;
}

template <int OP>
void run(const char* name, float* d) {
  const int iters = 4000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int threads : {256, 512, 1024}) {
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL((bench<OP>), dim3(256), dim3(threads), 0, 0, d, iters);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      if (rep && ms < best) best = ms;
    }
    const double per_simd = (double)iters * 64 * (threads / 256);
    printf("%-28s %d waves/SIMD: %7.3f ms  %.2f cyc/instr/SIMD @2.4GHz\n", name, threads / 256, best, best * 1e-3 * 2.4e9 / per_simd);
  }
}

int main() {
  float* d; (void)hipMalloc(&d, 256 * 1024 * 4);
  run<0>("v_fma_f32", d); run<4>("v_mul_f32", d); run<1>("v_pk_fma_f32", d); run<5>("v_pk_fma_f32 op_sel_hi", d);
  run<2>("v_pk_mul_f32", d); run<3>("v_pk_add_f32", d);
  return 0;
}
