// Micro-benchmark 3 (correctness): do VALU results of one wave stay bit-exact while another wave on the
// same SIMD runs a bf16 MFMA chain?  512-thread workgroups: waves 0-3 run MFMAs (MODE 1: bf16 32x32x16,
// MODE 2: f32 32x32x2, MODE 0: nothing), waves 4-7 run a deterministic packed/plain fp32 FMA recurrence whose
// final values are compared with the MODE 0 run.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE, bool PK>
__global__ __launch_bounds__(512, 1) void bench(float* out, int iters) {
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (wave < 4) {
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float a = 1.0f + tid * 1e-7f, b = 0.5f;
    bf16x8 pa, pb;
    for (int k = 0; k < 8; ++k) { pa[k] = (__bf16)(0.5f + k + tid * 0.01f); pb[k] = (__bf16)(0.25f * k); }
    if (MODE != 0) {
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (MODE == 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(pa), "v"(pb));
          else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
        }
      }
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc[r];
    if (s == 12345.678f) out[0] = s;  // keep the chain alive
    return;
  }
  // VALU waves: 16 independent recurrences per lane
  const int lane = tid & 63, g = blockIdx.x * 4 + (wave - 4);
  float x[16];
  for (int k = 0; k < 16; ++k) x[k] = 0.001f * (lane + 1) + 0.01f * k + 1e-4f * (g % 97);
  const float a = 0.99993f, b = 1.0e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (PK) {
#pragma unroll
        for (int k = 0; k < 16; k += 2) {
          f2 v; v.x = x[k]; v.y = x[k + 1];
          f2 aa; aa.x = a; aa.y = a; f2 bb; bb.x = b; bb.y = b;
          asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(aa), "v"(bb));
          x[k] = v.x; x[k + 1] = v.y;
        }
      } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(a), "v"(b));
      }
    }
  }
  for (int k = 0; k < 16; ++k) out[((size_t)g * 16 + k) * 64 + lane] = x[k];
}

template <int MODE, bool PK>
std::vector<float> run(float* d, int iters, size_t n) {
  hipLaunchKernelGGL((bench<MODE, PK>), dim3(512), dim3(512), 0, 0, d, iters);
  (void)hipDeviceSynchronize();
  std::vector<float> h(n);
  (void)hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
  return h;
}

template <bool PK>
void test(float* d, size_t n, const char* name) {
  const int it = 20000;
  auto ref = run<0, PK>(d, it, n);
  for (int mode = 1; mode <= 2; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      auto cur = mode == 1 ? run<1, PK>(d, it, n) : run<2, PK>(d, it, n);
      size_t bad = 0; int lanes[64] = {0};
      for (size_t i = 0; i < n; ++i) if (cur[i] != ref[i]) { ++bad; ++lanes[i % 64]; }
      printf("%s neighbour %s rep %d: %zu of %zu values differ", name, mode == 1 ? "bf16 mfma" : "f32 mfma ", rep, bad, n);
      if (bad) { printf("; lanes:"); for (int l = 0; l < 64; ++l) if (lanes[l]) printf(" %d(%d)", l, lanes[l]); }
      printf("\n");
    }
  }
}

int main() {
  const size_t n = (size_t)512 * 4 * 16 * 64;
  float* d; (void)hipMalloc(&d, n * 4);
  test<false>(d, n, "v_fma_f32   ");
  test<true>(d, n, "v_pk_fma_f32");
  return 0;
}
