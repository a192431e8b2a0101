// Reproducer attempt: an "aggressor" wave streams ds_read_b128 fragments into bf16 MFMAs while a "victim"
// wave on the same SIMD sums values it reads from LDS (uniform-address broadcast reads, or per-lane reads).
// The victim's result has a known exact value; count mismatching lanes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// AGG: 0 none, 1 mfma+lds, 2 mfma only, 3 lds only.   VIC: 0 broadcast b128, 1 per-lane b128 (48-byte stride)
template <int AGG, int VIC>
__global__ __launch_bounds__(512, 1) void k(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // 16 KB victim table + 64 KB aggressor area
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 20480; i += 512) lds[i] = (i < 4096) ? (float)((i * 7) % 13) : 0.001f * (i % 17);
  __syncthreads();
  if (wave < 4) {
    if (AGG == 0) return;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    bf16x8 fb;
    for (int q = 0; q < 8; ++q) fb[q] = (__bf16)(0.5f + q);
    const char* base = reinterpret_cast<const char*>(lds + 4096) + wave * 16384 + (lane & 31) * 416 + (lane >> 5) * 16;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < 12; ++s) {
        bf16x8 p = fb;
        if (AGG == 1 || AGG == 3) p = *reinterpret_cast<const bf16x8*>(base + s * 32);
        if (AGG == 1 || AGG == 2) {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p, fb, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p, fb, acc, 0, 0, 0);
        } else {
          acc[0] += (float)p[0];
        }
      }
    }
    float sacc = 0;
    for (int r = 0; r < 16; ++r) sacc += acc[r];
    if (sacc == 1234.5f) out[0] = sacc;
    return;
  }
  // victim: integer-valued sums (exact in fp32)
  float sum[4] = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int idx = VIC == 0 ? ((it * 16 + u) & 255) * 4 : (((it * 16 + u) & 15) * 64 + lane) * 12 / 4 * 4 % 4096;
      const float4 v = *reinterpret_cast<const float4*>(lds + (VIC == 0 ? idx : (((it + u) & 15) * 64 * 3 + lane * 3) * 4 % 4000 / 4 * 4));
      sum[0] += v.x; sum[1] += v.y; sum[2] += v.z; sum[3] += v.w;
      if ((u & 3) == 3) {
#pragma unroll
        for (int c = 0; c < 4; ++c) sum[c] = sum[c] > 1.0e6f ? sum[c] - 1.0e6f : sum[c];
      }
    }
  }
  const int g = blockIdx.x * 4 + (wave - 4);
  for (int c = 0; c < 4; ++c) out[((size_t)g * 4 + c) * 64 + lane] = sum[c];
}

template <int AGG, int VIC>
std::vector<float> run(float* d, int iters, size_t n) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<AGG, VIC>), hipFuncAttributeMaxDynamicSharedMemorySize, 84 * 1024);
  hipLaunchKernelGGL((k<AGG, VIC>), dim3(512), dim3(512), 82 * 1024, 0, d, iters);
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) printf("launch error %s\n", hipGetErrorString(e));
  std::vector<float> h(n);
  (void)hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
  return h;
}

template <int VIC>
void test(float* d, size_t n) {
  const int it = 4000;
  auto ref = run<0, VIC>(d, it, n);
  const char* names[4] = {"none", "mfma+lds", "mfma only", "lds only"};
  for (int agg = 1; agg <= 3; ++agg)
    for (int rep = 0; rep < 2; ++rep) {
      auto cur = agg == 1 ? run<1, VIC>(d, it, n) : agg == 2 ? run<2, VIC>(d, it, n) : run<3, VIC>(d, it, n);
      size_t bad = 0; int lanes[64] = {0};
      for (size_t i = 0; i < n; ++i) if (cur[i] != ref[i]) { ++bad; ++lanes[i % 64]; }
      printf("victim %s, aggressor %-9s rep %d: %zu of %zu differ", VIC == 0 ? "broadcast" : "per-lane ", names[agg], rep, bad, n);
      if (bad) { printf("; lanes:"); for (int l = 0; l < 64; ++l) if (lanes[l]) printf(" %d(%d)", l, lanes[l]); }
      printf("\n");
    }
}

int main() {
  const size_t n = (size_t)512 * 4 * 4 * 64;
  float* d; (void)hipMalloc(&d, n * 4);
  test<0>(d, n);
  test<1>(d, n);
  return 0;
}
