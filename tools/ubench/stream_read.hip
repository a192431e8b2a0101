// Achievable HBM read rate of the batch-major stream pattern (k_residual_bm / k_lbs_partsum_bm): a wave walks
// 6 sequential streams (3 coordinates x {posed vertices, targets}) of one 64-instance block, one 256-byte
// row per stream per vertex.  Variants: dwords per lane per load (1 = the product layout, 2, 4 = several
// vertices per lane-load), loads in flight per wave (steps issued before the first use), workgroups per CU
// (limited through dynamic LDS), and a plain linear dwordx4 read of the same footprint for reference.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float fv2 __attribute__((ext_vector_type(2)));
typedef float fv4 __attribute__((ext_vector_type(4)));

template <int VEC> struct V { };
template <> struct V<1> { typedef float t; static __device__ float sum(float v) { return v; } };
template <> struct V<2> { typedef fv2 t; static __device__ float sum(fv2 v) { return v.x + v.y; } };
template <> struct V<4> { typedef fv4 t; static __device__ float sum(fv4 v) { return (v.x + v.y) + (v.z + v.w); } };

// grid (groups, blocks), block 256 (4 waves); wave w of group g reads vertices [ (g*4+w)*per, +per )
template <int VEC, int DEPTH>
__global__ __launch_bounds__(256) void k_streams(const float* __restrict__ base, float* __restrict__ out, int Vp, int per) {
  extern __shared__ float lds[];
  typedef typename V<VEC>::t T;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t cstr = (size_t)Vp * 64;
  const float* blk = base + (size_t)blockIdx.y * 6 * cstr;
  const int v0 = (blockIdx.x * 4 + wave) * per;
  const T* p = reinterpret_cast<const T*>(blk + (size_t)v0 * 64) + lane;
  const size_t cs = cstr / VEC;
  float acc = 0.f;
  const int nstep = per / VEC;  // a step = VEC vertices of the 6 streams
  for (int s = 0; s < nstep; s += DEPTH) {
    T r[DEPTH][6];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
      for (int c = 0; c < 6; ++c) r[d][c] = (s + d < nstep) ? p[(size_t)c * cs + (size_t)(s + d) * 64] : T(0);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
      for (int c = 0; c < 6; ++c) acc += V<VEC>::sum(r[d][c]);
  }
  if (acc == 12345.678f) out[0] = acc + lds[0];
}

// reference: every workgroup reads one contiguous run with dwordx4, 4 loads in flight
__global__ __launch_bounds__(256) void k_linear(const fv4* __restrict__ base, float* __restrict__ out, size_t n4) {
  const size_t per = n4 / gridDim.x;
  const fv4* p = base + (size_t)blockIdx.x * per + threadIdx.x;
  float acc = 0.f;
  for (size_t i = 0; i + 1024 <= per; i += 1024) {
    fv4 a = p[i], b = p[i + 256], c = p[i + 512], d = p[i + 768];
    acc += (a.x + b.y) + (c.z + d.w);
  }
  if (acc == 12345.678f) out[0] = acc;
}

static float* d_base; static float* d_out; static int g_Vp = 6912, g_blocks = 64;

template <int VEC, int DEPTH>
void run(int wg_per_cu, int groups, int per) {
  const size_t lds = wg_per_cu >= 8 ? 0 : (size_t)(160 * 1024 / wg_per_cu - 1024);
  (void)hipFuncSetAttribute((const void*)k_streams<VEC, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k_streams<VEC, DEPTH>), dim3(groups, g_blocks), dim3(256), lds, 0, d_base, d_out, g_Vp, per);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  const double bytes = (double)g_blocks * 6 * groups * 4 * per * 256;
  printf("streams vec=%d depth=%d wg/cu=%d grid=%dx%d per=%d: %7.1f us  %6.0f GB/s\n", VEC, DEPTH, wg_per_cu, groups, g_blocks, per,
         best * 1e3, bytes / (best * 1e-3) / 1e9);
}

int main() {
  const size_t n = (size_t)g_blocks * 6 * g_Vp * 64;
  (void)hipMalloc(&d_base, n * 4); (void)hipMalloc(&d_out, 64);
  (void)hipMemset(d_base, 0, n * 4);
  printf("footprint %.0f MB\n", n * 4 / 1e6);
  {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int grid : {1024, 2048, 4096, 8192}) {
      float best = 1e9f;
      for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_linear, dim3(grid), dim3(256), 0, 0, (const fv4*)d_base, d_out, n / 4);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
      }
      printf("linear dwordx4 grid=%d: %7.1f us  %6.0f GB/s\n", grid, best * 1e3, n * 4.0 / (best * 1e-3) / 1e9);
    }
  }
  // the product's shape: 30 groups x 64 blocks of 4 waves, ~57 vertices per wave, 4 workgroups per CU
  for (int wg : {2, 4, 8}) {
    run<1, 1>(wg, 30, 56); run<1, 2>(wg, 30, 56); run<1, 4>(wg, 30, 56); run<1, 8>(wg, 30, 56);
    run<2, 1>(wg, 30, 56); run<2, 2>(wg, 30, 56); run<2, 4>(wg, 30, 56);
    run<4, 1>(wg, 30, 56); run<4, 2>(wg, 30, 56); run<4, 4>(wg, 30, 56);
  }
  // fewer, longer waves (one wave owns 4x the vertices) and more, shorter ones
  run<1, 2>(4, 8, 216); run<1, 4>(4, 8, 216); run<4, 2>(4, 8, 216);
  run<1, 2>(8, 108, 16); run<1, 4>(8, 108, 16); run<4, 2>(8, 108, 16); run<4, 4>(8, 108, 16);
  return 0;
}
