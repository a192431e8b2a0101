// Achievable HBM write rate of the GEMM's output pattern: a wave writes 16 rows of 256 bytes per 32x32 tile
// (rows 256 B apart inside an instance block), tiles visited column-tile after column-tile; and a plain linear
// dwordx4 write of the same footprint for reference.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float fv4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_linear(fv4* __restrict__ base, size_t n4) {
  const size_t per = n4 / gridDim.x;
  fv4* p = base + (size_t)blockIdx.x * per + threadIdx.x;
  fv4 v; v.x = threadIdx.x; v.y = 1.f; v.z = 2.f; v.w = 3.f;
  for (size_t i = 0; i + 1024 <= per; i += 1024) { p[i] = v; p[i + 256] = v; p[i + 512] = v; p[i + 768] = v; }
}

// grid (nchunk, nblk64/4): 8 waves, wave w writes instance half-block (blockIdx.y*8+w) (32 instances = 128 B per row)
// as the transposed GEMM does: C[(m0>>6)][n][64] with 32 consecutive instances per half wave
__global__ __launch_bounds__(512) void k_gemm_like(float* __restrict__ C, int N, int tiles_per_chunk) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, kg = lane >> 5;
  const int m0 = (blockIdx.y * 8 + wave) * 32, ntiles = N / 32;
  const int t0 = blockIdx.x * tiles_per_chunk, t1 = min(t0 + tiles_per_chunk, ntiles);
  for (int t = t0; t < t1; ++t) {
    float* ccol = C + ((size_t)(m0 >> 6) * N + t * 32) * 64 + (m0 & 63) + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) ccol[(size_t)((r & 3) + 8 * (r >> 2) + 4 * kg) * 64] = (float)(t + r);
  }
}

int main() {
  const int Mp = 4096, N = 3 * 6912;
  const size_t n = (size_t)Mp * N;
  float* d; (void)hipMalloc(&d, n * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  printf("footprint %.0f MB\n", n * 4 / 1e6);
  for (int grid : {1024, 2048, 4096}) {
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k_linear, dim3(grid), dim3(256), 0, 0, (fv4*)d, n / 4);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      if (rep && ms < best) best = ms;
    }
    printf("linear dwordx4 write grid=%d: %7.1f us  %6.0f GB/s\n", grid, best * 1e3, n * 4.0 / (best * 1e-3) / 1e9);
  }
  for (int nchunk : {16, 32, 64}) {
    const int ny = Mp / 256, ntiles = N / 32, per = (ntiles + nchunk - 1) / nchunk;
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k_gemm_like, dim3(nchunk, ny), dim3(512), 0, 0, d, N, per);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      if (rep && ms < best) best = ms;
    }
    printf("gemm-like writes nchunk=%d (grid %dx%d): %7.1f us  %6.0f GB/s\n", nchunk, nchunk, ny, best * 1e3, n * 4.0 / (best * 1e-3) / 1e9);
  }
  return 0;
}
