// Do memory-streaming waves disturb MFMA waves on the same SIMD?  waves 0-3: MFMA stream (registers only);
// waves 4-7: streaming y[i] = 2*x[i] over a big array (dword per lane, 8 loads in flight per wave).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int PRIO>
__global__ __launch_bounds__(512) void k(float* out, int iters, const float* x, float* y, size_t n_per_block) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave < 4) {
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const float av = (float)lane, bv = 0.5f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[q & 3], 0, 0, 0);
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  } else {
    if (PRIO) __builtin_amdgcn_s_setprio(3);
    const size_t base = (size_t)blockIdx.x * n_per_block;
    const int t = threadIdx.x - 256;
    for (size_t i = t; i + 7 * 256 < n_per_block; i += 8 * 256) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = x[base + i + u * 256];
#pragma unroll
      for (int u = 0; u < 8; ++u) y[base + i + u * 256] = 2.f * v[u];
    }
  }
}
template <int PRIO>
void run(const char* name, int iters, size_t n_per_block, float* out, float* x, float* y) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<PRIO><<<256, 512>>>(out, iters, x, y, n_per_block); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) k<PRIO><<<256, 512>>>(out, iters, x, y, n_per_block);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  printf("prio %d %-28s mfma iters %5d bytes %5.0f MB: %.3f ms  (MFMA ideal %.3f ms, %.2f TB/s)\n", PRIO, name, iters, 256.0 * n_per_block * 8 / 1e6, ms,
         (double)iters * 16 * 64 / 2.4e6, 256.0 * n_per_block * 8 / ms / 1e9);
}
int main() {
  float *out, *x, *y; (void)hipMalloc(&out, 1 << 24);
  const size_t n = (size_t)256 << 20;   // 1 GiB each
  (void)hipMalloc(&x, n * 4); (void)hipMalloc(&y, n * 4); (void)hipMemset(x, 0, n * 4);
  run<0>("MFMA only", 2048, 0, out, x, y);
  run<0>("copy only (512 MB)", 0, 262144, out, x, y);
  run<0>("MFMA + copy 512 MB", 2048, 262144, out, x, y);
  run<0>("copy only (1 GB)", 0, 524288, out, x, y);
  run<0>("MFMA + copy 1 GB", 2048, 524288, out, x, y);
  run<0>("MFMA(2x) + copy 1 GB", 4096, 524288, out, x, y);
  run<1>("MFMA + copy 512 MB", 2048, 262144, out, x, y);
  run<1>("MFMA + copy 1 GB", 2048, 524288, out, x, y);
  run<1>("MFMA(2x) + copy 1 GB", 4096, 524288, out, x, y);
  return 0;
}
