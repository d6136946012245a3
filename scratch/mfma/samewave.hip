// Does non-MFMA work interleaved in the SAME wave's instruction stream run under the MFMAs?
// per iteration: 16 v_mfma_f32_32x32x2_f32, and after each MFMA NV independent v_fma_f32 (KIND 0), or one global
// load+store pair per MFMA for KIND 1 (streaming copy, 16 dwords per lane per iteration).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int KIND, int NV>
__global__ __launch_bounds__(256) void k(float* out, int iters, const float* x, float* y) {
  const int lane = threadIdx.x & 63;
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  const float av = (float)lane, bv = 0.5f;
  float f[8];
  for (int i = 0; i < 8; ++i) f[i] = 0.001f * (lane + i);
  const size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x);
  const size_t stride = (size_t)gridDim.x * 256;
  for (int it = 0; it < iters; ++it) {
    float ld[16];
    if (KIND == 1) {
#pragma unroll
      for (int q = 0; q < 16; ++q) ld[q] = x[base + ((size_t)it * 16 + q) * stride];
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      acc[q & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[q & 3], 0, 0, 0);
      if (KIND == 0) {
#pragma unroll
        for (int v = 0; v < NV; ++v) f[v & 7] = __builtin_fmaf(f[v & 7], 1.0001f, 0.5f);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (KIND == 1) {
#pragma unroll
      for (int q = 0; q < 16; ++q) y[base + ((size_t)it * 16 + q) * stride] = 2.f * ld[q];
    }
  }
  float s = 0.f;
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  for (int i = 0; i < 8; ++i) s += f[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int KIND, int NV>
void run(const char* name, int wgs_per_cu, float* out, float* x, float* y) {
  const int iters = 1024, grid = 256 * wgs_per_cu;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<KIND, NV><<<grid, 256>>>(out, iters, x, y); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) k<KIND, NV><<<grid, 256>>>(out, iters, x, y);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double bytes = KIND == 1 ? (double)grid * 256 * iters * 16 * 8 : 0;
  printf("%-34s waves/SIMD %d: %.3f ms  (MFMA ideal %.3f ms)  %.2f TB/s\n", name, wgs_per_cu, ms, (double)iters * 16 * 64 * wgs_per_cu / 2.4e6, bytes / ms / 1e9);
}
int main() {
  float *out, *x, *y; (void)hipMalloc(&out, 1 << 24);
  const size_t n = (size_t)768 * 256 * 1024 * 16;   // floats
  (void)hipMalloc(&x, n * 4); (void)hipMalloc(&y, n * 4); (void)hipMemset(x, 0, n * 4);
  for (int w = 1; w <= 3; w += 2) {
    run<0, 0>("MFMA only", w, out, x, y);
    run<0, 2>("MFMA + 2 v_fma each", w, out, x, y);
    run<0, 4>("MFMA + 4 v_fma each", w, out, x, y);
    run<0, 8>("MFMA + 8 v_fma each", w, out, x, y);
    run<0, 14>("MFMA + 14 v_fma each", w, out, x, y);
    run<1, 0>("MFMA + 1 load + 1 store each", w, out, x, y);
  }
  return 0;
}
