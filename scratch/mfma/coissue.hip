#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
// waves 0-3: MFMA stream (registers only); waves 4-7: VALU stream (fma + exp) or LDS-write stream; report both rates
template <int KIND>
__global__ __launch_bounds__(512) void k(float* out, int iters, int valu_iters) {
  __shared__ float lds[16384];
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
    float x = (float)lane * 0.001f, y = 1.0f;
    if (KIND == 1) {
      for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) { y = __builtin_fmaf(x, y, 0.5f); x = __expf(-y); }
      }
    } else if (KIND == 2) {
      for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) { *(float4*)&lds[((wave - 4) * 64 + lane) * 4 + u * 1024] = make_float4(x, y, x, y); y += 1.f; }
      }
      x = lds[lane];
    }
    out[blockIdx.x * 512 + threadIdx.x] = x + y;
  }
}
template <int KIND>
void run(const char* name, int iters, int valu_iters) {
  float* out; hipMalloc(&out, 1 << 24);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<KIND><<<256, 512>>>(out, iters, valu_iters); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) k<KIND><<<256, 512>>>(out, iters, valu_iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  printf("%-40s mfma iters %5d side iters %6d: %.3f ms  (MFMA-only ideal %.3f ms)\n", name, iters, valu_iters, ms, (double)iters * 16 * 64 / 2.4e6);
  hipFree(out);
}
int main() {
  run<0>("MFMA waves only", 4096, 0);
  run<1>("side waves: 16x(fma+exp) per iter", 0, 4096);
  run<1>("MFMA + VALU side waves", 4096, 4096);
  run<1>("MFMA + VALU side waves (2x VALU)", 4096, 8192);
  run<2>("side waves: 16x ds_write_b128 per iter", 0, 4096);
  run<2>("MFMA + LDS-write side waves", 4096, 4096);
  return 0;
}
