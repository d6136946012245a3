// Cost of operand loads next to an MFMA stream: per iteration 16 v_mfma_f32_32x32x2_f32 (4 accumulators) plus
// NL ds_read_b128 (KIND 0) or NL global_load_dwordx4 (KIND 1, L1/L2-resident) whose results feed the MFMAs.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int KIND, int NL>
__global__ __launch_bounds__(256) void k(float* out, const float* g, int iters, int span) {
  __shared__ float lds[64 * 68 * 2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 64 * 68 * 2; i += blockDim.x) lds[i] = (float)(i % 7) * 0.01f;
  __syncthreads();
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  f32x4 v[8];
  for (int i = 0; i < 8; ++i) v[i] = f32x4{1.f, 2.f, 3.f, 4.f} * (float)(lane + i);
  const float* lp = &lds[(lane & 31) * 68 + (lane >> 5) * 4];
  const float* gp = g + ((size_t)(blockIdx.x % 8) * 4 + wave) * 65536 + lane * 4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      if (KIND == 0) v[i] = *(const f32x4*)(lp + ((it + i) & 7) * 8 + (i & 1) * 32 * 68);
      else v[i] = *(const f32x4*)(gp + (size_t)((it * NL + i) % span) * 256);
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[q & 7][q >> 2], v[(q + 3) & 7][q >> 2], acc[q & 3], 0, 0, 0);
  }
  float s = 0.f;
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND, int NL>
void run(int blocks_per_cu, int span) {
  float *out, *g; (void)hipMalloc(&out, 1 << 24); (void)hipMalloc(&g, 32 * 65536 * 4); (void)hipMemset(g, 0, 32 * 65536 * 4);
  const int iters = 2048, grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<KIND, NL><<<grid, 256>>>(out, g, iters, span); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) k<KIND, NL><<<grid, 256>>>(out, g, iters, span);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double flop = (double)grid * 4 * iters * 16 * 4096.0;
  printf("%s NL %d span %3d waves/SIMD %d: %.3f ms  %.1f TFLOP/s\n", KIND ? "global" : "lds   ", NL, span, blocks_per_cu, ms, flop / ms / 1e9);
  (void)hipFree(out); (void)hipFree(g);
}
template <int KIND, int NL> void sweep(int span) { run<KIND, NL>(1, span); run<KIND, NL>(3, span); }
int main() {
  sweep<0, 0>(1); sweep<0, 1>(1); sweep<0, 2>(1); sweep<0, 4>(1); sweep<0, 8>(1);
  sweep<1, 1>(4); sweep<1, 2>(4); sweep<1, 4>(4); sweep<1, 8>(4);       // 4 KiB per wave: L1 resident
  sweep<1, 2>(64); sweep<1, 4>(64); sweep<1, 8>(64);                    // 64 KiB per wave, 256 KiB per block: L2
  return 0;
}
