// Same as ldcost.hip KIND 1 (global_load_dwordx4 from L2 next to the MFMA stream) but software-pipelined inside the wave:
// the loads of iteration i+1 are issued BEFORE the 16 MFMAs of iteration i and land in a second register set.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NL>
__device__ __forceinline__ void mfma16(f32x16 (&acc)[4], const f32x4 (&v)[NL]) {
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[q % NL][q >> 2], v[(q + 1) % NL][q >> 2], acc[q & 3], 0, 0, 0);
}
template <int NL>
__global__ __launch_bounds__(256) void k(float* out, const float* g, int iters, int span) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  f32x4 va[NL], vb[NL];
  const float* gp = g + ((size_t)(blockIdx.x % 8) * 4 + wave) * 65536 + lane * 4;
#pragma unroll
  for (int i = 0; i < NL; ++i) va[i] = *(const f32x4*)(gp + (size_t)i * 256);
  for (int it = 0; it < iters; it += 2) {
#pragma unroll
    for (int i = 0; i < NL; ++i) vb[i] = *(const f32x4*)(gp + (size_t)(((it + 1) * NL + i) % span) * 256);
    __builtin_amdgcn_sched_barrier(0);
    mfma16<NL>(acc, va);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NL; ++i) va[i] = *(const f32x4*)(gp + (size_t)(((it + 2) * NL + i) % span) * 256);
    __builtin_amdgcn_sched_barrier(0);
    mfma16<NL>(acc, vb);
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0.f;
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NL>
void run(int blocks_per_cu, int span) {
  float *out, *g; (void)hipMalloc(&out, 1 << 24); (void)hipMalloc(&g, 32 * 65536 * 4); (void)hipMemset(g, 0, 32 * 65536 * 4);
  const int iters = 2048, grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<NL><<<grid, 256>>>(out, g, iters, span); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) k<NL><<<grid, 256>>>(out, g, iters, span);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double flop = (double)grid * 4 * iters * 16 * 4096.0;
  printf("pipelined global NL %d span %3d waves/SIMD %d: %.3f ms  %.1f TFLOP/s\n", NL, span, blocks_per_cu, ms, flop / ms / 1e9);
  (void)hipFree(out); (void)hipFree(g);
}
int main() {
  for (int b = 1; b <= 4; ++b) run<2>(b, 64);
  for (int b = 1; b <= 3; ++b) run<4>(b, 64);
  for (int b = 1; b <= 3; ++b) run<8>(b, 64);
  return 0;
}
