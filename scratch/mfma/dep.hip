// Dependent-accumulator distance test: 16 v_mfma_f32_32x32x2_f32 per iteration cycling through NACC accumulators.
// REP = how many consecutive MFMAs hit the same accumulator before moving on.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC, int REP>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  const int lane = threadIdx.x & 63;
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  float av = (float)lane, bv = 0.5f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 32; ++q) acc[(q / REP) % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[(q / REP) % NACC], 0, 0, 0);
  }
  float s = 0.f;
  for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC, int REP>
void run(int blocks_per_cu) {
  float* out; (void)hipMalloc(&out, 1 << 24);
  const int iters = 1024, grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<NACC, REP><<<grid, 256>>>(out, iters); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) k<NACC, REP><<<grid, 256>>>(out, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double flop = (double)grid * 4 * iters * 32 * 4096.0;
  printf("NACC %d REP %d waves/SIMD %d: %.3f ms  %.1f TFLOP/s\n", NACC, REP, blocks_per_cu, ms, flop / ms / 1e9);
  (void)hipFree(out);
}
template <int NACC, int REP> void sweep() { for (int b = 1; b <= 4; ++b) run<NACC, REP>(b); }
int main() {
  sweep<1, 1>(); sweep<2, 1>(); sweep<3, 1>(); sweep<4, 1>(); sweep<6, 1>(); sweep<8, 1>();
  sweep<4, 2>(); sweep<4, 4>(); sweep<8, 2>(); sweep<8, 4>();
  return 0;
}
