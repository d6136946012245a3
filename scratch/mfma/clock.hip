// Sustained FP32-MFMA rate and in-kernel clock on random operands: v_mfma_f32_32x32x2_f32 vs v_mfma_f32_16x16x4_f32, one wave per SIMD,
// with and without an HBM stream beside it (MI355X_MICROARCH.md "DVFS give-back": the chip lowers its clock under load).
// clock = d(s_memtime) / d(s_memrealtime) * 100 MHz, stamped around the loop after >= 2 s of back-to-back launches.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int STREAM>   // SHAPE 0: 32x32x2 (4 accumulators), 1: 16x16x4 (16 accumulators); STREAM: float4 loads per 16 MFMA-slots
__global__ __launch_bounds__(256, 1) void k(const float* __restrict__ rnd, const f32x4* __restrict__ big, size_t big_n4, float* out,
                                             unsigned long long* stamps, int iters) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = rnd[(blockIdx.x * 256 + threadIdx.x) * 16 + i]; b[i] = rnd[(blockIdx.x * 256 + threadIdx.x) * 16 + 8 + i]; }
  f32x16 acc32[4];
  f32x4 acc16[16];
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc32[j][r] = 0.f;
  for (int j = 0; j < 16; ++j) for (int r = 0; r < 4; ++r) acc16[j][r] = 0.f;
  f32x4 sink = {0.f, 0.f, 0.f, 0.f};
  size_t p = ((size_t)blockIdx.x * 256 + threadIdx.x);
  const size_t stride = (size_t)gridDim.x * 256;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if (STREAM) {
#pragma unroll
      for (int s = 0; s < STREAM; ++s) { f32x4 v = big[p % big_n4]; p += stride; sink += v; }
    }
    if (SHAPE == 0) {
#pragma unroll
      for (int q = 0; q < 16; ++q) acc32[q & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q & 7], b[(q >> 1) & 7], acc32[q & 3], 0, 0, 0);
    } else {
#pragma unroll
      for (int q = 0; q < 32; ++q) acc16[q & 15] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q & 7], b[(q >> 2) & 7], acc16[q & 15], 0, 0, 0);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = sink[0] + sink[1] + sink[2] + sink[3];
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc32[j][r];
  for (int j = 0; j < 16; ++j) for (int r = 0; r < 4; ++r) s += acc16[j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (lane == 0 && wave == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int SHAPE, int STREAM>
void run(const float* rnd, const f32x4* big, size_t n4, float* out, unsigned long long* stamps) {
  const int grid = 256, iters = 4096;   // 4096 * 16 * 64 cycles = 4.2 M cycles ~ 2 ms per launch
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int w = 0; w < 1000; ++w) k<SHAPE, STREAM><<<grid, 256>>>(rnd, big, n4, out, stamps, iters);   // ~2 s warm
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  const int reps = 100;
  for (int r = 0; r < reps; ++r) k<SHAPE, STREAM><<<grid, 256>>>(rnd, big, n4, out, stamps, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  std::vector<unsigned long long> h(2 * grid);
  (void)hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
  double clk = 0; for (int b = 0; b < grid; ++b) clk += (double)h[2 * b] / (double)h[2 * b + 1] * 100.0; clk /= grid;
  const double flop = (double)grid * 4 * iters * 16 * 4096.0;
  const double bytes = (double)grid * 256 * iters * STREAM * 16.0;
  printf("%s stream %d float4/iter: %.3f ms  %.1f TFLOP/s  in-kernel clock %.0f MHz  cycles/iter %.0f  HBM %.2f TB/s\n", SHAPE ? "16x16x4" : "32x32x2", STREAM, ms,
         flop / ms / 1e9, clk, (double)h[0] / iters, bytes / ms / 1e9);
}
int main() {
  const size_t n4 = (size_t)1 << 26;   // 1 GiB
  float* rnd; f32x4* big; float* out; unsigned long long* stamps;
  (void)hipMalloc(&rnd, 256 * 256 * 16 * 4); (void)hipMalloc(&big, n4 * 16); (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&stamps, 4096 * 8);
  std::vector<float> h(256 * 256 * 16);
  srand(1);
  for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  (void)hipMemcpy(rnd, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemset(big, 0x3c, n4 * 16);
  run<0, 0>(rnd, big, n4, out, stamps);
  run<1, 0>(rnd, big, n4, out, stamps);
  run<0, 2>(rnd, big, n4, out, stamps);
  run<1, 2>(rnd, big, n4, out, stamps);
  run<0, 4>(rnd, big, n4, out, stamps);
  run<1, 4>(rnd, big, n4, out, stamps);
  run<0, 0>(rnd, big, n4, out, stamps);
  return 0;
}
