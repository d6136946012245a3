// Weight-stationary feasibility: 256 weight registers per lane held in AGPRs (loaded there directly by global_load_dwordx4),
// used as the B operand of v_mfma_f32_32x32x2_f32 with the accumulators in VGPRs.  Checks rate and that the compiler adds no copies.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int Q>
__device__ __forceinline__ void mfma_ab(f32x16& acc, const f32x4& a, const f32x4& b) {
  asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a[Q]), "a"(b[Q]));
}
template <int Q>
__device__ __forceinline__ void mfma_ab0(f32x16& acc, const f32x4& a, const f32x4& b) {
  asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "=v"(acc) : "v"(a[Q]), "a"(b[Q]));
}
__global__ __launch_bounds__(256, 1) void k(const f32x4* __restrict__ w, const f32x4* __restrict__ x, float* out, unsigned long long* stamps, int iters) {
  const int lane = threadIdx.x & 63;
  f32x4 B[32][2];
#pragma unroll
  for (int c = 0; c < 32; ++c)
#pragma unroll
    for (int j = 0; j < 2; ++j) asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(B[c][j]) : "v"(w + (c * 2 + j) * 64 + lane) : "memory");
  f32x4 A[2] = {x[lane], x[64 + lane]};
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  f32x16 acc[2][2];
  float sum = 0.f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < 32; ++c) {
      if (c == 0) {
        mfma_ab0<0>(acc[0][0], A[0], B[c][0]); mfma_ab0<0>(acc[0][1], A[0], B[c][1]); mfma_ab0<0>(acc[1][0], A[1], B[c][0]); mfma_ab0<0>(acc[1][1], A[1], B[c][1]);
      } else {
        mfma_ab<0>(acc[0][0], A[0], B[c][0]); mfma_ab<0>(acc[0][1], A[0], B[c][1]); mfma_ab<0>(acc[1][0], A[1], B[c][0]); mfma_ab<0>(acc[1][1], A[1], B[c][1]);
      }
      mfma_ab<1>(acc[0][0], A[0], B[c][0]); mfma_ab<1>(acc[0][1], A[0], B[c][1]); mfma_ab<1>(acc[1][0], A[1], B[c][0]); mfma_ab<1>(acc[1][1], A[1], B[c][1]);
      mfma_ab<2>(acc[0][0], A[0], B[c][0]); mfma_ab<2>(acc[0][1], A[0], B[c][1]); mfma_ab<2>(acc[1][0], A[1], B[c][0]); mfma_ab<2>(acc[1][1], A[1], B[c][1]);
      mfma_ab<3>(acc[0][0], A[0], B[c][0]); mfma_ab<3>(acc[0][1], A[0], B[c][1]); mfma_ab<3>(acc[1][0], A[1], B[c][0]); mfma_ab<3>(acc[1][1], A[1], B[c][1]);
    }
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) sum += acc[i][j][it & 15];
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 256 + threadIdx.x] = sum;
  if (threadIdx.x == 0) stamps[blockIdx.x] = t1 - t0;
}
int main() {
  f32x4 *w, *x; float* out; unsigned long long* st;
  (void)hipMalloc(&w, 64 * 64 * 16); (void)hipMalloc(&x, 128 * 16); (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&st, 256 * 8);
  float h[64 * 64 * 4];
  for (int i = 0; i < 64 * 64 * 4; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
  (void)hipMemcpy(w, h, sizeof(h), hipMemcpyHostToDevice); (void)hipMemcpy(x, h, 128 * 16, hipMemcpyHostToDevice);
  const int iters = 2000;
  for (int r = 0; r < 3; ++r) k<<<256, 256>>>(w, x, out, st, iters);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  k<<<256, 256>>>(w, x, out, st, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long hs[256]; (void)hipMemcpy(hs, st, sizeof(hs), hipMemcpyDeviceToHost);
  float ho[4]; (void)hipMemcpy(ho, out, 16, hipMemcpyDeviceToHost);
  printf("weight-stationary MFMA loop: %.3f ms, %.1f TFLOP/s, %.2f cycles per MFMA (64 = pipe rate); out[0] = %g\n", ms,
         256.0 * 4 * iters * 512 * 4096.0 / ms / 1e9, (double)hs[0] / iters / 512, ho[0]);
  return 0;
}
