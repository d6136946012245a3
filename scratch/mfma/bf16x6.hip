// Beyond the FP32-MFMA ceiling (round 4, planning data for DESIGN section 6): FP32 products EMULATED on the BF16 matrix cores.
// a = a_h + a_m + a_l (three bf16 pieces, 24 mantissa bits), same for b; the six products h.h, h.m, m.h, m.m, h.l, l.h carry everything
// down to 2^-24 |a||b| and are exact in fp32; accumulation is fp32 inside the MFMA.  v_mfma_f32_32x32x16_bf16 does 16 k per 32 cycles
// against 2 k per 64 cycles of v_mfma_f32_32x32x2_f32: 6 / 16 of the matrix time for the same product.
//   part 1: accuracy of one 32 x 32 x 256 product (fp32 MFMA chain vs bf16x6 vs float64 on the host)
//   part 2: throughput of the bf16 MFMA alone, and with the vector-ALU work of splitting one operand on the fly beside it
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ inline void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)x; const float r = x - (float)h;
  m = (__bf16)r; l = (__bf16)(r - (float)m);
}

// A [32][256], B [256][32] row-major fp32; D32 / D6: [32][32]
__global__ void acc_test(const float* A, const float* B, float* D32, float* D6) {
  const int l = threadIdx.x, i = l & 31, h = l >> 5;
  f32x16 acc = {0};
  for (int k = 0; k < 256; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * 256 + k + h], B[(k + h) * 32 + i], acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D32[((r / 4) * 8 + h * 4 + (r % 4)) * 32 + i] = acc[r];
  f32x16 c = {0};
  for (int k = 0; k < 256; k += 16) {
    bf16x8 ah, am, al, bh, bm, bl;
    for (int j = 0; j < 8; ++j) {
      __bf16 x, y, z;
      split3(A[i * 256 + k + 8 * h + j], x, y, z); ah[j] = x; am[j] = y; al[j] = z;
      split3(B[(k + 8 * h + j) * 32 + i], x, y, z); bh[j] = x; bm[j] = y; bl[j] = z;
    }
    // smallest terms first
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) D6[((r / 4) * 8 + h * 4 + (r % 4)) * 32 + i] = c[r];
}

// NV: vector-ALU instructions interleaved per MFMA (0: none)
template <int NV>
__global__ __launch_bounds__(256) void rate(float* out, int iters, float seed) {
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  bf16x8 av, bv;
  for (int j = 0; j < 8; ++j) { av[j] = (__bf16)(seed + threadIdx.x + j); bv[j] = (__bf16)(0.5f + j); }
  float v0 = seed, v1 = seed * 2, v2 = seed * 3, v3 = seed * 5;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      acc[q & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[q & 3], 0, 0, 0);
#pragma unroll
      for (int n = 0; n < NV; ++n) {
        if ((n & 3) == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v0));
        if ((n & 3) == 1) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v1));
        if ((n & 3) == 2) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v2));
        if ((n & 3) == 3) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v3));
      }
    }
  }
  float s = v0 + v1 + v2 + v3;
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NV>
void run_rate() {
  float* out; (void)hipMalloc(&out, 1 << 22);
  const int iters = 2048, grid = 256;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  rate<NV><<<grid, 256>>>(out, iters, 1.f); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) rate<NV><<<grid, 256>>>(out, iters, 1.f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double flop = (double)grid * 4 * iters * 16 * 2.0 * 32 * 32 * 16;
  printf("bf16 32x32x16 MFMA + %d v_fma per MFMA: %.3f ms  %.0f TFLOP/s (bf16)  = %.0f TFLOP/s of emulated fp32 at 6 products\n", NV, ms, flop / ms / 1e9, flop / ms / 1e9 / 6);
  (void)hipFree(out);
}


// the real instruction mix of cutting fp32 values into three bf16 pieces, NE elements per MFMA
template <int NE>
__global__ __launch_bounds__(256) void rate_split(float* out, int iters, float seed) {
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  bf16x8 av, bv;
  for (int j = 0; j < 8; ++j) { av[j] = (__bf16)(seed + threadIdx.x + j); bv[j] = (__bf16)(0.5f + j); }
  float x[4] = {seed * 1.1f + threadIdx.x, seed * 1.3f, seed * 1.7f, seed * 1.9f};
  bf16x8 h = av, m = av, l = av;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      acc[q & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[q & 3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        const float v = x[(q + e) & 3];
        const __bf16 xh = (__bf16)v; const float r1 = v - (float)xh;
        const __bf16 xm = (__bf16)r1; const float r2 = r1 - (float)xm;
        h[(q + e) & 7] = xh; m[(q + e) & 7] = xm; l[(q + e) & 7] = (__bf16)r2;
        x[(q + e) & 3] = v * 1.0001f + r2;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = x[0] + x[1] + x[2] + x[3];
  for (int j = 0; j < 8; ++j) s += (float)h[j] + (float)m[j] + (float)l[j];
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NE>
void run_split() {
  float* out; (void)hipMalloc(&out, 1 << 22);
  const int iters = 2048, grid = 256;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  rate_split<NE><<<grid, 256>>>(out, iters, 1.f); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) rate_split<NE><<<grid, 256>>>(out, iters, 1.f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double flop = (double)grid * 4 * iters * 16 * 2.0 * 32 * 32 * 16;
  printf("bf16 MFMA + %d three-piece splits per MFMA: %.3f ms  %.0f TFLOP/s (bf16)\n", NE, ms, flop / ms / 1e9);
  (void)hipFree(out);
}

int main() {
  srand(3);
  static float A[32 * 256], B[256 * 32], D32[1024], D6[1024];
  for (int i = 0; i < 32 * 256; ++i) { A[i] = ((float)rand() / RAND_MAX * 2 - 1) * 3; B[i] = ((float)rand() / RAND_MAX * 2 - 1) / 16; }
  float *dA, *dB, *d32, *d6;
  (void)hipMalloc(&dA, sizeof(A)); (void)hipMalloc(&dB, sizeof(B)); (void)hipMalloc(&d32, 4096); (void)hipMalloc(&d6, 4096);
  (void)hipMemcpy(dA, A, sizeof(A), hipMemcpyHostToDevice); (void)hipMemcpy(dB, B, sizeof(B), hipMemcpyHostToDevice);
  acc_test<<<1, 64>>>(dA, dB, d32, d6);
  (void)hipMemcpy(D32, d32, 4096, hipMemcpyDeviceToHost); (void)hipMemcpy(D6, d6, 4096, hipMemcpyDeviceToHost);
  double e32 = 0, e6 = 0, sc = 0, r32 = 0, r6 = 0;
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      double ref = 0;
      for (int k = 0; k < 256; ++k) ref += (double)A[i * 256 + k] * B[k * 32 + j];
      sc = fmax(sc, fabs(ref));
      e32 = fmax(e32, fabs(D32[i * 32 + j] - ref)); e6 = fmax(e6, fabs(D6[i * 32 + j] - ref));
      r32 += (D32[i * 32 + j] - ref) * (D32[i * 32 + j] - ref); r6 += (D6[i * 32 + j] - ref) * (D6[i * 32 + j] - ref);
    }
  printf("32x32x256 product vs float64: fp32 MFMA max %.3e rms %.3e | bf16x6 max %.3e rms %.3e (relative to max |result| %.3f)\n", e32 / sc, sqrt(r32 / 1024) / sc,
         e6 / sc, sqrt(r6 / 1024) / sc, sc);
  run_rate<0>(); run_rate<1>(); run_rate<2>(); run_rate<4>(); run_rate<8>();
  run_split<0>(); run_split<1>(); run_split<2>();
  return 0;
}
