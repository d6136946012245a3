// What is the summation order INSIDE an FP32 MFMA?  Compares v_mfma_f32_32x32x2_f32 and v_mfma_f32_16x16x4_f32 on random operands with
// host models: a chain of fused multiply-adds over k ascending / descending, and a single rounding of the exact sum.
// Needed to give the 16 x 16 and 32 x 32 per-image blocks of linear_small.hip the same bits (round 4).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k32(const float* A, const float* B, const float* C, float* D) {   // A [32][2], B [2][32], C, D [32][32]
  const int l = threadIdx.x, i = l & 31, h = l >> 5;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = C[((r / 4) * 8 + h * 4 + (r % 4)) * 32 + i];
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * 2 + h], B[h * 32 + i], acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[((r / 4) * 8 + h * 4 + (r % 4)) * 32 + i] = acc[r];
}
__global__ void k16(const float* A, const float* B, const float* C, float* D) {   // A [16][4], B [4][16], C, D [16][16]
  const int l = threadIdx.x, i = l & 15, q = l >> 4;
  f32x4 acc;
  for (int r = 0; r < 4; ++r) acc[r] = C[(4 * q + r) * 16 + i];
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i * 4 + q], B[q * 16 + i], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(4 * q + r) * 16 + i] = acc[r];
}
// the same 16 x 16 x 4 product as two 32 x 32 x 2 MFMAs (rows / columns 16 .. 31 zero): k pairs (p0, p1) then (p2, p3)
__global__ void k32as16(const float* A, const float* B, const float* C, float* D, int p0, int p1, int p2, int p3) {
  const int l = threadIdx.x, i = l & 31, h = l >> 5;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) {
    const int row = (r / 4) * 8 + h * 4 + (r % 4);
    acc[r] = (row < 16 && i < 16) ? C[row * 16 + i] : 0.f;
  }
  const int ka = h ? p1 : p0, kb = h ? p3 : p2;
  float a0 = i < 16 ? A[i * 4 + ka] : 0.f, b0 = i < 16 ? B[ka * 16 + i] : 0.f;
  float a1 = i < 16 ? A[i * 4 + kb] : 0.f, b1 = i < 16 ? B[kb * 16 + i] : 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) {
    const int row = (r / 4) * 8 + h * 4 + (r % 4);
    if (row < 16 && i < 16) D[row * 16 + i] = acc[r];
  }
}

static float rnd() {
  const float m = (float)rand() / RAND_MAX * 2.f - 1.f;
  return ldexpf(m, rand() % 9 - 4);
}
static unsigned bits(float f) { unsigned u; memcpy(&u, &f, 4); return u; }

int main() {
  srand(1);
  float *dA, *dB, *dC, *dD;
  (void)hipMalloc(&dA, 4096); (void)hipMalloc(&dB, 4096); (void)hipMalloc(&dC, 4096); (void)hipMalloc(&dD, 4096);
  long n = 0, asc = 0, desc = 0, exact = 0, unf = 0;
  long n16 = 0, asc16 = 0, desc16 = 0, exact16 = 0, pair16 = 0, eq0123 = 0, eq0213 = 0, eq1032 = 0;
  for (int trial = 0; trial < 200; ++trial) {
    float A[64], B[64], C[1024], D[1024];
    for (int i = 0; i < 64; ++i) A[i] = rnd(), B[i] = rnd();
    for (int i = 0; i < 1024; ++i) C[i] = trial % 2 ? rnd() : rnd() * 1e-3f;
    (void)hipMemcpy(dA, A, 256, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B, 256, hipMemcpyHostToDevice); (void)hipMemcpy(dC, C, 4096, hipMemcpyHostToDevice);
    k32<<<1, 64>>>(dA, dB, dC, dD);
    (void)hipMemcpy(D, dD, 4096, hipMemcpyDeviceToHost);
    for (int i = 0; i < 32; ++i)
      for (int j = 0; j < 32; ++j) {
        const float a0 = A[i * 2], a1 = A[i * 2 + 1], b0 = B[j], b1 = B[32 + j], c = C[i * 32 + j], d = D[i * 32 + j];
        ++n;
        asc += bits(d) == bits(fmaf(a1, b1, fmaf(a0, b0, c)));
        desc += bits(d) == bits(fmaf(a0, b0, fmaf(a1, b1, c)));
        exact += bits(d) == bits((float)((long double)c + (long double)a0 * b0 + (long double)a1 * b1));
        unf += bits(d) == bits((c + a0 * b0) + a1 * b1);
      }
    // 16 x 16 x 4 on the first 64 entries of A / B, first 256 of C
    k16<<<1, 64>>>(dA, dB, dC, dD);
    (void)hipMemcpy(D, dD, 1024, hipMemcpyDeviceToHost);
    float D2[256];
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        float a[4], b[4];
        for (int k = 0; k < 4; ++k) a[k] = A[i * 4 + k], b[k] = B[k * 16 + j];
        const float c = C[i * 16 + j], d = D[i * 16 + j];
        ++n16;
        asc16 += bits(d) == bits(fmaf(a[3], b[3], fmaf(a[2], b[2], fmaf(a[1], b[1], fmaf(a[0], b[0], c)))));
        desc16 += bits(d) == bits(fmaf(a[0], b[0], fmaf(a[1], b[1], fmaf(a[2], b[2], fmaf(a[3], b[3], c)))));
        long double e = c;
        for (int k = 0; k < 4; ++k) e += (long double)a[k] * b[k];
        exact16 += bits(d) == bits((float)e);
        pair16 += bits(d) == bits(fmaf(a[3], b[3], fmaf(a[1], b[1], fmaf(a[2], b[2], fmaf(a[0], b[0], c)))));
      }
    float D16[256];
    memcpy(D16, D, 1024);
    const int perms[3][4] = {{0, 1, 2, 3}, {0, 2, 1, 3}, {1, 0, 3, 2}};
    long* cnt[3] = {&eq0123, &eq0213, &eq1032};
    for (int p = 0; p < 3; ++p) {
      k32as16<<<1, 64>>>(dA, dB, dC, dD, perms[p][0], perms[p][1], perms[p][2], perms[p][3]);
      (void)hipMemcpy(D2, dD, 1024, hipMemcpyDeviceToHost);
      for (int e = 0; e < 256; ++e) *cnt[p] += bits(D2[e]) == bits(D16[e]);
    }
  }
  printf("32x32x2: n %ld  fma-asc %ld  fma-desc %ld  exact-sum %ld  unfused-asc %ld\n", n, asc, desc, exact, unf);
  printf("16x16x4: n %ld  fma-asc %ld  fma-desc %ld  exact-sum %ld  fma-0213 %ld\n", n16, asc16, desc16, exact16, pair16);
  printf("16x16x4 == two 32x32x2 with k pairs (0,1)(2,3): %ld  (0,2)(1,3): %ld  (1,0)(3,2): %ld of %ld\n", eq0123, eq0213, eq1032, n16);
  return 0;
}
