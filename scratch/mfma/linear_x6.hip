// PROTOTYPE v3 (round 4, planning data for DESIGN section 6 - not product code): the N-row forward layer of linear_x9.hip,
//     Y[M, 256] = softplus(X[M, 256] W^T + b),        M = 131072 at BASELINE config #2,
// with SIX piece products (h l, m m, h m, l h, m h, h h: without l l, m l, l m) and TWO 128-row workgroups per CU: a wave owns 32 rows x all
// 256 columns (128 accumulator registers, <= 256 registers in all), so every SIMD holds two waves of different workgroups and one's loads,
// cuts, barrier and epilogue run beside the other's MFMA stream - no hand-interleaving of those, only the B-fragment reads sit in the stream.
// Per k step (16 k): the wave requests the next slab (24 KiB of cut weights, 6 x 16 B per thread) and its next 8 activation values, runs
// 48 MFMAs (column blocks in two halves of four, piece products grouped by B plane: l, m, m, h, h, h, two 16-register fragment buffers),
// then stores the slab to the other LDS buffer, cuts the activation values and meets the other waves at ONE barrier.
// Measured on an MI355X (warm clocks): whole layer 158 us; -DNO_EPI 128; + -DV_NOSLAB 121; + -DV_NOBAR 119 - slower than linear_x9.hip's
// one-wave-per-SIMD six-product build (142 / 116 / 87): a B fragment now feeds ONE MFMA, so the LDS delivers 24 KiB per 48 MFMAs and wave
// (half of its bandwidth with eight waves per CU), and the second wave per SIMD hides less of the epilogue than hoped (30 us remain).
// Neither shape beats the FP32-MFMA kernel's 140 us: see DESIGN section 6.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int K = 256, NOUT = 256, KS = K / 16, NCB = NOUT / 32;
constexpr int SLAB_BYTES = 3 * NCB * 64 * 16;      // one k step of the cut weights: 24 KiB

struct Cut3 { unsigned h, m, l; };
__host__ __device__ inline Cut3 cut3(float x) {
  Cut3 c;
  unsigned xb; memcpy(&xb, &x, 4);
  c.h = xb & 0xffff0000u;
  float hf; memcpy(&hf, &c.h, 4);
  const float r1 = x - hf;
  unsigned rb; memcpy(&rb, &r1, 4);
  c.m = rb & 0xffff0000u;
  float mf; memcpy(&mf, &c.m, 4);
  const float r2 = r1 - mf;
  memcpy(&c.l, &r2, 4);
  return c;
}
__device__ inline unsigned pack_hi(unsigned even, unsigned odd) { return __builtin_amdgcn_perm(odd, even, 0x07060302u); }
// (the product's softplus_f - csrc/common.h - adds a series branch for tiny exp(-|x|); ~8 v_* here)
__device__ inline float softplus_f(float x) {
  const float t = __builtin_amdgcn_exp2f(-1.44269504088896341f * fabsf(x));
  return fmaxf(x, 0.f) + 0.693147180559945309f * __builtin_amdgcn_logf(1.f + t);
}

template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}
constexpr int X9_PA[9] = {2, 1, 2, 0, 2, 1, 0, 1, 0}, X9_PB[9] = {2, 2, 1, 2, 0, 1, 1, 0, 0};
__device__ __forceinline__ void gl4(f32x4& dst, const float* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory"); }
__device__ __forceinline__ void gl4u(u32x4& dst, const u32x4* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory"); }

constexpr int X6_PA[6] = {0, 1, 0, 2, 1, 0}, X6_PB[6] = {2, 1, 1, 0, 0, 0};      // (A plane, B plane): h l, m m, h m, l h, m h, h h
// which fragment buffer a product reads, and which (plane, buffer) is requested after MFMA n of a half (see the header): l -> 0, m -> 1, h -> 0;
// the next half's l -> 1, m -> 0, h -> 1
__global__ __launch_bounds__(256, 2) void linear_x6_proto(const float* __restrict__ X, const unsigned char* __restrict__ Wc, const float* __restrict__ bias,
                                                           float* __restrict__ Y, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];      // 2 x SLAB_BYTES
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const float* xr = X + (size_t)(tile * 128 + wave * 32 + l31) * K + 8 * hh;
    f32x16 acc[NCB];
    static_for<NCB>([&](auto i) { acc[decltype(i)::value] = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}; });
    u32x4 wn[6];
    f32x4 xv[2];
    u32x4 Ac[3];
    auto cut_all = [&]() {
      static_for<4>([&](auto qq) {
        constexpr int q = decltype(qq)::value;
        const Cut3 e = cut3(xv[q >> 1][2 * (q & 1)]), o = cut3(xv[q >> 1][2 * (q & 1) + 1]);
        Ac[0][q] = pack_hi(e.h, o.h); Ac[1][q] = pack_hi(e.m, o.m); Ac[2][q] = pack_hi(e.l, o.l);
      });
    };
    __syncthreads();                 // (the previous tile's last reads of LDS buffer 0)
    {
      const u32x4* src = reinterpret_cast<const u32x4*>(Wc) + tid;
#pragma unroll
      for (int q = 0; q < 6; ++q) reinterpret_cast<u32x4*>(lds)[q * 256 + tid] = src[q * 256];
    }
    xv[0] = *reinterpret_cast<const f32x4*>(xr); xv[1] = *reinterpret_cast<const f32x4*>(xr + 4);
    cut_all();
    __syncthreads();
#pragma unroll 1
    for (int ks = 0; ks < KS; ++ks) {
      const unsigned char* cur = lds + (ks & 1) * SLAB_BYTES;
      unsigned char* oth = lds + ((ks + 1) & 1) * SLAB_BYTES;
      const int kn = ks + 1 < KS ? ks + 1 : ks;                       // (last k step: re-touches itself, results unused)
      const u32x4* wsrc = reinterpret_cast<const u32x4*>(Wc + (size_t)kn * SLAB_BYTES) + tid;
#ifndef V_NOSLAB
      static_for<6>([&](auto q) { gl4u(wn[decltype(q)::value], wsrc + decltype(q)::value * 256); });
#endif
      gl4(xv[0], xr + 16 * kn); gl4(xv[1], xr + 16 * kn + 4);
      bf16x8 A[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) A[p] = __builtin_bit_cast(bf16x8, Ac[p]);
      bf16x8 B[2][4];
      auto read_b = [&](auto pp, auto hf, auto bf) {     // plane, half of the column blocks, buffer
        constexpr int p = decltype(pp)::value, half = decltype(hf)::value, b = decltype(bf)::value;
        static_for<4>([&](auto cc) {
          constexpr int c = decltype(cc)::value;
          B[b][c] = *reinterpret_cast<const bf16x8*>(cur + ((p * NCB + 4 * half + c) * 64 + lane) * 16);
        });
      };
      using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
      read_b(I2{}, I0{}, I0{}); read_b(I1{}, I0{}, I1{});
      __builtin_amdgcn_sched_barrier(0);
      static_for<48>([&](auto nn) {
        constexpr int n = decltype(nn)::value, half = n / 24, m = n % 24, s = m / 4, c = m % 4;
        constexpr int pb = X6_PB[s];                                   // plane 2 (l) -> buffer half, 1 (m) -> 1 - half, 0 (h) -> half
        constexpr int buf = pb == 1 ? 1 - half : half;
        acc[4 * half + c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[X6_PA[s]], B[buf][c], acc[4 * half + c], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (n == 3) { read_b(I0{}, I0{}, I0{}); __builtin_amdgcn_sched_barrier(0); }        // h of half 0 -> buffer 0 (l done)
        if constexpr (n == 11) { read_b(I2{}, I1{}, I1{}); __builtin_amdgcn_sched_barrier(0); }       // l of half 1 -> buffer 1 (m done)
        if constexpr (n == 23) { read_b(I1{}, I1{}, I0{}); __builtin_amdgcn_sched_barrier(0); }       // m of half 1 -> buffer 0 (h done)
        if constexpr (n == 27) { read_b(I0{}, I1{}, I1{}); __builtin_amdgcn_sched_barrier(0); }       // h of half 1 -> buffer 1 (l done)
      });
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(xv[j]));
#pragma unroll
      for (int j = 0; j < 6; ++j) asm volatile("" : "+v"(wn[j]));
#ifndef V_NOSLAB
#pragma unroll
      for (int q = 0; q < 6; ++q) reinterpret_cast<u32x4*>(oth)[q * 256 + tid] = wn[q];
#endif
      cut_all();
#ifndef V_NOBAR
      __syncthreads();
#endif
    }
    // epilogue: lane = column l31 of block cb, register r = row (r & 3) + 8 (r >> 2) + 4 hh of the wave's row block
    static_for<NCB>([&](auto i) {
      constexpr int cb = decltype(i)::value;
      const int row0 = tile * 128 + wave * 32;
      const float bc = bias[cb * 32 + l31];
      static_for<16>([&](auto rr) {
        constexpr int r = decltype(rr)::value;
#ifdef NO_EPI
        if (acc[cb][r] == 12345.678f) Y[(size_t)(row0 + (r & 3) + 8 * (r >> 2) + 4 * hh) * NOUT + cb * 32 + l31] = bc;
#else
        Y[(size_t)(row0 + (r & 3) + 8 * (r >> 2) + 4 * hh) * NOUT + cb * 32 + l31] = softplus_f(acc[cb][r] + bc);
#endif
      });
    });
  }
}

static unsigned short hi16(unsigned v) { return (unsigned short)(v >> 16); }

int main() {
  srand(7);
  const int M = 131072, Mchk = 256;
  std::vector<float> W(NOUT * K), b(NOUT), X((size_t)M * K);
  for (auto& v : W) v = ((float)rand() / RAND_MAX * 2 - 1) / 16;
  for (auto& v : b) v = (float)rand() / RAND_MAX - 0.5f;
  for (auto& v : X) v = ((float)rand() / RAND_MAX * 2 - 1) * 3;
  // cut weights in fragment order: [ks][plane][cb][lane] x 8 bf16 (lane: column cb * 32 + lane % 32, k = 16 ks + 8 (lane / 32) + j)
  std::vector<unsigned short> Wc((size_t)KS * 3 * NCB * 64 * 8);
  for (int ks = 0; ks < KS; ++ks)
    for (int cb = 0; cb < NCB; ++cb)
      for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 8; ++j) {
          const Cut3 c = cut3(W[(cb * 32 + lane % 32) * K + 16 * ks + 8 * (lane / 32) + j]);
          const unsigned pc[3] = {c.h, c.m, c.l};
          for (int p = 0; p < 3; ++p) Wc[((((size_t)ks * 3 + p) * NCB + cb) * 64 + lane) * 8 + j] = hi16(pc[p]);
        }
  float *dX, *db, *dY; unsigned char* dW;
  (void)hipMalloc(&dX, X.size() * 4); (void)hipMalloc(&db, NOUT * 4); (void)hipMalloc(&dY, (size_t)M * NOUT * 4); (void)hipMalloc(&dW, Wc.size() * 2);
  (void)hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(db, b.data(), NOUT * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(dW, Wc.data(), Wc.size() * 2, hipMemcpyHostToDevice);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(linear_x6_proto), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SLAB_BYTES);
  const int ntiles = M / 128;
  linear_x6_proto<<<512, 256, 2 * SLAB_BYTES>>>(dX, dW, db, dY, ntiles);
  (void)hipDeviceSynchronize();
  std::vector<float> Y((size_t)Mchk * NOUT);
  (void)hipMemcpy(Y.data(), dY, Y.size() * 4, hipMemcpyDeviceToHost);
  double emax = 0, sc = 0;
  for (int i = 0; i < Mchk; ++i)
    for (int o = 0; o < NOUT; ++o) {
      double pre = b[o];
      for (int k = 0; k < K; ++k) pre += (double)X[(size_t)i * K + k] * W[o * K + k];
      const double ref = pre > 30 ? pre : log1p(exp(pre));
      emax = fmax(emax, fabs(Y[(size_t)i * NOUT + o] - ref)); sc = fmax(sc, fabs(ref));
    }
  printf("x6 forward layer vs float64 (first %d rows): max error %.3e of the output scale %.3f\n", Mchk, emax / sc, sc);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int grid : {512, 1024}) {
    for (int r = 0; r < 300; ++r) linear_x6_proto<<<grid, 256, 2 * SLAB_BYTES>>>(dX, dW, db, dY, ntiles);      // clocks ramp over tens of ms
    (void)hipEventRecord(e0);
    for (int r = 0; r < 100; ++r) linear_x6_proto<<<grid, 256, 2 * SLAB_BYTES>>>(dX, dW, db, dY, ntiles);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 100;
    printf("M = %d, grid %d: %.1f us per layer = %.0f TFLOP/s fp32-equivalent, %.2f TB/s of algorithmic HBM traffic (FP32-MFMA kernel today: ~140 us)\n", M, grid, ms * 1e3,
           2.0 * M * K * NOUT / ms / 1e9, ((double)M * (K + NOUT) * 4) / ms / 1e9);
  }
  return 0;
}
