// PROTOTYPE (round 4, planning data for DESIGN section 6 - not product code): one N-row forward layer of the cDAE update,
//     Y[M, 256] = softplus(X[M, 256] W^T + b),        M = 131072 at BASELINE config #2,
// with its fp32 products formed exactly on the BF16 matrix cores (three-piece cuts, nine piece products, fp32 accumulation: the arithmetic of
// csrc/wgrad_x9.hip).  Today's FP32-MFMA kernel (linear_wide_kernel, weight-stationary, asm-scheduled) takes ~140 us for this layer.
//   * the weights are cut ONCE (host side here; at pack time in the product) into three bf16 planes in MFMA-B fragment order
//     [k step (16)][plane (3)][column block (8)][lane (64)] x 16 bytes and streamed through LDS one k step (24 KiB) at a time, double buffered;
//   * a workgroup owns 256 rows x all 256 columns, wave w rows 64 w .. 64 w + 63 (256 accumulator registers, one wave per SIMD): lane
//     (row, k group) reads its 8 consecutive k of the row straight from the row-major X (one k step ahead) and cuts them once;
//   * per k step 144 MFMAs per wave as ONE stream, the next k step's loads, cuts, LDS writes and fragment reads in the gaps between them;
//   * epilogue: bias + softplus on the 256 accumulator values per lane, stores of 128-byte row segments.
// Measured on an MI355X (clocks warm: 300 untimed launches first - the first milliseconds of a burst run ~10 % faster):
//     build                                         bf16x9      bf16x6 (-DX6: without the products l l, m l, l m)
//     whole layer                                   162.4 us    142.4 us         (max error vs float64 4.1e-7 of the output scale for both)
//     -DNO_EPI (no bias / softplus / stores)        140.8       115.7
//     + -DV_NOGL -DV_NOBAR -DV_NOLDS (MFMAs only)   112.3        86.7
// i.e. the bare stream of 2 x 16 x 144 MFMAs per workgroup sustains 1.38 PFLOP/s on real operands (wgrad_x9_kernel: 1.61 with everything; the
// 2.03 of bf16x6.hip's rate test is for constant operands), and with one wave per SIMD nothing hides the tile prologue, the per-step
// hand-over (~28 us) or the epilogue (~22 us).  Conclusion for DESIGN section 6: nine products cannot beat the FP32-MFMA kernel's 140 us
// by a margin worth its rewrite; six products can (MFMA floor 87 us) if prologue and epilogue overlap the stream (two 128-row workgroups per CU).
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

// v2: 256-row tiles (a wave: 64 rows = 2 row blocks x all 8 column blocks, 256 accumulator registers: every B fragment feeds two MFMAs),
// one k step = 144 MFMAs as ONE stream with the next k step's loads, cuts, LDS writes and fragment reads in the gaps between them.
__global__ __launch_bounds__(256, 1) void linear_x9_proto(const float* __restrict__ X, const unsigned char* __restrict__ Wc, const float* __restrict__ bias,
                                                           float* __restrict__ Y, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];      // 2 x SLAB_BYTES
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const float* xr0 = X + (size_t)(tile * 256 + wave * 64 + l31) * K + 8 * hh;
    const float* xr1 = xr0 + (size_t)32 * K;
    f32x16 acc[2][NCB];
    static_for<2 * NCB>([&](auto i) { acc[decltype(i)::value / NCB][decltype(i)::value % NCB] = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}; });
    // k step 0: slab -> LDS buffer 0, A values in registers, cut
    {
      const u32x4* src = reinterpret_cast<const u32x4*>(Wc);
#pragma unroll
      for (int q = 0; q < 6; ++q) reinterpret_cast<u32x4*>(lds)[q * 256 + tid] = src[q * 256 + tid];
    }
    f32x4 xv[4] = {*reinterpret_cast<const f32x4*>(xr0), *reinterpret_cast<const f32x4*>(xr0 + 4), *reinterpret_cast<const f32x4*>(xr1),
                   *reinterpret_cast<const f32x4*>(xr1 + 4)};
    u32x4 An[2][3];      // the NEXT k step's operand slices (cut in the gaps of this one)
    // pair q (0 .. 3) of row block rb: values 2 q, 2 q + 1 of the lane's 8 -> dword q of the three planes (every index a compile-time constant)
    auto cut_pair = [&](auto rbq) {
      constexpr int rb = decltype(rbq)::value >> 2, q = decltype(rbq)::value & 3;
      const Cut3 e = cut3(xv[2 * rb + (q >> 1)][2 * (q & 1)]), o = cut3(xv[2 * rb + (q >> 1)][2 * (q & 1) + 1]);
      An[rb][0][q] = pack_hi(e.h, o.h); An[rb][1][q] = pack_hi(e.m, o.m); An[rb][2][q] = pack_hi(e.l, o.l);
    };
    static_for<8>([&](auto i) { cut_pair(i); });
    __syncthreads();
#pragma unroll 1
    for (int ks = 0; ks < KS; ++ks) {
      const unsigned char* cur = lds + (ks & 1) * SLAB_BYTES;
      unsigned char* oth = lds + ((ks + 1) & 1) * SLAB_BYTES;
      const int kn = ks + 1 < KS ? ks + 1 : ks;                       // (last k step: re-touches itself, results unused)
      const u32x4* wsrc = reinterpret_cast<const u32x4*>(Wc + (size_t)kn * SLAB_BYTES) + tid;
      bf16x8 A[2][3];
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int p = 0; p < 3; ++p) A[rb][p] = __builtin_bit_cast(bf16x8, An[rb][p]);
      u32x4 wn[6];
      // B fragments of FOUR column blocks at a time (48 registers): a plane's registers are re-loaded with the second half's fragments as
      // soon as the first half's last product using that plane has issued (plane l after product 3, m after product 6, h after product 8)
      bf16x8 B[3][4];
      auto read_plane = [&](auto ph) {
        constexpr int p = decltype(ph)::value >> 1, half = decltype(ph)::value & 1;
        static_for<4>([&](auto cc) {
          constexpr int c = decltype(cc)::value;
          B[p][c] = *reinterpret_cast<const bf16x8*>(cur + ((p * NCB + 4 * half + c) * 64 + lane) * 16);
        });
      };
      read_plane(std::integral_constant<int, 4>{}); read_plane(std::integral_constant<int, 2>{}); read_plane(std::integral_constant<int, 0>{});
      static_for<144>([&](auto nn) {
        constexpr int n = decltype(nn)::value, half = n / 72, m = n % 72, s = m / 8, c = (m % 8) / 2, rb = m % 2;
#ifdef X6
        if constexpr (s >= 3)       // bf16x6: without the three smallest piece products (l l, m l, l m)
#endif
        acc[rb][4 * half + c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[rb][X9_PA[s]], B[X9_PB[s]][c], acc[rb][4 * half + c], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#ifndef V_NOLDS
        if constexpr (n == 31) { read_plane(std::integral_constant<int, 5>{}); __builtin_amdgcn_sched_barrier(0); }
        if constexpr (n == 55) { read_plane(std::integral_constant<int, 3>{}); __builtin_amdgcn_sched_barrier(0); }
        if constexpr (n == 71) { read_plane(std::integral_constant<int, 1>{}); __builtin_amdgcn_sched_barrier(0); }
#endif
#ifndef V_NOGL
        if constexpr (n >= 4 && n < 24 && n % 2 == 0) {                // the next k step's 10 loads: 6 of the slab, 4 of X
          constexpr int q = (n - 4) / 2;
          if constexpr (q < 6) gl4u(wn[q], wsrc + q * 256);
          else gl4(xv[q - 6], (q - 6 < 2 ? xr0 : xr1) + 16 * kn + 4 * ((q - 6) & 1));
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (n >= 96 && n < 128 && n % 4 == 0) {               // cut the next k step's 16 values: one pair per gap
          constexpr int q = (n - 96) / 4;
          if constexpr (q == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(xv[j]));
#pragma unroll
            for (int j = 0; j < 6; ++j) asm volatile("" : "+v"(wn[j]));
          }
          cut_pair(std::integral_constant<int, q>{});
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (n >= 128 && n < 140 && n % 2 == 0) {              // the next slab into the other LDS buffer, one fragment per gap
          constexpr int q = (n - 128) / 2;
          reinterpret_cast<u32x4*>(oth)[q * 256 + tid] = wn[q];
          __builtin_amdgcn_sched_barrier(0);
        }
#endif
      });
#ifndef V_NOBAR
      __syncthreads();
#endif
    }
    // epilogue: lane = column l31 of block cb, register r = row (r & 3) + 8 (r >> 2) + 4 hh of the row block
    static_for<2 * NCB>([&](auto i) {
      constexpr int rb = decltype(i)::value / NCB, cb = decltype(i)::value % NCB;
      const int row0 = tile * 256 + wave * 64 + 32 * rb;
      const float bc = bias[cb * 32 + l31];
      static_for<16>([&](auto rr) {
        constexpr int r = decltype(rr)::value;
#ifdef NO_EPI
        if (acc[rb][cb][r] == 12345.678f) Y[(size_t)(row0 + (r & 3) + 8 * (r >> 2) + 4 * hh) * NOUT + cb * 32 + l31] = bc;
#else
        Y[(size_t)(row0 + (r & 3) + 8 * (r >> 2) + 4 * hh) * NOUT + cb * 32 + l31] = softplus_f(acc[rb][cb][r] + bc);
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
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(linear_x9_proto), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SLAB_BYTES);
  const int ntiles = M / 256;
  linear_x9_proto<<<256, 256, 2 * SLAB_BYTES>>>(dX, dW, db, dY, ntiles);
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
  printf("x9 forward layer vs float64 (first %d rows): max error %.3e of the output scale %.3f\n", Mchk, emax / sc, sc);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int grid : {256, 512}) {
    for (int r = 0; r < 300; ++r) linear_x9_proto<<<grid, 256, 2 * SLAB_BYTES>>>(dX, dW, db, dY, ntiles);      // clocks ramp over tens of ms
    (void)hipEventRecord(e0);
    for (int r = 0; r < 100; ++r) linear_x9_proto<<<grid, 256, 2 * SLAB_BYTES>>>(dX, dW, db, dY, ntiles);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 100;
    printf("M = %d, grid %d: %.1f us per layer = %.0f TFLOP/s fp32-equivalent, %.2f TB/s of algorithmic HBM traffic (FP32-MFMA kernel today: ~140 us)\n", M, grid, ms * 1e3,
           2.0 * M * K * NOUT / ms / 1e9, ((double)M * (K + NOUT) * 4) / ms / 1e9);
  }
  return 0;
}
