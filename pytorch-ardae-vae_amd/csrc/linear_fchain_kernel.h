// Fused chain of N-row layers (gfx950): consecutive 256 -> 256 layers of one MLP pass, each reading the previous one's
// output, run in ONE persistent launch.  A workgroup (one per CU, four waves, one per SIMD) owns 64 rows and walks the
// layers for them: the epilogue of layer l writes its result tile to HBM (deferred stores, as in linear_wide_kernel)
// AND into a 64 x 256 LDS tile that layer l+1 reads its activation fragments from - no activation re-read from HBM, no
// launch, cold start, store flush and inter-kernel gap per layer (≈30 us each).  Same K loop, same
// schedule (Sched / ChunkOps with XLDS = true) and same epilogue arithmetic as linear_wide_kernel, hence the same results.
//
// All layers of a chain share the epilogue kind; optional operands a layer does not have (row bias, sigma column, score
// seed, Y2) are neutralised at run time (WideEpi), so one instantiation serves the chain.
#pragma once
#include "linear_wide_kernel.h"

namespace ardae {
namespace wide {

constexpr int FC_MAX_LAYERS = 6;
constexpr int FC_TILE_BYTES = WBM * FC_TLD * 4;   // 66,560 B; two of them (in / out) fit the 160 KiB LDS

struct FcArgs {
  int nl, ntiles;
  LinArgs a[FC_MAX_LAYERS];
};
static_assert(sizeof(FcArgs) <= 4000, "kernel argument block too large");

// the four K panels of one layer for one tile, straight-line (see tile_panels in linear_wide_kernel.h)
template <int P, bool HP, class EPI_T>
__device__ __forceinline__ void fc_panels(f32x16 (&acc)[2][2], f32x4 (&A)[2][2], f32x4 (&B)[BDEPTH + 1][2], f32x4 (&xv)[PanelGeo<8>::NX], float* l0,
                                          float* l1, float (&rb)[2], WideState& s, const EPI_T& epi, unsigned rbase, unsigned bvoff, int tid,
                                          int row0, int colw, int prev_row0, const LinArgs* prev, unsigned pvY, unsigned pvY2,
                                          const float* const (&wnext)[2]) {
  if constexpr (P < 4) {
    // weight panel after this one: the next panel of the layer, or panel 0 of the next layer
    s.bnxt[0] = P == 3 ? wnext[0] : s.bcur[0] + 8 * 256;
    s.bnxt[1] = P == 3 ? wnext[1] : s.bcur[1] + 8 * 256;
    const PanelCtx<8, EPI_T> x{s, epi, rbase, rbase, rbase, bvoff, tid, row0, colw, prev_row0, prev, pvY, pvY2};
    panel<P, 8, 4, HP, EPI_T, true>(acc, A, B, xv, l0, l1, rb, x);
    s.bcur[0] = s.bnxt[0];
    s.bcur[1] = s.bnxt[1];
    fc_panels<P + 1, HP>(acc, A, B, xv, l0, l1, rb, s, epi, rbase, bvoff, tid, row0, colw, prev_row0, prev, pvY, pvY2, wnext);
  }
}

// Everything a wave carries from layer to layer.
struct FcCarry {
  int prev_row0;           // < 0: nothing to store yet
  const LinArgs* prev;     // layer whose results sit in l0 / l1
  unsigned pvY, pvY2;
  int buf;                 // LDS tile holding the current layer's input
};

// Layer LI of the chain for one tile.  The layer index is a compile-time constant: indexing the kernel-argument array
// with a run-time value would make hipcc copy it to scratch, and pointers loaded from there count as divergent (they
// could not feed the scalar-base memory instructions).
template <int LI, int NL, class EPI_T>
__device__ __forceinline__ void fc_layers(const FcArgs& c, f32x4 (&A)[2][2], f32x4 (&B)[BDEPTH + 1][2], f32x4 (&xv)[PanelGeo<8>::NX], float* l0,
                                          float* l1, float (&rb)[2], WideState& s, FcCarry& k, float* lds, unsigned rlane, unsigned bvoff,
                                          int tid, int lane, int wave, int row0, int colw, int tile, bool first_tile) {
  if constexpr (LI < NL) {
    const LinArgs& a = c.a[LI];
    const int l31 = lane & 31, hh = lane >> 5;
    EPI_T epi(a, lane);
    epi.column_operands(colw, lane);
    const unsigned rbase = rlane + (unsigned)k.buf * FC_TILE_BYTES;
    constexpr int LN = LI + 1 < NL ? LI + 1 : 0;
    const float* const wnext[2] = {c.a[LN].src[0].wp + (size_t)((wave * 2 + 0) * 32) * 256, c.a[LN].src[0].wp + (size_t)((wave * 2 + 1) * 32) * 256};
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    s.bcur[0] = a.src[0].wp + (size_t)((wave * 2 + 0) * 32) * 256;
    s.bcur[1] = a.src[0].wp + (size_t)((wave * 2 + 1) * 32) * 256;
    bool hp = true;
    if constexpr (LI == 0) {
      if (first_tile) {
        // very first layer of this workgroup: weight fragments of chunks 0..2 (later layers get them from the previous
        // layer's last panel) - and no results to store yet
        hp = false;
        s.bnxt[0] = s.bcur[0];
        s.bnxt[1] = s.bcur[1];
        issue_b<0, 8>(B, s, bvoff);
        issue_b<1, 8>(B, s, bvoff);
        issue_b<2, 8>(B, s, bvoff);
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(B[0][0]), "+v"(B[0][1]), "+v"(B[1][0]), "+v"(B[1][1]), "+v"(B[2][0]), "+v"(B[2][1])
                     :
                     : "memory");
      }
    }
    lds_read4<0>(A[0][0], rbase);
    lds_read4<32 * FC_TLD * 4>(A[0][1], rbase);
    // K loop + drain as one unit per code path: nothing may be in flight where the two paths of layer 0 join (the
    // compiler reconciles registers there), nor across the epilogue (see linear_wide_kernel)
    auto kloop = [&](auto hp_tag) {
      fc_panels<0, decltype(hp_tag)::value>(acc, A, B, xv, l0, l1, rb, s, epi, rbase, bvoff, tid, row0, colw, k.prev_row0, k.prev, k.pvY, k.pvY2,
                                            wnext);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                   : "+v"(B[0][0]), "+v"(B[0][1]), "+v"(B[1][0]), "+v"(B[1][1]), "+v"(B[2][0]), "+v"(B[2][1])
                   :
                   : "memory");
    };
    if constexpr (LI == 0) {
      if (!hp) kloop(std::false_type{});
      else kloop(std::true_type{});
    } else {
      kloop(std::true_type{});
    }
    epi.template run<0, !Sched<8, 4, EPI_T::NLT, EPI_T::NST, true, 0>::DEFER>(acc, l0, l1, rb, lane, row0, colw, tile);
    if constexpr (LI + 1 < NL) {   // the next layer's activation tile
      float* out = lds + (k.buf ^ 1) * (WBM * FC_TLD);
#pragma unroll
      for (int hb = 0; hb < 8; ++hb) {
        const int J = hb >> 2, I = (hb >> 1) & 1, H = hb & 1;
#pragma unroll
        for (int e = 0; e < 8; ++e) out[(32 * I + 16 * H + (e & 3) + 8 * (e >> 2) + 4 * hh) * FC_TLD + colw + 32 * J + l31] = l0[8 * hb + e];
      }
    }
    __syncthreads();   // out tile complete; every wave is done reading the in tile
    k.prev = &a;
    k.prev_row0 = row0;
    k.pvY = epi.vY;
    k.pvY2 = epi.vY2;
    k.buf ^= 1;
    fc_layers<LI + 1, NL, EPI_T>(c, A, B, xv, l0, l1, rb, s, k, lds, rlane, bvoff, tid, lane, wave, row0, colw, tile, first_tile);
  }
}

template <int NL, int EPI, int ACT, bool F1, bool F2>
__global__ __launch_bounds__(256, 1) void linear_fchain_kernel(const FcArgs c) {
  using EPI_T = WideEpi<EPI, ACT, F1, F2>;
  __shared__ float lds[2 * WBM * FC_TLD];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_f32*)lds;
  const unsigned bvoff = (unsigned)lane * 16u;
  const unsigned rlane = lds0 + (unsigned)((l31 * FC_TLD + hh * 4) * 4);   // fragment reads (buffer 0)
  const int colw = wave * 64;                                             // Nout == 256: one column panel

  f32x4 A[2][2], B[BDEPTH + 1][2], xv[PanelGeo<8>::NX];
  float l0[64], l1[64], rb[2];
  WideState s;
  s.xnxt = nullptr;
  s.ldnxt = 0;
  FcCarry k{-1, &c.a[0], 0u, 0u, 0};

  for (int tile = blockIdx.x; tile < c.ntiles; tile += gridDim.x) {
    const int row0 = tile * WBM;
    // ---- the input tile of layer 0: HBM -> LDS (plain loads: nothing else is in flight at a tile boundary)
    {
      const int ld = c.a[0].src[0].ld;
      const float* x = c.a[0].src[0].x + (size_t)row0 * ld;
      float* in = lds + k.buf * (WBM * FC_TLD);
      const int r = tid >> 6, c4 = tid & 63;
      // four rounds of four float4 per thread: the previous tile's results may still occupy 64 registers (deferred stores)
#pragma unroll 1
      for (int q = 0; q < 4; ++q) {
        f32x4 v[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) v[p] = *reinterpret_cast<const f32x4*>(x + (size_t)(16 * q + 4 * p + r) * ld + c4 * 4);
#pragma unroll
        for (int p = 0; p < 4; ++p) *reinterpret_cast<f32x4*>(in + (16 * q + 4 * p + r) * FC_TLD + c4 * 4) = v[p];
      }
      __syncthreads();
    }
    fc_layers<0, NL, EPI_T>(c, A, B, xv, l0, l1, rb, s, k, lds, rlane, bvoff, tid, lane, wave, row0, colw, tile, tile == (int)blockIdx.x);
  }
  // results of the last layer executed are still in registers (deferred stores)
  if (Sched<8, 4, EPI_T::NLT, EPI_T::NST, true, 0>::DEFER && k.prev_row0 >= 0) {
    EPI_T epi(c.a[NL - 1], lane);
    epi.store_all(l0, l1, k.prev_row0, colw);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

constexpr int FC_NL = 5;   // chain length instantiated: the 2L-1 = 5 h x h layers of a pass at L = 3

template <int EPI, int ACT, bool F1, bool F2>
int launch_fchain(const LinArgs* layers, int nl, hipStream_t st) {
  ARDAE_CHECK_ARG(nl == FC_NL, "fchain: %d layers (only chains of %d are instantiated)", nl, FC_NL);
  FcArgs c;
  memset(&c, 0, sizeof(c));
  c.nl = nl;
  c.ntiles = layers[0].M / WBM;
  double fl = 0, by = 0;
  for (int i = 0; i < nl; ++i) {
    c.a[i] = layers[i];
    fl += 2.0 * layers[i].M * 256.0 * 256.0;
    const double tensors = 1.0 + (layers[i].Y2 ? 1 : 0) + ((EPI == EPI_DACT || EPI == EPI_CHAIN) ? 1 : 0) + ((EPI == EPI_CHAIN) ? 1 : 0) +
                           ((EPI == EPI_DACT && layers[i].Q) ? 1 : 0);
    by += 4.0 * ((i == 0 ? (double)layers[i].M * 256.0 : 0.0) + tensors * layers[i].M * 256.0 + 256.0 * 256.0);
  }
  const int grid = wide_grid(c.ntiles, 1);
  if (g_prof_enabled) {
    char name[96];
    snprintf(name, sizeof(name), "linear_fchain_kernel<%d, %d, %d, %d> x%d", EPI, ACT, (int)F1, (int)F2, nl);
    prof_begin(st, name, fl, by);
  }
  hipLaunchKernelGGL((linear_fchain_kernel<FC_NL, EPI, ACT, F1, F2>), dim3(grid), dim3(256), 0, st, c);
  prof_end(st);
  ARDAE_LAUNCH_CHECK();
  return 0;
}

#define ARDAE_FCHAIN_FOR_ALL(X)            \
  X(EPI_ACT, ACT_SOFTPLUS, true, true)     \
  X(EPI_DACT, ACT_SOFTPLUS, false, false)  \
  X(EPI_DACT, ACT_SOFTPLUS, true, false)   \
  X(EPI_CHAIN, ACT_SOFTPLUS, false, false)
#define ARDAE_FCHAIN_EXTERN(EPI, ACT, F1, F2) extern template int launch_fchain<EPI, ACT, F1, F2>(const LinArgs*, int, hipStream_t);
#define ARDAE_FCHAIN_INSTANTIATE(EPI, ACT, F1, F2) template int launch_fchain<EPI, ACT, F1, F2>(const LinArgs*, int, hipStream_t);
#ifndef ARDAE_FCHAIN_INST_TU
ARDAE_FCHAIN_FOR_ALL(ARDAE_FCHAIN_EXTERN)
#endif

}  // namespace wide
}  // namespace ardae
