// Software-pipelined weight-gradient kernel for the big N-row problems (gfx950):  dW[o][i] = sum_pairs sum_m G[m][o] X[m][i]
// with O % 256 == 0, I % 256 == 0, M % 32 == 0.  Ragged / small problems stay on wgrad_kernel (wgrad.hip).
//
// One workgroup per CU, four waves = one per SIMD; the workgroup owns a 256 (o) x 256 (i) tile over a contiguous slice
// of the (pair, row) reduction range, each wave a 128 x 128 block = 4 x 4 v_mfma_f32_32x32x2_f32 accumulators (256
// accumulator registers: the wave has the SIMD's whole register file).  Compared with the 128 x 256 tile of wgrad_kernel
// both operands are read from HBM exactly once per tile and a k-step of 16 MFMAs needs only 8 LDS fragment reads.
// A second geometry, 256 (o) x 32 (i) with each wave 64 x 32, serves the first-layer problems (I = z_dim = 32), which are
// a pure stream of G through HBM.
//
// The schedule follows what the microbenchmarks under scratch/mfma/ established for this chip (see linear_wide_kernel.h):
// the wave runs ONE continuous stream of MFMAs; fragment reads run one k-step ahead, the next 32-row chunk travels
// HBM -> registers during k-steps 0-1, registers -> the other LDS buffer during k-steps 12-13, one s_barrier per chunk;
// every memory instruction sits behind an MFMA, issued as inline asm because hipcc sinks prefetches to their use.
#include <stdlib.h>

#include "profile.h"
#include "wgrad.h"
#include "wgrad_wide_tiles.h"

namespace ardae {
namespace {

constexpr int WT = 256;                      // tile edge along o (and along i in the square geometry)
constexpr int WRC = 32;                      // rows per chunk
constexpr int WCHUNK_BYTES = WRC * WT * 4;   // the G chunk in LDS (32 KiB); the X chunk follows it

// NA x NB MFMA blocks per wave, WOW waves along o
template <int NA, int NB, int WOW>
struct WGeo {
  static constexpr int NA_ = NA, NB_ = NB;
  static constexpr int WIW = 4 / WOW;
  static constexpr int TO = NA * 32 * WOW, TI = NB * 32 * WIW;
  static constexpr int NM = NA * NB;                 // MFMAs per k-step
  static constexpr int XC4 = TI / 4;                 // float4 per X row
  static constexpr int XRPP = 256 / XC4;             // X rows per staging pass
  static constexpr int NXP = WRC / XRPP;             // X staging passes per chunk (float4 per thread)
  static constexpr int XCHUNK_BYTES = WRC * TI * 4;
  static constexpr int BUF_BYTES = WCHUNK_BYTES + XCHUNK_BYTES;
  static_assert(TO == WT, "the G staging layout assumes 256 output rows per tile");
};
using GeoSquare = WGeo<4, 4, 2>;   // 256 x 256
using GeoNarrow = WGeo<2, 1, 4>;   // 256 x 32

typedef __attribute__((address_space(3))) float lds_f32;



// s_nop 4: see linear_wide_kernel.h (VALU-written scalar base -> VMEM needs 5 wait states, invisible to the compiler here)
template <int OFF>
__device__ __forceinline__ void gload4(f32x4& dst, unsigned voff, const float* sbase) {
  asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void gload1(float& dst, unsigned voff, const float* sbase) {
  asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_read1(float& dst, unsigned addr) {
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_write4(unsigned addr, const f32x4& v) {
  asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}

struct WwCtx {
  const float* gnext;   // next chunk: G rows (row and column origin applied)
  const float* xnext;
  const float* rsnext;  // sigma of the next chunk's rows (or any readable address)
  int ldg, ldx;
  unsigned gvoff, xvoff, rsvoff;   // per-lane byte offsets of the staging loads
  unsigned gfrag, xfrag;           // fragment read bases (current buffer)
  unsigned gfrag_n, xfrag_n;       // ... of the other buffer
  unsigned gw, xw;                 // staging write bases (other buffer)
  float fb, fr;                    // 1.0 when the NEXT chunk contributes to the bias / sigma-weighted column sums
};

// What rides behind the MFMAs of k-step KS besides the NA + NB fragment reads of the next k-step: the j-th "stage"
// instruction.  kinds: 0 G load, 1 X load, 2 sigma load, 3 wait + column-sum arithmetic, 4 G LDS write, 5 X LDS write.
template <class GEO>
struct Stage {
  static constexpr bool SQ = GEO::NM >= 8;
  // square: one k-step per kind (8 instructions behind 16 MFMAs); narrow (2 MFMAs per k-step): 2 per k-step
  static constexpr int PER = SQ ? 8 : 2;
  static constexpr int k_gl = 0, n_gl = 8 / PER;                       // k-steps [k_gl, k_gl + n_gl)
  static constexpr int k_xl = k_gl + n_gl, n_xl = (GEO::NXP + PER - 1) / PER;
  static constexpr int k_rs = k_xl + n_xl, n_rs = 8 / PER;
  static constexpr int k_sum = 11;                                      // all 8 in one k-step (needs the wait first)
  static constexpr int k_gw = 12, n_gw = SQ ? 1 : 2;                    // narrow: 4 writes per k-step
  static constexpr int k_xw = k_gw + n_gw;
  static_assert(k_rs + n_rs <= k_sum - 2 && k_xw <= 14, "stage schedule does not fit the chunk");
  static constexpr int count(int ks) {
    if (ks >= k_gl && ks < k_gl + n_gl) return PER;
    if (ks >= k_xl && ks < k_xl + n_xl) return (GEO::NXP - (ks - k_xl) * PER) < PER ? (GEO::NXP - (ks - k_xl) * PER) : PER;
    if (ks >= k_rs && ks < k_rs + n_rs) return PER;
    if (ks == k_sum) return 8;
    if (ks >= k_gw && ks < k_gw + n_gw) return 8 / n_gw;
    if (ks == k_xw) return GEO::NXP;
    return 0;
  }
};

template <class GEO, int KS, int J>
__device__ __forceinline__ void stage_op(f32x4 (&gv)[8], f32x4 (&xv)[GEO::NXP], float (&rs)[8], f32x4& bsum, f32x4& rsum, const WwCtx& c) {
  using ST = Stage<GEO>;
  if constexpr (KS >= ST::k_gl && KS < ST::k_gl + ST::n_gl) {          // next chunk: G rows 4 p + (tid >> 6), p = 0..7
    constexpr int p = (KS - ST::k_gl) * ST::PER + J;
    gload4<0>(gv[p], c.gvoff, c.gnext + (size_t)(4 * p) * c.ldg);
  } else if constexpr (KS >= ST::k_xl && KS < ST::k_xl + ST::n_xl) {
    constexpr int p = (KS - ST::k_xl) * ST::PER + J;
    gload4<0>(xv[p], c.xvoff, c.xnext + (size_t)(GEO::XRPP * p) * c.ldx);
  } else if constexpr (KS >= ST::k_rs && KS < ST::k_rs + ST::n_rs) {
    constexpr int p = (KS - ST::k_rs) * ST::PER + J;
    gload1<16 * p>(rs[p], c.rsvoff, c.rsnext);
  } else if constexpr (KS == ST::k_sum) {
    // the chunk has landed (issued >= 2 us ago): column sums of G for the bias / sigma gradients (plain VALU, priced in
    // matrix time)
    if constexpr (J == 0) {
      asm volatile("s_waitcnt vmcnt(0)"
                   : "+v"(gv[0]), "+v"(gv[1]), "+v"(gv[2]), "+v"(gv[3]), "+v"(gv[4]), "+v"(gv[5]), "+v"(gv[6]), "+v"(gv[7])
                   :
                   : "memory");
      if constexpr (GEO::NXP == 8)
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(xv[0]), "+v"(xv[1]), "+v"(xv[2]), "+v"(xv[3]), "+v"(xv[4]), "+v"(xv[5]), "+v"(xv[6]), "+v"(xv[7])
                     :
                     : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" : "+v"(xv[0]) : : "memory");
      asm volatile("s_waitcnt vmcnt(0)"
                   : "+v"(rs[0]), "+v"(rs[1]), "+v"(rs[2]), "+v"(rs[3]), "+v"(rs[4]), "+v"(rs[5]), "+v"(rs[6]), "+v"(rs[7])
                   :
                   : "memory");
    }
    bsum += gv[J] * c.fb;
    rsum += gv[J] * (rs[J] * c.fr);
  } else if constexpr (KS >= ST::k_gw && KS < ST::k_gw + ST::n_gw) {
    constexpr int p = (KS - ST::k_gw) * (8 / ST::n_gw) + J;
    lds_write4<p * 4096>(c.gw, gv[p]);
  } else if constexpr (KS == ST::k_xw) {
    lds_write4<J * GEO::XRPP * GEO::TI * 4>(c.xw, xv[J]);
  }
}

// instruction number Q of k-step KS: first the NA + NB fragment reads of the next k-step (k-step 0 of the next chunk
// after the last one), then the stage instructions
template <class GEO, int KS, int Q>
__device__ __forceinline__ void kstep_op(float (&A)[2][GEO::NA_], float (&B)[2][GEO::NB_], f32x4 (&gv)[8], f32x4 (&xv)[GEO::NXP], float (&rs)[8],
                                         f32x4& bsum, f32x4& rsum, const WwCtx& c) {
  constexpr int NA = GEO::NA_, NB = GEO::NB_;
  constexpr int nxt = (KS + 1) & 1;
  if constexpr (Q < NA) {
    if constexpr (KS < 15) lds_read1<(KS + 1) * 2048 + Q * 128>(A[nxt][Q], c.gfrag);
    else lds_read1<Q * 128>(A[nxt][Q], c.gfrag_n);
  } else if constexpr (Q < NA + NB) {
    constexpr int f = Q - NA;
    if constexpr (KS < 15) lds_read1<(KS + 1) * 2 * GEO::TI * 4 + f * 128>(B[nxt][f], c.xfrag);
    else lds_read1<f * 128>(B[nxt][f], c.xfrag_n);
  } else {
    stage_op<GEO, KS, Q - NA - NB>(gv, xv, rs, bsum, rsum, c);
  }
}

template <class GEO, int KS, int Q, int QEND>
__device__ __forceinline__ void kstep_ops(float (&A)[2][GEO::NA_], float (&B)[2][GEO::NB_], f32x4 (&gv)[8], f32x4 (&xv)[GEO::NXP], float (&rs)[8],
                                          f32x4& bsum, f32x4& rsum, const WwCtx& c) {
  if constexpr (Q < QEND) {
    kstep_op<GEO, KS, Q>(A, B, gv, xv, rs, bsum, rsum, c);
    kstep_ops<GEO, KS, Q + 1, QEND>(A, B, gv, xv, rs, bsum, rsum, c);
  }
}

template <class GEO, int KS, int S = 0>
__device__ __forceinline__ void kstep_mfmas(f32x16 (&acc)[GEO::NA_][GEO::NB_], float (&A)[2][GEO::NA_], float (&B)[2][GEO::NB_], f32x4 (&gv)[8],
                                            f32x4 (&xv)[GEO::NXP], float (&rs)[8], f32x4& bsum, f32x4& rsum, const WwCtx& c) {
  if constexpr (S < GEO::NM) {
    constexpr int a = S / GEO::NB_, b = S % GEO::NB_, cur = KS & 1;
    constexpr int total = GEO::NA_ + GEO::NB_ + Stage<GEO>::count(KS);
    constexpr int per = (total + GEO::NM - 1) / GEO::NM;
    constexpr int q0 = S * per < total ? S * per : total, q1 = (S + 1) * per < total ? (S + 1) * per : total;
    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[cur][a], B[cur][b], acc[a][b], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    kstep_ops<GEO, KS, q0, q1>(A, B, gv, xv, rs, bsum, rsum, c);
    __builtin_amdgcn_sched_barrier(0);
    kstep_mfmas<GEO, KS, S + 1>(acc, A, B, gv, xv, rs, bsum, rsum, c);
  }
}

template <int N>
__device__ __forceinline__ void wait_frags(float (&v)[N]) {
  if constexpr (N == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) : : "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]) : : "memory");
  else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]) : : "memory");
}

template <class GEO, int KS = 0>
__device__ __forceinline__ void chunk_ksteps(f32x16 (&acc)[GEO::NA_][GEO::NB_], float (&A)[2][GEO::NA_], float (&B)[2][GEO::NB_], f32x4 (&gv)[8],
                                             f32x4 (&xv)[GEO::NXP], float (&rs)[8], f32x4& bsum, f32x4& rsum, const WwCtx& c) {
  if constexpr (KS < 16) {
    constexpr int cur = KS & 1;
    // fragments of this k-step were read one k-step ago; lgkmcnt(0) also covers the staging writes of k-steps 12-14
    wait_frags(A[cur]);
    wait_frags(B[cur]);
    if constexpr (KS == 15) __builtin_amdgcn_s_barrier();   // every wave wrote its share of the next chunk
    __builtin_amdgcn_sched_barrier(0);
    kstep_mfmas<GEO, KS>(acc, A, B, gv, xv, rs, bsum, rsum, c);
    chunk_ksteps<GEO, KS + 1>(acc, A, B, gv, xv, rs, bsum, rsum, c);
  }
}

template <class GEO>
__global__ __launch_bounds__(256, 1) void wgrad_wide_kernel(const WwBatchDev batch) {
  constexpr int NA = GEO::NA_, NB = GEO::NB_, NXP = GEO::NXP;
  __shared__ float lds[2 * GEO::BUF_BYTES / 4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  const int wo = wave / GEO::WIW, wi = wave % GEO::WIW;
  const int ti = (int)blockIdx.x / batch.splits, split = (int)blockIdx.x - ti * batch.splits;
  const WwTile& T = batch.t[ti];

  // this workgroup's slice of the concatenated (pair, row) range, in chunks of 32 rows
  const int cpp = T.M / WRC;                       // chunks per pair
  const int ctot = cpp * T.npairs;
  const int cps = (ctot + batch.splits - 1) / batch.splits;
  const int c_begin = split * cps;
  const int c_end = c_begin + cps < ctot ? c_begin + cps : ctot;

  const unsigned lds0 = (unsigned)(uintptr_t)(lds_f32*)lds;
  const int sr = tid >> 6, sc4 = tid & 63;         // G staging: row within a pass of 4, float4 column
  const int xr = tid / GEO::XC4, xc4 = tid % GEO::XC4;
  const unsigned gwbase = lds0 + (unsigned)((sr * WT + sc4 * 4) * 4);
  const unsigned xwbase = lds0 + WCHUNK_BYTES + (unsigned)((xr * GEO::TI + xc4 * 4) * 4);
  const unsigned gfrag0 = lds0 + (unsigned)((hh * WT + wo * NA * 32 + l31) * 4);
  const unsigned xfrag0 = lds0 + WCHUNK_BYTES + (unsigned)((hh * GEO::TI + wi * NB * 32 + l31) * 4);

  WwCtx c;
  auto chunk_ptrs = [&](int ch) {   // wave-uniform
    const int cc = ch < c_end ? ch : c_end - 1;   // past the end: re-touch the last chunk (never used)
    const int pr = cc >= cpp ? 1 : 0;
    const int m0 = (cc - pr * cpp) * WRC;
    c.ldg = pr ? T.ldG[1] : T.ldG[0];
    c.ldx = pr ? T.ldX[1] : T.ldX[0];
    c.gnext = (pr ? T.G[1] : T.G[0]) + (size_t)m0 * c.ldg + T.o0;
    c.xnext = (pr ? T.X[1] : T.X[0]) + (size_t)m0 * c.ldx + T.i0;
    const bool vec = T.want_vec && pr == T.bias_pair && ch < c_end;
    c.fb = vec ? 1.f : 0.f;
    c.fr = (vec && T.rowscale) ? 1.f : 0.f;
    c.rsnext = T.rowscale ? T.rowscale + m0 : c.gnext;
    c.gvoff = (unsigned)((sr * c.ldg + sc4 * 4) * 4);
    c.xvoff = (unsigned)((xr * c.ldx + xc4 * 4) * 4);
    c.rsvoff = (unsigned)(sr * 4);
  };

  f32x16 acc[NA][NB];
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f}, rsum = {0.f, 0.f, 0.f, 0.f};

  float A[2][NA], B[2][NB], rs[8];
  f32x4 gv[8], xv[NXP];
  if (c_begin < c_end) {
    // ---- prologue: first chunk -> LDS buffer 0, fragments of k-step 0 in flight (compiler-scheduled: nothing else is
    //      in flight yet, so plain loads are fine here)
    chunk_ptrs(c_begin);
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      gv[p] = *reinterpret_cast<const f32x4*>(c.gnext + (size_t)(4 * p + sr) * c.ldg + sc4 * 4);
      rs[p] = T.rowscale ? T.rowscale[(c.rsnext - T.rowscale) + 4 * p + sr] : 0.f;
    }
#pragma unroll
    for (int p = 0; p < NXP; ++p) xv[p] = *reinterpret_cast<const f32x4*>(c.xnext + (size_t)(GEO::XRPP * p + xr) * c.ldx + xc4 * 4);
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      bsum += gv[p] * c.fb;
      rsum += gv[p] * (rs[p] * c.fr);
      *reinterpret_cast<f32x4*>(&lds[((4 * p + sr) * WT + sc4 * 4)]) = gv[p];
    }
#pragma unroll
    for (int p = 0; p < NXP; ++p) *reinterpret_cast<f32x4*>(&lds[WCHUNK_BYTES / 4 + (GEO::XRPP * p + xr) * GEO::TI + xc4 * 4]) = xv[p];
    __syncthreads();
#pragma unroll
    for (int f = 0; f < NA; ++f) A[0][f] = lds[hh * WT + wo * NA * 32 + f * 32 + l31];
#pragma unroll
    for (int f = 0; f < NB; ++f) B[0][f] = lds[WCHUNK_BYTES / 4 + hh * GEO::TI + wi * NB * 32 + f * 32 + l31];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

    int buf = 0;
    for (int ch = c_begin; ch < c_end; ++ch) {
      chunk_ptrs(ch + 1);
      const unsigned cur = (unsigned)buf * GEO::BUF_BYTES, oth = (unsigned)(buf ^ 1) * GEO::BUF_BYTES;
      c.gfrag = gfrag0 + cur; c.xfrag = xfrag0 + cur;
      c.gfrag_n = gfrag0 + oth; c.xfrag_n = xfrag0 + oth;
      c.gw = gwbase + oth; c.xw = xwbase + oth;
      chunk_ksteps<GEO, 0>(acc, A, B, gv, xv, rs, bsum, rsum, c);
      buf ^= 1;
    }
    wait_frags(A[0]);
    wait_frags(B[0]);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }

  // ---- partial tile store: partial[split][o][i] (zeros when the slice was empty: the reduction sums every split)
  float* __restrict__ part = T.partial + (size_t)split * T.O * T.I;
#pragma unroll
  for (int a = 0; a < NA; ++a) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int i = T.i0 + wi * NB * 32 + b * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = T.o0 + wo * NA * 32 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        part[(size_t)o * T.I + i] = acc[a][b][r];
      }
    }
  }
  // ---- column sums of G (bias gradient) and sigma-weighted column sums (the sigma column of W1): reduce the four row
  //      groups of the staging layout through LDS
  if (T.want_vec && T.i0 == 0) {
    __syncthreads();
    *reinterpret_cast<f32x4*>(&lds[sr * WT + sc4 * 4]) = bsum;
    *reinterpret_cast<f32x4*>(&lds[4 * WT + sr * WT + sc4 * 4]) = rsum;
    __syncthreads();
    const float s0 = (lds[tid] + lds[WT + tid]) + (lds[2 * WT + tid] + lds[3 * WT + tid]);
    const float s1 = (lds[4 * WT + tid] + lds[5 * WT + tid]) + (lds[6 * WT + tid] + lds[7 * WT + tid]);
    T.partial_vec[((size_t)split * 2 + 0) * T.O + T.o0 + tid] = s0;
    T.partial_vec[((size_t)split * 2 + 1) * T.O + T.o0 + tid] = s1;
  }
}

bool al16g(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// 0: not eligible, 1: square geometry (256 x 256 tiles), 2: narrow geometry (256 x 32 tiles)
int wgrad_wide_geometry(const WgradProblem& p) {
  static const bool off = debug_knob("ARDAE_WGRAD_WIDE") && atoi(debug_knob("ARDAE_WGRAD_WIDE")) == 0;
  if (off) return 0;
  if (p.O % WT || p.I % 32 || p.M % WRC || p.M < 64 * WRC) return 0;
  for (int k = 0; k < p.npairs; ++k) {
    if ((p.ldG[k] & 3) || (p.ldX[k] & 3) || !al16g(p.G[k]) || !al16g(p.X[k])) return 0;
    if ((int64_t)p.ldG[k] * 4 * 4 >= (int64_t)1 << 31 || (int64_t)p.ldX[k] * 32 * 4 >= (int64_t)1 << 31) return 0;
  }
  if (p.I % WT == 0) return 1;
  return p.I <= 64 ? 2 : 0;   // a few 32-column tiles; wider ragged problems are better off on wgrad_kernel
}

bool wgrad_wide_eligible(const WgradProblem& p) { return wgrad_wide_geometry(p) != 0; }

// Runs the problems idx[0..n) of one geometry (their .splits are lowered in place to the count actually used, so that
// the common reduction sums exactly the partials written here).
template <class GEO>
int launch_wgrad_wide_geo(WgradProblem* probs, const int* idx, int n, hipStream_t st, const char* name) {
  if (n == 0) return 0;
  WwBatchDev b;
  memset(&b, 0, sizeof(b));
  int ntiles = 0, min_splits = 1 << 30;
  for (int k = 0; k < n; ++k) {
    const WgradProblem& p = probs[idx[k]];
    ntiles += (p.O / GEO::TO) * (p.I / GEO::TI);
    if (p.splits < min_splits) min_splits = p.splits;
  }
  ARDAE_CHECK_ARG(ntiles <= WW_MAX_TILES, "wgrad_wide: too many tiles (%d)", ntiles);
  // one workgroup per CU: as many row splits as fill the 256 CUs once
  int splits = 256 / ntiles;
  if (splits < 1) splits = 1;
  if (splits > min_splits) splits = min_splits;
  b.ntiles = ntiles;
  b.splits = splits;
  int t = 0;
  double fl = 0, by = 0;
  for (int k = 0; k < n; ++k) {
    WgradProblem& p = probs[idx[k]];
    p.splits = splits;
    for (int o0 = 0; o0 < p.O; o0 += GEO::TO)
      for (int i0 = 0; i0 < p.I; i0 += GEO::TI) {
        WwTile& w = b.t[t++];
        for (int q = 0; q < 2; ++q) {
          w.G[q] = p.G[q]; w.X[q] = p.X[q]; w.ldG[q] = p.ldG[q]; w.ldX[q] = p.ldX[q];
        }
        w.rowscale = p.rowscale; w.partial = p.partial; w.partial_vec = p.partial_vec;
        w.M = p.M; w.npairs = p.npairs; w.O = p.O; w.I = p.I; w.o0 = o0; w.i0 = i0;
        w.bias_pair = p.bias_pair; w.want_vec = (p.bias_pair >= 0 && p.partial_vec != nullptr) ? 1 : 0;
      }
    fl += 2.0 * p.npairs * (double)p.M * p.O * p.I;
    by += 4.0 * (p.npairs * (double)p.M * (p.O + p.I) + (double)splits * p.O * p.I);
  }
  // 256 x 256 tiles: the products are formed on the BF16 matrix cores (wgrad_x9.hip); the FP32-MFMA kernel keeps the 256 x 32 geometry
  // (a pure stream of G through HBM) and the square problems x9 cannot take
  bool x9 = GEO::TI == 256;
  for (int k = 0; x9 && k < n; ++k) x9 = wgrad_x9_eligible(probs[idx[k]]);
  if (g_prof_enabled) prof_begin(st, x9 ? "wgrad_x9_kernel<256x256>" : name, fl, by);
  if (x9) {
    ARDAE_TRY(launch_wgrad_x9(b, st));
  } else {
    hipLaunchKernelGGL(wgrad_wide_kernel<GEO>, dim3(ntiles * splits), dim3(256), 0, st, b);
  }
  prof_end(st);
  ARDAE_LAUNCH_CHECK();
  return n;
}

int launch_wgrad_wide(WgradProblem* probs, const int* idx, int n, hipStream_t st) {
  int sq[WGRAD_MAX_PROBLEMS], nr[WGRAD_MAX_PROBLEMS], nsq = 0, nnr = 0;
  for (int k = 0; k < n; ++k) {
    if (wgrad_wide_geometry(probs[idx[k]]) == 1) sq[nsq++] = idx[k];
    else nr[nnr++] = idx[k];
  }
  int rc = launch_wgrad_wide_geo<GeoSquare>(probs, sq, nsq, st, "wgrad_wide_kernel<256x256>");
  if (rc < 0) return rc;
  rc = launch_wgrad_wide_geo<GeoNarrow>(probs, nr, nnr, st, "wgrad_wide_kernel<256x32>");
  return rc < 0 ? rc : n;
}

}  // namespace ardae
