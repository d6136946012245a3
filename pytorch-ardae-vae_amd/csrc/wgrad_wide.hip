// Software-pipelined weight-gradient kernel for the big N-row problems (gfx950):  dW[o][i] = sum_pairs sum_m G[m][o] X[m][i]
// with O % 256 == 0, I % 256 == 0, M % 32 == 0.  Ragged / small problems stay on wgrad_kernel (wgrad.hip).
//
// One workgroup per CU, four waves = one per SIMD; the workgroup owns a 256 (o) x 256 (i) tile over a contiguous slice
// of the (pair, row) reduction range, each wave a 128 x 128 block = 4 x 4 v_mfma_f32_32x32x2_f32 accumulators (256
// accumulator registers: the wave has the SIMD's whole register file).  Compared with the 128 x 256 tile of wgrad_kernel
// both operands are read from HBM exactly once per tile and a k-step of 16 MFMAs needs only 8 LDS fragment reads.
//
// The schedule follows what the microbenchmarks under scratch/mfma/ established for this chip (see linear_wide_kernel.h):
// the wave runs ONE continuous stream of MFMAs; fragment reads run one k-step ahead, the next 32-row chunk travels
// HBM -> registers during k-steps 0-1, registers -> the other LDS buffer during k-steps 12-13, one s_barrier per chunk;
// every memory instruction sits behind an MFMA, issued as inline asm because hipcc sinks prefetches to their use.
#include <stdlib.h>

#include "profile.h"
#include "wgrad.h"

namespace ardae {
namespace {

constexpr int WT = 256;                      // tile edge (o and i)
constexpr int WRC = 32;                      // rows per chunk
constexpr int WCHUNK_BYTES = WRC * WT * 4;   // one operand chunk in LDS (32 KiB)
constexpr int WBUF_BYTES = 2 * WCHUNK_BYTES; // G chunk + X chunk
constexpr int WW_MAX_TILES = 32;

typedef __attribute__((address_space(3))) float lds_f32;

struct WwTile {
  const float* G[2];
  const float* X[2];
  const float* rowscale;   // sigma per row (pair `bias_pair` only) or null
  float* partial;          // [splits][O][I]
  float* partial_vec;      // [splits][2][O] or null
  int ldG[2], ldX[2];
  int M, npairs, O, I, o0, i0, bias_pair, want_vec;
};

struct WwBatchDev {
  int ntiles, splits;
  WwTile t[WW_MAX_TILES];
};
static_assert(sizeof(WwBatchDev) <= 4000, "kernel argument block too large");

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

// the memory instruction(s) behind MFMA number S of k-step KS
template <int KS, int S>
__device__ __forceinline__ void slot(float (&A)[2][4], float (&B)[2][4], f32x4 (&gv)[8], f32x4 (&xv)[8], float (&rs)[8], f32x4& bsum,
                                     f32x4& rsum, const WwCtx& c) {
  constexpr int nxt = (KS + 1) & 1;
  if constexpr (S < 8) {   // fragment reads of the next k-step (k-step 0 of the next chunk after the last one)
    constexpr int f = S & 3;
    if constexpr (KS < 15) {
      if constexpr (S < 4) lds_read1<(KS + 1) * 2048 + f * 128>(A[nxt][f], c.gfrag);
      else lds_read1<(KS + 1) * 2048 + f * 128>(B[nxt][f], c.xfrag);
    } else {
      if constexpr (S < 4) lds_read1<f * 128>(A[nxt][f], c.gfrag_n);
      else lds_read1<f * 128>(B[nxt][f], c.xfrag_n);
    }
  } else if constexpr (KS == 0) {   // next chunk: G rows 4 p + (tid >> 6), p = 0..7
    constexpr int p = S - 8;
    gload4<0>(gv[p], c.gvoff, c.gnext + (size_t)(4 * p) * c.ldg);
  } else if constexpr (KS == 1) {
    constexpr int p = S - 8;
    gload4<0>(xv[p], c.xvoff, c.xnext + (size_t)(4 * p) * c.ldx);
  } else if constexpr (KS == 2) {
    constexpr int p = S - 8;
    gload1<16 * p>(rs[p], c.rsvoff, c.rsnext);
  } else if constexpr (KS == 11) {
    // the chunk has landed (>= 9 k-steps, ~4 us): column sums of G for the bias / sigma gradients (plain VALU, priced in
    // matrix time: 2 instructions per MFMA slot)
    constexpr int p = S - 8;
    if constexpr (p == 0) {
      asm volatile("s_waitcnt vmcnt(0)"
                   : "+v"(gv[0]), "+v"(gv[1]), "+v"(gv[2]), "+v"(gv[3]), "+v"(gv[4]), "+v"(gv[5]), "+v"(gv[6]), "+v"(gv[7])
                   :
                   : "memory");
      asm volatile("s_waitcnt vmcnt(0)"
                   : "+v"(xv[0]), "+v"(xv[1]), "+v"(xv[2]), "+v"(xv[3]), "+v"(xv[4]), "+v"(xv[5]), "+v"(xv[6]), "+v"(xv[7])
                   :
                   : "memory");
      asm volatile("s_waitcnt vmcnt(0)"
                   : "+v"(rs[0]), "+v"(rs[1]), "+v"(rs[2]), "+v"(rs[3]), "+v"(rs[4]), "+v"(rs[5]), "+v"(rs[6]), "+v"(rs[7])
                   :
                   : "memory");
    }
    bsum += gv[p] * c.fb;
    rsum += gv[p] * (rs[p] * c.fr);
  } else if constexpr (KS == 12) {
    constexpr int p = S - 8;
    lds_write4<p * 4096>(c.gw, gv[p]);
  } else if constexpr (KS == 13) {
    constexpr int p = S - 8;
    lds_write4<p * 4096>(c.xw, xv[p]);
  }
}

template <int KS, int S = 0>
__device__ __forceinline__ void kstep_mfmas(f32x16 (&acc)[4][4], float (&A)[2][4], float (&B)[2][4], f32x4 (&gv)[8], f32x4 (&xv)[8],
                                            float (&rs)[8], f32x4& bsum, f32x4& rsum, const WwCtx& c) {
  if constexpr (S < 16) {
    constexpr int a = S >> 2, b = S & 3, cur = KS & 1;
    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[cur][a], B[cur][b], acc[a][b], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    slot<KS, S>(A, B, gv, xv, rs, bsum, rsum, c);
    __builtin_amdgcn_sched_barrier(0);
    kstep_mfmas<KS, S + 1>(acc, A, B, gv, xv, rs, bsum, rsum, c);
  }
}

template <int KS = 0>
__device__ __forceinline__ void chunk_ksteps(f32x16 (&acc)[4][4], float (&A)[2][4], float (&B)[2][4], f32x4 (&gv)[8], f32x4 (&xv)[8],
                                             float (&rs)[8], f32x4& bsum, f32x4& rsum, const WwCtx& c) {
  if constexpr (KS < 16) {
    constexpr int cur = KS & 1;
    // fragments of this k-step were read one k-step ago; lgkmcnt(0) also covers the staging writes of k-steps 12-13
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(A[cur][0]), "+v"(A[cur][1]), "+v"(A[cur][2]), "+v"(A[cur][3]), "+v"(B[cur][0]), "+v"(B[cur][1]), "+v"(B[cur][2]),
                   "+v"(B[cur][3])
                 :
                 : "memory");
    if constexpr (KS == 15) __builtin_amdgcn_s_barrier();   // every wave wrote its share of the next chunk (k-steps 12-13)
    __builtin_amdgcn_sched_barrier(0);
    kstep_mfmas<KS>(acc, A, B, gv, xv, rs, bsum, rsum, c);
    chunk_ksteps<KS + 1>(acc, A, B, gv, xv, rs, bsum, rsum, c);
  }
}

__global__ __launch_bounds__(256, 1) void wgrad_wide_kernel(const WwBatchDev batch) {
  __shared__ float lds[2 * WBUF_BYTES / 4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  const int wo = wave >> 1, wi = wave & 1;
  const int ti = (int)blockIdx.x / batch.splits, split = (int)blockIdx.x - ti * batch.splits;
  const WwTile& T = batch.t[ti];

  // this workgroup's slice of the concatenated (pair, row) range, in chunks of 32 rows
  const int cpp = T.M / WRC;                       // chunks per pair
  const int ctot = cpp * T.npairs;
  const int cps = (ctot + batch.splits - 1) / batch.splits;
  const int c_begin = split * cps;
  const int c_end = c_begin + cps < ctot ? c_begin + cps : ctot;

  const unsigned lds0 = (unsigned)(uintptr_t)(lds_f32*)lds;
  const int sr = tid >> 6, sc4 = tid & 63;         // staging: row within a pass of 4, float4 column
  const unsigned wbase = lds0 + (unsigned)((sr * WT + sc4 * 4) * 4);
  const unsigned gfrag0 = lds0 + (unsigned)((hh * WT + wo * 128 + l31) * 4);
  const unsigned xfrag0 = lds0 + WCHUNK_BYTES + (unsigned)((hh * WT + wi * 128 + l31) * 4);

  auto chunk_ptrs = [&](int ch, WwCtx& c) {   // wave-uniform
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
    c.xvoff = (unsigned)((sr * c.ldx + sc4 * 4) * 4);
    c.rsvoff = (unsigned)(sr * 4);
  };

  f32x16 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f}, rsum = {0.f, 0.f, 0.f, 0.f};

  float A[2][4], B[2][4], rs[8];
  f32x4 gv[8], xv[8];
  WwCtx c;
  if (c_begin < c_end) {
    // ---- prologue: first chunk -> LDS buffer 0, fragments of k-step 0 in flight
    chunk_ptrs(c_begin, c);
#pragma unroll
    for (int p = 0; p < 8; ++p) gload4<0>(gv[p], c.gvoff, c.gnext + (size_t)(4 * p) * c.ldg);
#pragma unroll
    for (int p = 0; p < 8; ++p) gload4<0>(xv[p], c.xvoff, c.xnext + (size_t)(4 * p) * c.ldx);
    gload1<0>(rs[0], c.rsvoff, c.rsnext); gload1<16>(rs[1], c.rsvoff, c.rsnext); gload1<32>(rs[2], c.rsvoff, c.rsnext);
    gload1<48>(rs[3], c.rsvoff, c.rsnext); gload1<64>(rs[4], c.rsvoff, c.rsnext); gload1<80>(rs[5], c.rsvoff, c.rsnext);
    gload1<96>(rs[6], c.rsvoff, c.rsnext); gload1<112>(rs[7], c.rsvoff, c.rsnext);
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(gv[0]), "+v"(gv[1]), "+v"(gv[2]), "+v"(gv[3]), "+v"(gv[4]), "+v"(gv[5]), "+v"(gv[6]), "+v"(gv[7])
                 :
                 : "memory");
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(xv[0]), "+v"(xv[1]), "+v"(xv[2]), "+v"(xv[3]), "+v"(xv[4]), "+v"(xv[5]), "+v"(xv[6]), "+v"(xv[7])
                 :
                 : "memory");
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(rs[0]), "+v"(rs[1]), "+v"(rs[2]), "+v"(rs[3]), "+v"(rs[4]), "+v"(rs[5]), "+v"(rs[6]), "+v"(rs[7])
                 :
                 : "memory");
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      bsum += gv[p] * c.fb;
      rsum += gv[p] * (rs[p] * c.fr);
    }
    lds_write4<0 * 4096>(wbase, gv[0]); lds_write4<1 * 4096>(wbase, gv[1]); lds_write4<2 * 4096>(wbase, gv[2]); lds_write4<3 * 4096>(wbase, gv[3]);
    lds_write4<4 * 4096>(wbase, gv[4]); lds_write4<5 * 4096>(wbase, gv[5]); lds_write4<6 * 4096>(wbase, gv[6]); lds_write4<7 * 4096>(wbase, gv[7]);
    const unsigned xw0 = wbase + WCHUNK_BYTES;
    lds_write4<0 * 4096>(xw0, xv[0]); lds_write4<1 * 4096>(xw0, xv[1]); lds_write4<2 * 4096>(xw0, xv[2]); lds_write4<3 * 4096>(xw0, xv[3]);
    lds_write4<4 * 4096>(xw0, xv[4]); lds_write4<5 * 4096>(xw0, xv[5]); lds_write4<6 * 4096>(xw0, xv[6]); lds_write4<7 * 4096>(xw0, xv[7]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    lds_read1<0>(A[0][0], gfrag0); lds_read1<128>(A[0][1], gfrag0); lds_read1<256>(A[0][2], gfrag0); lds_read1<384>(A[0][3], gfrag0);
    lds_read1<0>(B[0][0], xfrag0); lds_read1<128>(B[0][1], xfrag0); lds_read1<256>(B[0][2], xfrag0); lds_read1<384>(B[0][3], xfrag0);

    int buf = 0;
    for (int ch = c_begin; ch < c_end; ++ch) {
      chunk_ptrs(ch + 1, c);
      const unsigned cur = (unsigned)buf * WBUF_BYTES, oth = (unsigned)(buf ^ 1) * WBUF_BYTES;
      c.gfrag = gfrag0 + cur; c.xfrag = xfrag0 + cur;
      c.gfrag_n = gfrag0 + oth; c.xfrag_n = xfrag0 + oth;
      c.gw = wbase + oth; c.xw = wbase + oth + WCHUNK_BYTES;
      chunk_ksteps<0>(acc, A, B, gv, xv, rs, bsum, rsum, c);
      buf ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                 : "+v"(A[0][0]), "+v"(A[0][1]), "+v"(A[0][2]), "+v"(A[0][3]), "+v"(B[0][0]), "+v"(B[0][1]), "+v"(B[0][2]), "+v"(B[0][3])
                 :
                 : "memory");
  }

  // ---- partial tile store: partial[split][o][i] (zeros when the slice was empty: the reduction sums every split)
  float* __restrict__ part = T.partial + (size_t)split * T.O * T.I;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int i = T.i0 + wi * 128 + b * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = T.o0 + wo * 128 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
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

bool wgrad_wide_eligible(const WgradProblem& p) {
  static const bool off = getenv("ARDAE_WGRAD_WIDE") && atoi(getenv("ARDAE_WGRAD_WIDE")) == 0;
  if (off) return false;
  if (p.O % WT || p.I % WT || p.M % WRC || p.M < 64 * WRC) return false;
  for (int k = 0; k < p.npairs; ++k) {
    if ((p.ldG[k] & 3) || (p.ldX[k] & 3) || !al16g(p.G[k]) || !al16g(p.X[k])) return false;
    if ((int64_t)p.ldG[k] * 4 * 4 >= (int64_t)1 << 31 || (int64_t)p.ldX[k] * 4 * 4 >= (int64_t)1 << 31) return false;
  }
  return true;
}

// Runs the eligible problems (their .splits are lowered in place to the count actually used, so that the common reduction
// sums exactly the partials written here).  Returns the number of problems handled, or < 0 on error.
int launch_wgrad_wide(WgradProblem* probs, const int* idx, int n, hipStream_t st) {
  if (n == 0) return 0;
  WwBatchDev b;
  memset(&b, 0, sizeof(b));
  int ntiles = 0, min_splits = 1 << 30;
  for (int k = 0; k < n; ++k) {
    const WgradProblem& p = probs[idx[k]];
    ntiles += (p.O / WT) * (p.I / WT);
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
    for (int o0 = 0; o0 < p.O; o0 += WT)
      for (int i0 = 0; i0 < p.I; i0 += WT) {
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
  if (g_prof_enabled) prof_begin(st, "wgrad_wide_kernel", fl, by);
  hipLaunchKernelGGL(wgrad_wide_kernel, dim3(ntiles * splits), dim3(256), 0, st, b);
  prof_end(st);
  ARDAE_LAUNCH_CHECK();
  return n;
}

}  // namespace ardae
