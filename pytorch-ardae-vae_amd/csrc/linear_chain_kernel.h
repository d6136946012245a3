// Row-local layer CHAINS of the N-row passes in ONE launch (gfx950): the small-shard regime of linear_wide_kernel.
//
// The cDAE update walks its N rows through runs of h -> h layers that are row-local: forward (A_2 .. A_L | W_1 .. W_L), the score
// pass (DACT x (2L - 1)), the forward-mode pass (CHAIN x (2L - 1)) and the ordinary backward (DACT + Q x (2L - 1)) - SURVEY
// appendix A, models/graddae/mlp.py:400-444.  linear_wide_kernel runs each layer as one launch with the wave's weight slab
// resident in its AGPRs; with many tiles per workgroup that is the right shape (a launch boundary is < 5 % of a layer).  On the
// 8-rank shard of config #2 (16384 rows = ONE 64-row tile per workgroup) it is not: a launch is 16 us of K loop inside 24-26 us
// (kernel start, first panel from HBM, slab from L2, store drain, launch boundary), 22 times per step.
//
// Here a workgroup keeps its row tile and walks the layers of a chain:
//   * the tile's activations never leave the CU between layers: the epilogue's result (the next layer's input) goes from the
//     accumulator layout straight into the A-operand image in LDS (two tile sets of four 64 x 64 panels, 136 KiB: layer l reads
//     set l & 1 and writes set (l + 1) & 1, one s_barrier per layer); it is still stored to HBM - the weight gradients need every
//     layer's tensors - but those stores ride behind the MFMAs of the NEXT layer's K loop, as the operand loads (S, R / Q) do;
//   * the weight slab of layer l + 1 is reloaded into the SAME AGPRs chunk by chunk, one chunk behind the MFMAs that read layer
//     l's fragment (a load issued after an MFMA has issued cannot overtake its operand read), so a layer switch costs no
//     exposed L2 latency except for the last chunk's fragment;
//   * no panel loads, no per-panel barriers and no vmcnt waits inside the K loop.
// Per layer that leaves K loop + epilogue arithmetic + the LDS hand-over; the fixed ~9 us of a launch is paid once per chain.
//
// Layers of a chain share the epilogue kind; for EPI_ACT the FIRST layer may carry the per-image row bias + sigma rank-1 term
// (W_1 of the energy network, F2) and the LAST one the score seed output (F1) - exactly the reference's neglogprob stack.
// Shapes: K = Nout = 256 (h_dim 256: BASELINE configs #1-#3 and the shipped mlp / aux recipes), M % 64 == 0, row-bias groups
// that are multiples of 64 rows.  Everything else stays on the per-layer kernels.
#pragma once
#include "linear_wide_kernel.h"

namespace ardae {
namespace wide {

constexpr int CH_MAXL = 6;
constexpr int CH_NCH = 8, CH_NP = 4, CH_NJ = 2, CH_G = CH_NCH * CH_NP;   // 32 chunks of 8 k: K = 256
constexpr int CH_SET_BYTES = CH_NP * WBUF_BYTES;                         // one A-tile image: four 64 x 64 panels

struct ChainArgs {
  int nl, M;
  LinArgs L[CH_MAXL];   // L[0].src[0].x = the chain's input rows; L[l > 0]'s input is layer l - 1's Y (taken from LDS)
};

// reload of a slab fragment IN PLACE (tied operand: the fragment keeps its AGPRs for the whole kernel)
__device__ __forceinline__ void gload4_agpr_again(f32x4& dst, unsigned voff, const float* sbase) {
  asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "+a"(dst) : "v"(voff), "s"(sbase) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_write1_at(unsigned addr, float v) {
  asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
template <class T>
struct TypeTag {
  using type = T;
};

// what the previous layer left in the result slots and where it goes
struct PrevOut {
  i32x4 rY, rY2;
  unsigned vY, vY2, sY, sY2;
};

template <class EPI_T>
struct ChainCtx {
  const EPI_T& epi;
  const PrevOut& prev;
  unsigned raddr;           // fragment base of the tile set this layer reads
  unsigned bvoff;           // lane * 16
  const float* wnext;       // next layer's slab of this wave (HN) / the first layer's again (NT)
  int kch;
  int row0, colw;
  // NT (last layer of a tile, another tile follows): the next tile's input panels travel HBM -> registers -> the free tile set
  const i32x4& rX;
  unsigned xvoff, xstep, xsoff_next, wfree;
};

// One chunk of a chain layer's K loop: 16 MFMAs with, spread behind them,
//   2 fragment reads A(g + 1) | 2 slab reloads of chunk g - 1 (next layer) | deferred stores of the PREVIOUS layer
//   | 2 row-bias loads | epilogue-operand loads of THIS layer
// PNST: tensors the previous layer stores per element (0: first layer of a chain); HN: there is a next layer.
template <int GC, bool HN, bool NT, int PNST, class EPI_T>
struct ChainChunk {
  static constexpr int NLT = EPI_T::NLT, NJ = CH_NJ, NMF = 8 * CH_NJ, NX = PanelGeo<CH_NCH>::NX;
  using SC = Sched<CH_NCH, CH_NP, CH_NJ, NLT, (PNST > 0 ? PNST : 1), (PNST > 0)>;
  static constexpr int C = GC % CH_NCH, P = GC / CH_NCH;
  static constexpr bool RS = HN || NT;                         // the slab is reloaded behind the MFMAs (next layer's / the first layer's)
  static constexpr int XW = CH_NCH - 3;                        // chunk of a panel that moves the prefetched panel to LDS
  static constexpr int n_a = GC + 1 < CH_G ? 2 : 0, n_sl = (RS && GC >= 1) ? 2 : 0, n_x = (NT && C == 0) ? NX : 0, n_w = (NT && C == XW) ? NX : 0,
                       n_st = SC::st_hi(GC) - SC::st_lo(GC), n_rb = SC::n_rb(GC), n_op = SC::ld_hi(GC) - SC::ld_lo(GC);
  static constexpr int o_a = 0, o_sl = o_a + n_a, o_x = o_sl + n_sl, o_w = o_x + n_x, o_st = o_w + n_w, o_rb = o_st + n_st, o_op = o_rb + n_rb,
                       total = o_op + n_op;
  static constexpr int PER = (total + NMF - 1) / NMF;
  // vector-memory operations of chunk g besides the panel loads; those younger than the panel loads of chunk (P, 0) when chunk (P, XW)
  // moves them to LDS (vmcnt retires in order, so capping at 63 only waits longer)
  static constexpr int vmem_rest(int g) { return (SC::st_hi(g) - SC::st_lo(g)) + SC::n_rb(g) + (SC::ld_hi(g) - SC::ld_lo(g)); }
  static constexpr int vm_panel() {
    int n = vmem_rest(P * CH_NCH);
    for (int k = 1; k < XW; ++k) n += vmem_rest(P * CH_NCH + k) + 2;
    n += 2;
    return n > 63 ? 63 : n;
  }

  using Ctx = ChainCtx<EPI_T>;
  struct Ptrs {
    unsigned py, py2, p0, p1;
  };

  template <int K>
  static __device__ __forceinline__ void op(f32x4 (&A)[2][2], f32x4 (&Bw)[CH_G][CH_NJ], f32x4 (&xv)[NX], f32x2* l0, f32x2* l1, float (&rb)[CH_NJ], const Ctx& x,
                                            Ptrs& q) {
    const EPI_T& ep = x.epi;
    if constexpr (K < o_sl) {   // fragments of the next chunk (panels are contiguous: one base, compile-time offsets)
      constexpr int i = K - o_a, G1 = GC + 1;
      lds_read4<(G1 / CH_NCH) * WBUF_BYTES + (G1 % CH_NCH) * 32 + i * 32 * WLDW * 4>(A[G1 & 1][i], x.raddr);
    } else if constexpr (K < o_x) {
      constexpr int j = K - o_sl;
      gload4_agpr_again(Bw[GC - 1][j], x.bvoff, x.wnext + ((size_t)j * x.kch + (GC - 1)) * 256);
    } else if constexpr (K < o_w) {   // panel P of the NEXT tile's input
      constexpr int u = K - o_x;
      bload4<0>(xv[u], x.xvoff, x.rX, x.xsoff_next + (unsigned)(P * 256) + (unsigned)u * x.xstep);
    } else if constexpr (K < o_st) {
      constexpr int u = K - o_w;
      if constexpr (u == 0) wait_panel<vm_panel(), NX>(xv);
      lds_write4<P * WBUF_BYTES + u * PanelGeo<CH_NCH>::RPP * WLDW * 4>(x.wfree, xv[u]);
    } else if constexpr (K < o_rb) {
      constexpr int idx = SC::st_lo(GC) + (K - o_st), PN = PNST > 0 ? PNST : 1;
      constexpr int HB = idx / (8 * PN), tns = (idx / 8) % PN, e = idx % 8;
      constexpr int J = HB >> 2, I = (HB >> 1) & 1, H = HB & 1;
      constexpr bool fresh = e == 0 || K == o_st;
      constexpr int roff = 32 * I + 16 * H + (e & 3) + 8 * (e >> 2);
      if constexpr (tns == 0) {
        if constexpr (fresh) q.py = (unsigned)(x.row0 + roff) * x.prev.sY + (unsigned)(x.colw + 32 * J) * 4u;
        bstore1(x.prev.vY, l0[4 * HB + (e >> 1)][e & 1], x.prev.rY, q.py);
        q.py += (e == 3) ? 5u * x.prev.sY : x.prev.sY;
      } else {
        if constexpr (fresh) q.py2 = (unsigned)(x.row0 + roff) * x.prev.sY2 + (unsigned)(x.colw + 32 * J) * 4u;
        bstore1(x.prev.vY2, l1[4 * HB + (e >> 1)][e & 1], x.prev.rY2, q.py2);
        q.py2 += (e == 3) ? 5u * x.prev.sY2 : x.prev.sY2;
      }
    } else if constexpr (K < o_op) {
      constexpr int k = K - o_rb;
      if constexpr (k == 0) ep.template issue_rowbias_one<0>(rb[0], x.row0, x.colw);
      else ep.template issue_rowbias_one<1>(rb[NJ - 1], x.row0, x.colw);
    } else {
      constexpr int idx = SC::ld_lo(GC) + (K - o_op), NL_ = NLT > 0 ? NLT : 1;
      constexpr int HB = idx / (8 * NL_), tns = (idx / 8) % NL_, e = idx % 8;
      constexpr int J = HB >> 2, I = (HB >> 1) & 1, H = HB & 1;
      constexpr bool fresh = e == 0 || K == o_op;
      constexpr int roff = 32 * I + 16 * H + (e & 3) + 8 * (e >> 2);
      if constexpr (EPI_T::SIGMA_OPERAND) {
        constexpr int off = ((e & 3) + 8 * (e >> 2)) * 4;
        bload1<off, (e & 1)>(l0[4 * HB + (e >> 1)], ep.vRS, ep.rRS, (unsigned)(x.row0 + 32 * I + 16 * H) * 4u);
      } else if constexpr (tns == 0) {
        if constexpr (fresh) q.p0 = (unsigned)(x.row0 + roff) * ep.sL0 + (unsigned)(x.colw + 32 * J) * 4u;
        bload1<0, (e & 1)>(l0[4 * HB + (e >> 1)], ep.vL0, ep.rL0, q.p0);
        q.p0 += (e == 3) ? 5u * ep.sL0 : ep.sL0;
      } else {
        if constexpr (fresh) q.p1 = (unsigned)(x.row0 + roff) * ep.sL1 + (unsigned)(x.colw + 32 * J) * 4u;
        bload1<0, (e & 1)>(l1[4 * HB + (e >> 1)], ep.vL1, ep.rL1, q.p1);
        q.p1 += (e == 3) ? 5u * ep.sL1 : ep.sL1;
      }
    }
  }

  template <int S, int R = 0>
  static __device__ __forceinline__ void slot(f32x4 (&A)[2][2], f32x4 (&Bw)[CH_G][CH_NJ], f32x4 (&xv)[NX], f32x2* l0, f32x2* l1, float (&rb)[CH_NJ], const Ctx& x,
                                              Ptrs& q) {
    if constexpr (R < PER && S * PER + R < total) {
      op<S * PER + R>(A, Bw, xv, l0, l1, rb, x, q);
      slot<S, R + 1>(A, Bw, xv, l0, l1, rb, x, q);
    }
  }

  template <int S>
  static __device__ __forceinline__ void steps(f32x16 (&acc)[2][CH_NJ], f32x4 (&A)[2][2], f32x4 (&Bw)[CH_G][CH_NJ], f32x4 (&xv)[NX], f32x2* l0, f32x2* l1,
                                               float (&rb)[CH_NJ], const Ctx& x, Ptrs& q) {
    if constexpr (S < NMF) {
      constexpr int kq = S / (2 * NJ), i = (S / NJ) & 1, j = S % NJ;
      mfma_vab<GC == 0 && kq == 0, GC == CH_G - 1 && S == NMF - 1>(acc[i][j], A[GC & 1][i][kq], Bw[GC][j][kq]);
      slot<S>(A, Bw, xv, l0, l1, rb, x, q);
      steps<S + 1>(acc, A, Bw, xv, l0, l1, rb, x, q);
    }
  }

  static __device__ __forceinline__ void run(f32x16 (&acc)[2][CH_NJ], f32x4 (&A)[2][2], f32x4 (&Bw)[CH_G][CH_NJ], f32x4 (&xv)[NX], f32x2* l0, f32x2* l1,
                                             float (&rb)[CH_NJ], const Ctx& x) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(A[GC & 1][0]), "+v"(A[GC & 1][1]) : : "memory");   // this chunk's fragments have landed
    Ptrs q{0u, 0u, 0u, 0u};
    steps<0>(acc, A, Bw, xv, l0, l1, rb, x, q);
  }
};

template <int GC, bool HN, bool NT, int PNST, class EPI_T>
__device__ __forceinline__ void chain_chunks(f32x16 (&acc)[2][CH_NJ], f32x4 (&A)[2][2], f32x4 (&Bw)[CH_G][CH_NJ], f32x4 (&xv)[PanelGeo<CH_NCH>::NX], f32x2* l0,
                                             f32x2* l1, float (&rb)[CH_NJ], const ChainCtx<EPI_T>& x) {
  if constexpr (GC < CH_G) {
    ChainChunk<GC, HN, NT, PNST, EPI_T>::run(acc, A, Bw, xv, l0, l1, rb, x);
    chain_chunks<GC + 1, HN, NT, PNST, EPI_T>(acc, A, Bw, xv, l0, l1, rb, x);
  }
}

// result slots (accumulator layout) -> the A-operand image of the next layer: wave w's 64 columns are panel w of the tile set;
// element e of half-block HB = (J, I, H) is row 32 I + 16 H + (e & 3) + 8 (e >> 2) (+ 4 hh in the lane part), column 32 J (+ l31)
template <int IDX, int NJ>
__device__ __forceinline__ void hand_over(const f32x2* l0, unsigned waddr) {
  if constexpr (IDX < 32 * NJ) {
    constexpr int HB = IDX / 8, e = IDX % 8;
    constexpr int J = HB >> 2, I = (HB >> 1) & 1, H = HB & 1;
    lds_write1_at<((32 * I + 16 * H + (e & 3) + 8 * (e >> 2)) * WLDW + 32 * J) * 4>(waddr, l0[4 * HB + (e >> 1)][e & 1]);
    hand_over<IDX + 1, NJ>(l0, waddr);
  }
}

// EPI / ACT: the chain's epilogue; F2F: the first layer carries row bias + sigma (EPI_ACT); F1L: the last layer writes the score
// seed Y2 (EPI_ACT); FQ: EPI_DACT with the additive Q (all layers)
template <int EPI, int ACT, bool F2F, bool F1L, bool FQ>
__global__ __launch_bounds__(256, 1) void linear_chain_kernel(const ChainArgs ca, int nrt) {
  using EPI_FIRST = WideEpi<EPI, ACT, (EPI == EPI_DACT ? FQ : false), (EPI == EPI_ACT ? F2F : false), CH_NJ>;
  using EPI_MID = WideEpi<EPI, ACT, (EPI == EPI_DACT ? FQ : false), false, CH_NJ>;
  using EPI_LAST = WideEpi<EPI, ACT, (EPI == EPI_DACT ? FQ : (EPI == EPI_ACT ? F1L : false)), false, CH_NJ>;
  constexpr int NST_MID = EPI_MID::NST;    // what a non-last layer leaves for the next layer's K loop to store
  static_assert(EPI_FIRST::NST == EPI_MID::NST, "non-last layers store the same number of tensors");
  using PG = PanelGeo<CH_NCH>;
  constexpr int NX = PG::NX;
  extern __shared__ float lds[];   // 2 sets x 4 panels x 64 x 68 floats

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_f32*)lds;
  const unsigned rlane = lds0 + (unsigned)((l31 * WLDW + hh * 4) * 4);                          // fragment reads
  const unsigned wlane = lds0 + (unsigned)(((tid / PG::C4) * WLDW + (tid % PG::C4) * 4) * 4);    // input-panel stores
  const unsigned hlane = lds0 + (unsigned)wave * WBUF_BYTES + (unsigned)((4 * hh * WLDW + l31) * 4);   // hand-over writes (panel = wave)
  const int colw = wave * 64;
  const unsigned bvoff = (unsigned)lane * 16u;
  const int nl = ca.nl;
  constexpr int kch = 32;                                                                      // K / 8

  f32x4 A[2][2], Bw[CH_G][CH_NJ];
  f32x2 l0[16 * CH_NJ], l1[16 * CH_NJ];
  float rb[CH_NJ];
#pragma unroll
  for (int i = 0; i < 16 * CH_NJ; ++i) l0[i] = l1[i] = f32x2{0.f, 0.f};

  const LinArgs& a0 = ca.L[0];
  const unsigned ldx4 = (unsigned)a0.src[0].ld * 4u;
  const i32x4 rX = make_rsrc(a0.src[0].x, (unsigned)a0.M * ldx4);
  const unsigned xvoff = (unsigned)(tid / PG::C4) * ldx4 + (unsigned)(tid % PG::C4) * 16u;
  const unsigned xstep = (unsigned)PG::RPP * ldx4;
  // PF: the last layer of a tile prefetches the NEXT tile's input and the first layer's slab behind its MFMAs (16 more VGPRs: only the
  // epilogues whose register budget has them - the others pay an exposed input load per tile)
  constexpr bool PF = EPI == EPI_ACT || (EPI == EPI_DACT && !FQ);
  int tpar = 0;                 // layer l of this tile reads tile set (l + tpar) & 1
  // the chain's input tile (four panels) into tile set tpar and the first layer's slab into the AGPRs, latency exposed: once per
  // launch with PF (every later tile is staged by its predecessor's last layer), once per tile without.  No branches around it:
  // values defined by the asm loads must not meet other definitions at a control-flow join (see linear_wide_kernel)
  auto stage = [&](int row0) {
    f32x4 xv[CH_NP][NX];
#pragma unroll
    for (int p = 0; p < CH_NP; ++p)
#pragma unroll
      for (int u = 0; u < NX; ++u) bload4<0>(xv[p][u], xvoff, rX, (unsigned)row0 * ldx4 + (unsigned)p * 256u + (unsigned)u * xstep);
    const float* wp = a0.src[0].wp + (size_t)(wave * CH_NJ) * kch * 256;
#pragma unroll
    for (int g = 0; g < CH_G; ++g) {
      gload4_agpr_again(Bw[g][0], bvoff, wp + (size_t)g * 256);
      gload4_agpr_again(Bw[g][1], bvoff, wp + ((size_t)kch + g) * 256);
    }
    // the panels are older than the slab: landed when at most the slab loads are outstanding (vmcnt tops out at 63)
#pragma unroll
    for (int p = 0; p < CH_NP; ++p) {
      wait_panel<63, NX>(xv[p]);
      store_panel<CH_NCH, NX>(xv[p], wlane + (unsigned)(tpar & 1) * CH_SET_BYTES + (unsigned)p * WBUF_BYTES);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  if constexpr (PF) stage((int)blockIdx.x * WBM);
  for (int tile = (int)blockIdx.x; tile < nrt; tile += (int)gridDim.x) {
    const int row0 = tile * WBM;
    const bool more = tile + (int)gridDim.x < nrt;
    if constexpr (!PF) stage(row0);
    PrevOut prev{};
    // one layer: K loop (+ the previous layer's stores, this layer's operands, the next layer's slab), epilogue, hand-over
    auto layer = [&](auto epi_tag, auto hn_tag, auto nt_tag, auto pnst_tag, int l) {
      using EPI_T = typename decltype(epi_tag)::type;
      constexpr bool HN = decltype(hn_tag)::value, NT = decltype(nt_tag)::value;
      constexpr int PNST = decltype(pnst_tag)::value;
      const LinArgs& a = ca.L[l];
      EPI_T epi(a, lane);
      epi.column_operands(colw, lane);
      const unsigned rset = rlane + (unsigned)((l + tpar) & 1) * CH_SET_BYTES;
      lds_read4<0>(A[0][0], rset);
      lds_read4<32 * WLDW * 4>(A[0][1], rset);
      const float* wnext = (HN ? ca.L[l + 1].src[0].wp : a0.src[0].wp) + (size_t)(wave * CH_NJ) * kch * 256;
      f32x16 acc[2][CH_NJ];
      f32x4 xv[NX];
      // (no next tile: the staging loads re-touch this tile's rows and are never used)
      const ChainCtx<EPI_T> x{epi, prev, rset, bvoff, wnext, kch, row0, colw, rX, xvoff, xstep, (unsigned)(more ? row0 + (int)gridDim.x * WBM : row0) * ldx4,
                              wlane + (unsigned)((l + 1 + tpar) & 1) * CH_SET_BYTES};
      chain_chunks<0, HN, NT, PNST, EPI_T>(acc, A, Bw, xv, l0, l1, rb, x);
      if constexpr (HN || NT) {   // the last chunk's fragment of the next slab
        gload4_agpr_again(Bw[CH_G - 1][0], bvoff, wnext + (size_t)(CH_G - 1) * 256);
        gload4_agpr_again(Bw[CH_G - 1][1], bvoff, wnext + ((size_t)kch + CH_G - 1) * 256);
      }
      // nothing may be in flight across the compiler-scheduled epilogue (see linear_wide_kernel)
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                   : "+v"(A[0][0]), "+v"(A[0][1]), "+v"(acc[0][0]), "+v"(acc[0][CH_NJ - 1]), "+v"(acc[1][0]), "+v"(acc[1][CH_NJ - 1])
                   :
                   : "memory");
      epi.template run<0, !HN>(acc, l0, l1, rb, lane, row0, colw, tile);
      if constexpr (HN) {
        // the result is the next layer's input: accumulator layout -> A image of the other tile set (panel = this wave)
        hand_over<0, CH_NJ>(l0, hlane + (unsigned)((l + 1 + tpar) & 1) * CH_SET_BYTES);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        prev.rY = epi.rY; prev.rY2 = epi.rY2; prev.vY = epi.vY; prev.vY2 = epi.vY2; prev.sY = epi.sY; prev.sY2 = epi.sY2;
      }
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    layer(TypeTag<EPI_FIRST>{}, T_{}, F_{}, std::integral_constant<int, 0>{}, 0);
    for (int l = 1; l + 1 < nl; ++l) layer(TypeTag<EPI_MID>{}, T_{}, F_{}, std::integral_constant<int, NST_MID>{}, l);
    layer(TypeTag<EPI_LAST>{}, F_{}, std::integral_constant<bool, PF>{}, std::integral_constant<int, NST_MID>{}, nl - 1);
    if constexpr (PF) tpar = (tpar + nl) & 1;   // the free set of the last layer is where the next tile starts
    if (more) {   // another tile: everybody is done with the tile sets (and the prefetched panels are in place) before they are used
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

int chain_grid(int ntiles);

template <int EPI, int ACT, bool F2F, bool F1L, bool FQ>
int launch_chain(const ChainArgs& ca, hipStream_t st) {
  const int nrt = ca.M / WBM;
  auto kern = linear_chain_kernel<EPI, ACT, F2F, F1L, FQ>;
  constexpr int lds_bytes = 2 * CH_SET_BYTES;
  static bool attr_set = false;      // 136 KiB of the CU's 160 KiB: above the default dynamic-LDS limit
  if (!attr_set) {
    ARDAE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    attr_set = true;
  }
  if (g_prof_enabled) {
    char name[96];
    snprintf(name, sizeof(name), "linear_chain_kernel<%d, %d, %d, %d, %d> x%d", EPI, ACT, (int)F2F, (int)F1L, (int)FQ, ca.nl);
    double tensors = 0;
    for (int l = 0; l < ca.nl; ++l) {
      const LinArgs& a = ca.L[l];
      tensors += 1.0 + (a.Y2 ? 1 : 0) + ((EPI == EPI_DACT || EPI == EPI_CHAIN) ? 1 : 0) + ((EPI == EPI_CHAIN) ? 1 : 0) + ((EPI == EPI_DACT && a.Q) ? 1 : 0);
    }
    // algorithmic bytes: the chain's input once, every layer's operands / results (the weight gradients need them), the weights
    prof_begin(st, name, ca.nl * 2.0 * ca.M * 256.0 * 256.0, 4.0 * ((double)ca.M * 256.0 * (1.0 + tensors) + ca.nl * 256.0 * 256.0));
  }
  hipLaunchKernelGGL(kern, dim3(chain_grid(nrt)), dim3(256), lds_bytes, st, ca, nrt);
  prof_end(st);
  ARDAE_LAUNCH_CHECK();
  return 0;
}

// explicit instantiations live in linear_chain_inst_*.hip
#define ARDAE_CHAIN_EXTERN(EPI, ACT, F2F, F1L, FQ) extern template int launch_chain<EPI, ACT, F2F, F1L, FQ>(const ChainArgs&, hipStream_t);
#define ARDAE_CHAIN_INSTANTIATE(EPI, ACT, F2F, F1L, FQ) template int launch_chain<EPI, ACT, F2F, F1L, FQ>(const ChainArgs&, hipStream_t);
#define ARDAE_CHAIN_FOR_ACT(X, ACT)     \
  X(EPI_ACT, ACT, false, false, false)  \
  X(EPI_ACT, ACT, true, false, false)   \
  X(EPI_ACT, ACT, true, true, false)
#define ARDAE_CHAIN_FOR_DACT(X, ACT)    \
  X(EPI_DACT, ACT, false, false, false) \
  X(EPI_DACT, ACT, false, false, true)
#ifndef ARDAE_CHAIN_INST_TU
ARDAE_CHAIN_FOR_ACT(ARDAE_CHAIN_EXTERN, ACT_SOFTPLUS)
ARDAE_CHAIN_FOR_ACT(ARDAE_CHAIN_EXTERN, ACT_RELU)
ARDAE_CHAIN_FOR_DACT(ARDAE_CHAIN_EXTERN, ACT_SOFTPLUS)
ARDAE_CHAIN_FOR_DACT(ARDAE_CHAIN_EXTERN, ACT_RELU)
ARDAE_CHAIN_EXTERN(EPI_CHAIN, ACT_SOFTPLUS, false, false, false)
#endif

}  // namespace wide
}  // namespace ardae
