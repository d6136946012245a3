// Software-pipelined FP32-MFMA linear kernel for the big N-row layers (gfx950): 64 x 256 output tile per workgroup,
// 4 waves (one per SIMD), each wave a 64 x 64 block = 2 x 2 v_mfma_f32_32x32x2_f32 accumulators.
//
// Why a second kernel.  Measured on MI355X (scratch/mfma/ldasm.hip): a wave that issues its operand loads, waits for them
// and then runs its 16 MFMAs loses 20-40 % of the matrix pipe NO MATTER how many waves share the SIMD (2048 MFMAs: 155
// TFLOP/s with the loads one iteration ahead, 93-136 without) - the SIMD's round-robin MFMA arbitration phase-locks the
// co-resident waves, so they all reach their load phase together and nobody covers the latency.  Operand loads have to
// be in flight INSIDE each wave's own MFMA stream.  hipcc does not keep such a schedule (it sinks prefetch loads to
// their first use and reuses the registers), so the K loop below issues its loads and counter waits as inline asm:
//
//   * weight fragments: global_load_dwordx4 (saddr form, SALU-only addressing) from the packed L2-resident image,
//     BDEPTH = 3 chunks (48 MFMAs) ahead in a 4-slot register ring;
//   * activation fragments: ds_read_b128 with immediate offsets, one chunk ahead (2 register sets);
//   * next activation panel HBM -> registers: issued right after the first chunk's weight loads of a panel, so that by the
//     time an in-order vmcnt wait has to pass them (4 chunks later) they have landed; written to the other LDS buffer at
//     the end of the panel, one s_barrier per panel.
//
// Shapes: M % 64 == 0, Nout % 256 == 0, every source K % 32 == 0 (panels of 64 or 32), 16-byte aligned rows.  Anything
// else goes to linear_kernel (linear.hip), which handles ragged edges.
#include <stdlib.h>

#include <algorithm>

#include "linear.h"
#include "profile.h"

namespace ardae {
namespace {

constexpr int WBM = 64;                       // rows per tile
constexpr int WLDW = 68;                      // LDS row stride (floats): conflict-free ds_read_b128 fragments
constexpr int WBUF_BYTES = WBM * WLDW * 4;    // one K panel (<= 64 wide)
constexpr int BDEPTH = 3;                     // weight-fragment prefetch distance (chunks); ring of BDEPTH + 1 slots

typedef __attribute__((address_space(3))) float lds_f32;

template <int OFF>
__device__ __forceinline__ void gload4(f32x4& dst, unsigned voff, const float* sbase) {
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_read4(f32x4& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_write4(unsigned addr, const f32x4& v) {
  asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
// counter waits that "produce" the registers they guard, so no consumer can be scheduled above them
template <int VM>
__device__ __forceinline__ void wait_frag(f32x4& a0, f32x4& a1, f32x4& b0, f32x4& b1) {
  asm volatile("s_waitcnt vmcnt(%4) lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1) : "n"(VM) : "memory");
}
template <int VM, int NX>
__device__ __forceinline__ void wait_panel(f32x4 (&x)[NX]) {
  if (NX == 4) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]) : "n"(VM) : "memory");
  else asm volatile("s_waitcnt vmcnt(%2)" : "+v"(x[0]), "+v"(x[1]) : "n"(VM) : "memory");
}

// panel geometry for NCH chunks of 8 k: 2*NCH float4 per row, 256 / (2*NCH) rows per pass, NX passes
template <int NCH>
struct PanelGeo {
  static constexpr int C4 = 2 * NCH;
  static constexpr int RPP = 256 / C4;
  static constexpr int NX = WBM / RPP;
};

struct WideState {
  const float* bcur[2];   // packed-weight pointers of the two 32-column blocks at chunk 0 of the current panel
  const float* bnxt[2];   // ... of the next panel (== bcur when there is none: never dereferenced then)
  const float* xnxt;      // activation rows of the next panel (row0, k0 applied)
  int ldnxt;              // its leading dimension (floats)
};

template <int NCH, int NX>
__device__ __forceinline__ void issue_panel_loads(f32x4 (&xv)[NX], const float* xp, int ld, int tid) {
  using PG = PanelGeo<NCH>;
  const unsigned voff = (unsigned)(((tid / PG::C4) * ld + (tid % PG::C4) * 4) * 4);
  const size_t step = (size_t)PG::RPP * ld;
  gload4<0>(xv[0], voff, xp);
  gload4<0>(xv[1], voff, xp + step);
  if (NX == 4) {
    gload4<0>(xv[2], voff, xp + 2 * step);
    gload4<0>(xv[3], voff, xp + 3 * step);
  }
}

template <int NCH, int NX>
__device__ __forceinline__ void store_panel(const f32x4 (&xv)[NX], unsigned waddr) {
  using PG = PanelGeo<NCH>;
  lds_write4<0>(waddr, xv[0]);
  lds_write4<PG::RPP * WLDW * 4>(waddr, xv[1]);
  if (NX == 4) {
    lds_write4<2 * PG::RPP * WLDW * 4>(waddr, xv[2]);
    lds_write4<3 * PG::RPP * WLDW * 4>(waddr, xv[3]);
  }
}

template <int C, int NCH>
__device__ __forceinline__ void issue_b(f32x4 (&B)[BDEPTH + 1][2], const WideState& s, unsigned bvoff) {
  // chunk C of the current panel (C >= NCH: chunk C - NCH of the next one)
  constexpr int slot = C % (BDEPTH + 1);
  // chunk offset split into a 4-KiB step on the scalar base and an immediate (< 4096)
  constexpr int CC = C < NCH ? C : C - NCH;
  const float* const* base = C < NCH ? s.bcur : s.bnxt;
  gload4<(CC & 3) * 1024>(B[slot][0], bvoff, base[0] + (CC >> 2) * 1024);
  gload4<(CC & 3) * 1024>(B[slot][1], bvoff, base[1] + (CC >> 2) * 1024);
}

template <int C>
__device__ __forceinline__ void issue_a(f32x4 (&A)[2][2], unsigned raddr) {
  lds_read4<C * 32>(A[C & 1][0], raddr);
  lds_read4<C * 32 + 32 * WLDW * 4>(A[C & 1][1], raddr);
}

__device__ __forceinline__ void mfma16(f32x16 (&acc)[2][2], const f32x4 (&A)[2], const f32x4 (&B)[2]) {
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i][q], B[j][q], acc[i][j], 0, 0, 0);
}

// One chunk: wait for its fragments, put the next loads in flight, 16 MFMAs.
template <int C, int NCH, bool HAS_NEXT, int NX>
__device__ __forceinline__ void chunk(f32x16 (&acc)[2][2], f32x4 (&A)[2][2], f32x4 (&B)[BDEPTH + 1][2], f32x4 (&xv)[NX],
                                      const WideState& s, unsigned raddr, unsigned bvoff, int tid) {
  // loads younger than B(C) at this point: B(C+1) .. B(C+BDEPTH-1), plus the next panel's activation loads (issued in
  // chunk 0 after B(BDEPTH)) while B(C) is older than them, i.e. for 1 <= C <= BDEPTH
  constexpr int last_b = HAS_NEXT ? C + BDEPTH - 1 : (C + BDEPTH - 1 < NCH - 1 ? C + BDEPTH - 1 : NCH - 1);
  constexpr int vm = 2 * (last_b - C) + ((HAS_NEXT && C >= 1 && C <= BDEPTH) ? NX : 0);
  constexpr int slot = C % (BDEPTH + 1);
  wait_frag<vm>(A[C & 1][0], A[C & 1][1], B[slot][0], B[slot][1]);
  if (C + 1 < NCH) issue_a<C + 1>(A, raddr);
  if (HAS_NEXT || C + BDEPTH < NCH) issue_b<C + BDEPTH, NCH>(B, s, bvoff);
  if (HAS_NEXT && C == 0) issue_panel_loads<NCH, NX>(xv, s.xnxt, s.ldnxt, tid);
  __builtin_amdgcn_sched_barrier(0);
  mfma16(acc, A[C & 1], B[slot]);
  __builtin_amdgcn_sched_barrier(0);
}

template <int NCH, bool HAS_NEXT, int NX>
__device__ __forceinline__ void panel(f32x16 (&acc)[2][2], f32x4 (&A)[2][2], f32x4 (&B)[BDEPTH + 1][2], f32x4 (&xv)[NX],
                                      const WideState& s, unsigned raddr, unsigned bvoff, int tid) {
  chunk<0, NCH, HAS_NEXT, NX>(acc, A, B, xv, s, raddr, bvoff, tid);
  chunk<1, NCH, HAS_NEXT, NX>(acc, A, B, xv, s, raddr, bvoff, tid);
  chunk<2, NCH, HAS_NEXT, NX>(acc, A, B, xv, s, raddr, bvoff, tid);
  chunk<3, NCH, HAS_NEXT, NX>(acc, A, B, xv, s, raddr, bvoff, tid);
  if (NCH == 8) {
    chunk<4 % NCH, NCH, HAS_NEXT, NX>(acc, A, B, xv, s, raddr, bvoff, tid);
    chunk<5 % NCH, NCH, HAS_NEXT, NX>(acc, A, B, xv, s, raddr, bvoff, tid);
    chunk<6 % NCH, NCH, HAS_NEXT, NX>(acc, A, B, xv, s, raddr, bvoff, tid);
    chunk<7 % NCH, NCH, HAS_NEXT, NX>(acc, A, B, xv, s, raddr, bvoff, tid);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Epilogue.  The accumulator of a 32x32 block puts column l&31 and rows (r&3) + 8(r>>2) + 4(l>>5) in lane l, so operands
// and results move as dwords: one instruction = two full 128-byte lines.  Addressing is the saddr form (row base in
// SGPRs, per-lane column offset in one VGPR per tensor): no vector ALU work per access, which matters because VALU
// issue is paid for in matrix time on this chip.  Half-blocks of 8 registers are software-pipelined like the K loop:
// the operand loads of half-block h+1 are issued before the math and the stores of half-block h (outputs may alias
// inputs element-wise - Y == Q in place - and different half-blocks touch different elements).
// ---------------------------------------------------------------------------------------------------------------------
template <int OFF>
__device__ __forceinline__ void gload1(float& dst, unsigned voff, const float* sbase) {
  asm volatile("global_load_dword %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}
__device__ __forceinline__ void gstore1(unsigned voff, float v, float* sbase) {
  asm volatile("global_store_dword %0, %1, %2" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
}
template <int VM>
__device__ __forceinline__ void wait8(float (&v)[8]) {
  asm volatile("s_waitcnt vmcnt(%8)"
               : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])
               : "n"(VM)
               : "memory");
}

// row offset of register e (0..7) of half h inside a 32-row block, without the 4*(l>>5) lane part
__device__ __forceinline__ constexpr int erow(int h, int e) { return 16 * h + (e & 3) + 8 * (e >> 2); }

// F1 / F2:  EPI_ACT: F1 = score seed Y2, F2 = per-row scale (sigma column);  EPI_DACT: F1 = additive Q;  EPI_CHAIN: unused
template <int EPI, int ACT, bool F1, bool F2>
struct WideEpi {
  static constexpr int NLT = EPI == EPI_ACT ? (F2 ? 1 : 0) : EPI == EPI_DACT ? (F1 ? 2 : 1) : 2;   // tensors loaded per element
  static constexpr int NST = EPI == EPI_ACT ? (F1 ? 2 : 1) : EPI == EPI_DACT ? 1 : 2;              // tensors stored per element

  const LinArgs& a;
  int row0, colw;                      // first row of the tile, first column of the wave (64 columns)
  unsigned vY, vY2, vL0, vL1, vRS;     // per-lane byte offsets
  float bcol[2], wsig[2], wfc[2];

  __device__ __forceinline__ WideEpi(const LinArgs& a_, int lane) : a(a_) {
    const int l31 = lane & 31, hh = lane >> 5;
    vY = (unsigned)((4 * hh * a.ldY + l31) * 4);
    vY2 = NST == 2 ? (unsigned)((4 * hh * a.ldY2 + l31) * 4) : 0u;
    vL0 = (EPI != EPI_ACT) ? (unsigned)((4 * hh * a.ldS + l31) * 4) : 0u;
    vL1 = (EPI == EPI_CHAIN) ? (unsigned)((4 * hh * a.ldR + l31) * 4) : (EPI == EPI_DACT && F1) ? (unsigned)((4 * hh * a.ldQ + l31) * 4) : 0u;
    vRS = (unsigned)(16 * hh);
  }

  // column-only operands of the tile (bias, group row-bias, sigma weight, fc weight of the score seed)
  __device__ __forceinline__ void begin_tile(int row0_, int colw_, int lane) {
    row0 = row0_;
    colw = colw_;
    if (EPI == EPI_ACT) {
      const int l31 = lane & 31;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = colw + 32 * j + l31;
        float b = a.bias ? a.bias[col] : 0.f;
        if (a.rowbias) b += a.rowbias[(size_t)(row0 / a.rows_per_group) * a.rowbias_ld + col];   // group is tile-uniform
        bcol[j] = b;
        wsig[j] = F2 ? a.rowscale_w[col] : 0.f;
        wfc[j] = F1 ? a.R[col] : 0.f;
      }
    }
  }

  template <int I, int J, int H>
  __device__ __forceinline__ void loads(float (&l0)[8], float (&l1)[8]) const {
    if (NLT == 0) return;
    const int r0 = row0 + 32 * I;
    const int c0 = colw + 32 * J;
    if (EPI == EPI_ACT) {   // sigma of the row
      const float* p = a.rowscale + r0 + 16 * H;
      gload1<0>(l0[0], vRS, p); gload1<4>(l0[1], vRS, p); gload1<8>(l0[2], vRS, p); gload1<12>(l0[3], vRS, p);
      gload1<32>(l0[4], vRS, p); gload1<36>(l0[5], vRS, p); gload1<40>(l0[6], vRS, p); gload1<44>(l0[7], vRS, p);
      return;
    }
    const int ld0 = a.ldS;
    const float* p0 = a.S + (size_t)(r0 + 16 * H) * ld0 + c0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      gload1<0>(l0[e], vL0, p0);
      p0 += (e == 3) ? (size_t)5 * ld0 : (size_t)ld0;
    }
    if (NLT == 2) {
      const float* T = EPI == EPI_CHAIN ? a.R : a.Q;
      const int ld1 = EPI == EPI_CHAIN ? a.ldR : a.ldQ;
      const float* p1 = T + (size_t)(r0 + 16 * H) * ld1 + c0;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        gload1<0>(l1[e], vL1, p1);
        p1 += (e == 3) ? (size_t)5 * ld1 : (size_t)ld1;
      }
    }
  }

  template <int I, int J, int H>
  __device__ __forceinline__ void math_store(const f32x16& acc16, const float (&l0)[8], const float (&l1)[8], float& csum) const {
    float y[8], y2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = acc16[8 * H + e];
      if (EPI == EPI_ACT) {
        y[e] = act_fwd<ACT>(v + bcol[J] + (F2 ? l0[e] * wsig[J] : 0.f));
        if (F1) y2[e] = -wfc[J] * act_d1<ACT>(y[e]);
      } else if (EPI == EPI_DACT) {
        y[e] = v * act_d1<ACT>(l0[e]) + (F1 ? l1[e] : 0.f);
      } else {
        const float em = (ACT == ACT_SOFTPLUS) ? __expf(-l0[e]) : 0.f;   // 1 - s without cancellation
        y[e] = v * act_d1<ACT>(l0[e]);
        y2[e] = v * l1[e] * em;
      }
      csum += y[e];
    }
    const int r0 = row0 + 32 * I + 16 * H;
    const int c0 = colw + 32 * J;
    float* py = a.Y + (size_t)r0 * a.ldY + c0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      gstore1(vY, y[e], py);
      py += (e == 3) ? (size_t)5 * a.ldY : (size_t)a.ldY;
    }
    if (NST == 2) {
      float* p2 = a.Y2 + (size_t)r0 * a.ldY2 + c0;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        gstore1(vY2, y2[e], p2);
        p2 += (e == 3) ? (size_t)5 * a.ldY2 : (size_t)a.ldY2;
      }
    }
  }

  // half-block HB = 4*J + 2*I + H; its loads were issued one step earlier
  template <int HB>
  __device__ __forceinline__ void step(f32x16 (&acc)[2][2], float (&l0)[2][8], float (&l1)[2][8], float (&csum)[2]) const {
    constexpr int J = HB >> 2, I = (HB >> 1) & 1, H = HB & 1;
    constexpr int N = HB + 1;
    constexpr int NI = (N >> 1) & 1, NJ = (N >> 2) & 1, NH = N & 1;
    if (HB < 7) this->template loads<NI, NJ, NH>(l0[N & 1], l1[N & 1]);
    if (NLT > 0) {
      // younger than this half-block's loads: the stores of the previous one and the loads of the next one
      constexpr int vm = (HB >= 1 ? 8 * NST : 0) + (HB < 7 ? 8 * NLT : 0);
      wait8<vm>(l0[HB & 1]);
      if (NLT == 2) wait8<vm>(l1[HB & 1]);
    }
    this->template math_store<I, J, H>(acc[I][J], l0[HB & 1], l1[HB & 1], csum[J]);
  }

  __device__ __forceinline__ void run(f32x16 (&acc)[2][2], int lane, int tile_row) const {
    float l0[2][8], l1[2][8], csum[2] = {0.f, 0.f};
    this->template loads<0, 0, 0>(l0[0], l1[0]);
    step<0>(acc, l0, l1, csum); step<1>(acc, l0, l1, csum); step<2>(acc, l0, l1, csum); step<3>(acc, l0, l1, csum);
    step<4>(acc, l0, l1, csum); step<5>(acc, l0, l1, csum); step<6>(acc, l0, l1, csum); step<7>(acc, l0, l1, csum);
    if (a.colsum != nullptr) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float c2 = csum[j] + __shfl_xor(csum[j], 32);
        if (lane < 32) a.colsum[(size_t)tile_row * a.Nout + colw + 32 * j + lane] = c2;
      }
    }
  }
};

// Persistent: workgroup b takes tiles b, b + grid, ...  The dispatcher places workgroups b, b+256, b+512 on the same CU
// (scratch/mfma/hwid.hip), and workgroups that start together run their load / MFMA / epilogue phases in lockstep, so
// nobody's MFMAs cover anybody's epilogue.  The second and third resident workgroup of a CU therefore start one third /
// two thirds of a tile period late; the offset persists because all three are bound by the same matrix pipe.
template <int NCH, int EPI, int ACT, bool F1, bool F2, int MINB>
__global__ __launch_bounds__(256, MINB) void linear_wide_kernel(const LinArgs a, int ntiles, int ncp, int stagger) {
  using PG = PanelGeo<NCH>;
  constexpr int NX = PG::NX;
  static_assert(NCH > BDEPTH && NCH % (BDEPTH + 1) == 0, "ring slots must line up across panels");
  __shared__ float lds[2 * WBM * WLDW];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;

  // panel sequence over the (at most two) sources; every K is a multiple of 8 * NCH
  const int np0 = a.src[0].K / (8 * NCH);
  const int np1 = a.nsrc > 1 ? a.src[1].K / (8 * NCH) : 0;
  const int npanels = np0 + np1;

  const unsigned lds0 = (unsigned)(uintptr_t)(lds_f32*)lds;
  const unsigned bvoff = (unsigned)lane * 16u;
  const unsigned rlane = lds0 + (unsigned)((l31 * WLDW + hh * 4) * 4);                         // fragment reads
  const unsigned wlane = lds0 + (unsigned)(((tid / PG::C4) * WLDW + (tid % PG::C4) * 4) * 4);   // panel stores

  WideEpi<EPI, ACT, F1, F2> epi(a, lane);

  if (stagger > 0) {
    const int phase = (int)blockIdx.x >> 8;
    for (int i = 0; i < phase * stagger; ++i) __builtin_amdgcn_s_sleep(127);
  }

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int tr = tile / ncp, cp = tile - tr * ncp;
    const int row0 = tr * WBM;
    const int nb0 = cp * 8 + wave * 2;
    auto panel_b = [&](int p, int j) -> const float* {
      const bool s1 = p >= np0;
      const float* wp = s1 ? a.src[1].wp : a.src[0].wp;
      const int kch = (s1 ? a.src[1].K : a.src[0].K) >> 3;
      const int q = s1 ? p - np0 : p;
      return wp + ((size_t)(nb0 + j) * kch + (size_t)q * NCH) * 256;
    };
    auto panel_x = [&](int p, int& ld) -> const float* {
      const bool s1 = p >= np0;
      ld = s1 ? a.src[1].ld : a.src[0].ld;
      const int q = s1 ? p - np0 : p;
      return (s1 ? a.src[1].x : a.src[0].x) + (size_t)row0 * ld + (size_t)q * (8 * NCH);
    };

#ifdef ARDAE_STAMPS
    const unsigned long long T0 = __builtin_amdgcn_s_memtime();
#endif
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 A[2][2], B[BDEPTH + 1][2], xv[NX];
    WideState s;
    // ---- prologue: panel 0 into LDS buffer 0, weight fragments of chunks 0 .. BDEPTH-1 in flight
    {
      int ld0;
      const float* x0 = panel_x(0, ld0);
      issue_panel_loads<NCH, NX>(xv, x0, ld0, tid);
      s.bcur[0] = panel_b(0, 0);
      s.bcur[1] = panel_b(0, 1);
      const int pn = npanels > 1 ? 1 : 0;
      s.bnxt[0] = panel_b(pn, 0);
      s.bnxt[1] = panel_b(pn, 1);
      s.xnxt = panel_x(pn, s.ldnxt);
      issue_b<0, NCH>(B, s, bvoff);
      issue_b<1, NCH>(B, s, bvoff);
      issue_b<2, NCH>(B, s, bvoff);
      epi.begin_tile(row0, nb0 * 32, lane);
      if (tile != (int)blockIdx.x) __builtin_amdgcn_s_barrier();   // every wave is done reading the previous tile's panels
      wait_panel<2 * BDEPTH, NX>(xv);
      store_panel<NCH, NX>(xv, wlane);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      issue_a<0>(A, rlane);
    }
#ifdef ARDAE_STAMPS
    const unsigned long long T1 = __builtin_amdgcn_s_memtime();
#endif
    int buf = 0;
    for (int p = 0; p + 1 < npanels; ++p) {
      panel<NCH, true, NX>(acc, A, B, xv, s, rlane + buf * WBUF_BYTES, bvoff, tid);
      // next panel -> other buffer (its loads were passed by the in-order vmcnt wait of chunk BDEPTH + 1)
      wait_panel<2 * BDEPTH, NX>(xv);
      buf ^= 1;
      store_panel<NCH, NX>(xv, wlane + buf * WBUF_BYTES);
      s.bcur[0] = s.bnxt[0];
      s.bcur[1] = s.bnxt[1];
      const int pn = p + 2 < npanels ? p + 2 : p + 1;
      s.bnxt[0] = panel_b(pn, 0);
      s.bnxt[1] = panel_b(pn, 1);
      s.xnxt = panel_x(pn, s.ldnxt);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      issue_a<0>(A, rlane + buf * WBUF_BYTES);
    }
    panel<NCH, false, NX>(acc, A, B, xv, s, rlane + buf * WBUF_BYTES, bvoff, tid);
#ifdef ARDAE_STAMPS
    const unsigned long long T2 = __builtin_amdgcn_s_memtime();
#endif

#ifdef ARDAE_DBG_NOEPI
    {
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
      if (sum == 12345.678f) a.Y[tid] = sum;
    }
#else
    epi.run(acc, lane, tr);
#endif
#ifdef ARDAE_STAMPS
    if (a.tile_loss != nullptr && lane == 0) {
      const unsigned long long T3 = __builtin_amdgcn_s_memtime();
      const int iter = (tile - (int)blockIdx.x) / (int)gridDim.x;
      unsigned long long* o = reinterpret_cast<unsigned long long*>(a.tile_loss) + (((size_t)blockIdx.x * 4 + iter) * 4 + wave) * 4;
      o[0] = T0; o[1] = T1; o[2] = T2; o[3] = T3;
    }
#endif
  }
}

bool al16w(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <int NCH, int EPI, int ACT, bool F1, bool F2>
int launch_wide(const LinArgs& a, hipStream_t st) {
  const int ncp = a.Nout / 256;
  const int ntiles = (a.M / WBM) * ncp;
  static const char* genv = getenv("ARDAE_WIDE_GRID");   // experiments: resident workgroups
  const int gmax = genv ? atoi(genv) : 768;
  const int grid = ntiles < gmax ? ntiles : gmax;
  double ksum = 0;
  for (int s = 0; s < a.nsrc; ++s) ksum += a.src[s].K;
  // one tile of one wave = ksum * 128 matrix-pipe cycles; s_sleep(127) ~ 8128 cycles
  static const char* env = getenv("ARDAE_STAGGER");
  const int stagger = env ? atoi(env) : (grid > 256 ? (int)(ksum * 128.0 / 8128.0 + 0.5) : 0);
  if (g_prof_enabled) {
    char name[96];
    snprintf(name, sizeof(name), "linear_wide_kernel<%d, %d, %d, %d, %d>", NCH, EPI, ACT, (int)F1, (int)F2);
    double tensors = 1.0 + (a.Y2 ? 1 : 0) + ((EPI == EPI_DACT || EPI == EPI_CHAIN) ? 1 : 0) + ((EPI == EPI_CHAIN) ? 1 : 0) +
                     ((EPI == EPI_DACT && a.Q) ? 1 : 0);
    prof_begin(st, name, 2.0 * a.M * (double)a.Nout * ksum, 4.0 * ((double)a.M * ksum + tensors * a.M * (double)a.Nout + ksum * a.Nout));
  }
  hipLaunchKernelGGL((linear_wide_kernel<NCH, EPI, ACT, F1, F2, 3>), dim3(grid), dim3(256), 0, st, a, ntiles, ncp, stagger);
  prof_end(st);
  ARDAE_LAUNCH_CHECK();
  return 0;
}

template <int EPI, int ACT, bool F1, bool F2>
int launch_wide_nch(const LinArgs& a, hipStream_t st) {
  bool k64 = true;
  for (int s = 0; s < a.nsrc; ++s) k64 = k64 && (a.src[s].K % 64 == 0);
  return k64 ? launch_wide<8, EPI, ACT, F1, F2>(a, st) : launch_wide<4, EPI, ACT, F1, F2>(a, st);
}

template <int EPI, int ACT>
int launch_wide_flags(const LinArgs& a, hipStream_t st) {
  if constexpr (EPI == EPI_ACT) {
    const bool y2 = a.Y2 != nullptr, rs = a.rowscale != nullptr;
    if (y2 && rs) return launch_wide_nch<EPI, ACT, true, true>(a, st);
    if (y2) return launch_wide_nch<EPI, ACT, true, false>(a, st);
    if (rs) return launch_wide_nch<EPI, ACT, false, true>(a, st);
    return launch_wide_nch<EPI, ACT, false, false>(a, st);
  }
  else if constexpr (EPI == EPI_DACT) {
    if (a.Q) return launch_wide_nch<EPI, ACT, true, false>(a, st);
    return launch_wide_nch<EPI, ACT, false, false>(a, st);
  } else {
    return launch_wide_nch<EPI, ACT, false, false>(a, st);
  }
}

}  // namespace

bool linear_wide_eligible(const LinArgs& a, int epi) {
  if (a.M <= 0 || (a.M % WBM) || a.Nout <= 0 || (a.Nout % 256)) return false;
  if ((int64_t)(a.M / WBM) * (a.Nout / 256) < 128) return false;   // small problems: the 32 x 128 tiling fills the chip better
  for (int s = 0; s < a.nsrc; ++s) {
    if (a.src[s].K % 32 || (a.src[s].ld & 3) || !al16w(a.src[s].x)) return false;
    if ((int64_t)a.src[s].ld * WBM * 4 >= (int64_t)1 << 31) return false;   // 32-bit lane offsets
  }
  if (epi == EPI_DAE_LOSS) return false;   // Nout = z_dim there (narrow geometry)
  if (epi == EPI_ACT && a.rowbias && (a.rows_per_group <= 0 || a.rows_per_group % WBM)) return false;   // group must be tile-uniform
  if (epi == EPI_ACT && a.rowscale && !a.rowscale_w) return false;
  // 32-bit per-lane byte offsets in the epilogue
  const int64_t ldmax = std::max<int64_t>({a.ldY, a.ldY2, a.ldS, a.ldR, a.ldQ});
  if (ldmax * 8 * 4 >= (int64_t)1 << 31) return false;
  return true;
}

int launch_linear_wide(const LinArgs& a, int epi, hipStream_t st) {
  switch (epi) {
    case EPI_ACT:
      if (a.act == ACT_NONE) return launch_wide_flags<EPI_ACT, ACT_NONE>(a, st);
      if (a.act == ACT_RELU) return launch_wide_flags<EPI_ACT, ACT_RELU>(a, st);
      if (a.act == ACT_SOFTPLUS) return launch_wide_flags<EPI_ACT, ACT_SOFTPLUS>(a, st);
      break;
    case EPI_DACT:
      if (a.act == ACT_NONE) return launch_wide_flags<EPI_DACT, ACT_NONE>(a, st);
      if (a.act == ACT_RELU) return launch_wide_flags<EPI_DACT, ACT_RELU>(a, st);
      if (a.act == ACT_SOFTPLUS) return launch_wide_flags<EPI_DACT, ACT_SOFTPLUS>(a, st);
      break;
    case EPI_CHAIN:
      if (a.act == ACT_SOFTPLUS) return launch_wide_flags<EPI_CHAIN, ACT_SOFTPLUS>(a, st);
      break;
  }
  ARDAE_CHECK_ARG(false, "linear_wide: unsupported epilogue/activation combination (epi=%d act=%d)", epi, a.act);
  return -1;
}

}  // namespace ardae
