// Software-pipelined persistent FP32-MFMA linear kernel for the big N-row layers (gfx950).
//
// One workgroup per CU, four waves = one per SIMD, each wave a 64 x 64 block of the 64 x 256 output tile (2 x 2
// v_mfma_f32_32x32x2_f32 accumulators); the workgroup walks tiles b, b + grid, ...
//
// What the design rests on (all measured on MI355X, sources under scratch/mfma/):
//  * dep.hip       155 TFLOP/s from any number of waves per SIMD: one wave with four accumulators saturates the pipe.
//  * samewave.hip  FP32 MFMA and the vector ALU are ONE resource: every v_* instruction, from this wave or another,
//                  adds its 4 cycles to the matrix time.  So VALU work is priced, never hidden - keep it minimal.
//  * comem2.hip    a co-resident wave's VALU instructions are arbitrated 1:1 against 64-cycle MFMAs: an epilogue
//                  running beside another wave's K loop crawls (900 v_* -> 900 MFMA slots), whereas VALU-free streams
//                  (saddr loads/stores, SALU pointer bumps) overlap with MFMAs completely.
//  * ldasm.hip     a wave that waits for operand loads right before using them loses 20-40 % whatever the occupancy
//                  (round-robin MFMA arbitration phase-locks the waves); loads must be in flight inside the wave's own
//                  MFMA stream.  hipcc does not keep such a schedule (it sinks prefetches to their use), hence the
//                  inline-asm loads and s_waitcnt below.
//
// So each wave runs ONE continuous instruction stream in which every memory access is issued long before its use and
// nothing but the epilogue arithmetic touches the vector ALU:
//   * weight fragments: global_load_dwordx4 (saddr form) from the packed, L2-resident image, BDEPTH = 3 chunks ahead
//     in a 4-slot register ring;
//   * activation fragments: ds_read_b128, one chunk ahead, running across panel and tile boundaries;
//   * activation panels (64 rows x 64 k) HBM -> registers at chunk 0 of the previous panel, -> LDS at chunk NCH-3, one
//     s_barrier at chunk NCH-2; three LDS buffers make the single barrier per panel sufficient;
//   * the epilogue's saved-activation operands (S, R / Q, sigma) are loaded during the first half of the tile's LAST
//     K panel into registers (one wave per SIMD owns all 512), so the epilogue itself is arithmetic + stores;
//   * stores drain under the next tile's K loop.
// vmcnt retires in order, so every wait below is a compile-time count of the younger operations (Sched).
//
// Shapes: M % 64 == 0, Nout % 256 == 0, every source K % 32 == 0 (panels of 64 or 32), 16-byte aligned rows, row-bias
// groups that are multiples of 64 rows.  Anything else goes to linear_kernel (linear.hip), which handles ragged edges.
#pragma once
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "linear.h"
#include "profile.h"

namespace ardae {
namespace wide {

constexpr int WBM = 64;                       // rows per tile
constexpr int WLDW = 68;                      // LDS row stride (floats): conflict-free ds_read_b128 fragments
constexpr int WBUF_BYTES = WBM * WLDW * 4;    // one K panel (<= 64 wide)
constexpr int NBUF = 3;                       // panel ring
constexpr int BDEPTH = 3;                     // weight-fragment prefetch distance (chunks); ring of BDEPTH + 1 slots

typedef __attribute__((address_space(3))) float lds_f32;

// s_nop 4: a scalar base that the compiler produced with a VALU instruction (v_readlane of a spilled SGPR, v_readfirstlane)
// needs 5 wait states before a VMEM instruction may read it; the hazard recogniser does not look inside inline asm
// (seen as a memory fault once SGPR spills appeared).  It costs issue cycles of this wave only, not matrix-pipe time.
template <int OFF>
__device__ __forceinline__ void gload4(f32x4& dst, unsigned voff, const float* sbase) {
  asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}
// (NOP = false would be legal where the base is known to be SALU-written, but a spilled SGPR comes back through
// v_readlane at the compiler's discretion, so the wait states stay everywhere)
template <int OFF, bool NOP = true>
__device__ __forceinline__ void gload1(float& dst, unsigned voff, const float* sbase) {
  if (NOP) asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
  else asm volatile("global_load_dword %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}
template <bool NOP = true>
__device__ __forceinline__ void gstore1(unsigned voff, float v, float* sbase) {
  if (NOP) asm volatile("s_nop 4\n\tglobal_store_dword %0, %1, %2" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
  else asm volatile("global_store_dword %0, %1, %2" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
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
template <int VM>
__device__ __forceinline__ void wait16(float* v) {
  asm volatile("s_waitcnt vmcnt(%16)"
               : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]),
                 "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15])
               : "n"(VM)
               : "memory");
}

// panel geometry for NCH chunks of 8 k: 2*NCH float4 per row, 256 / (2*NCH) rows per pass, NX passes
template <int NCH>
struct PanelGeo {
  static constexpr int C4 = 2 * NCH;
  static constexpr int RPP = 256 / C4;
  static constexpr int NX = WBM / RPP;
};

// The order in which one chunk issues its vector-memory operations (all of them retire in order):
//   panel P, chunk k:  B(k + BDEPTH) x2 | k == 0: next panel's activation loads x NX
//                      | deferred stores of one half-block of the PREVIOUS tile x 8*NST (panels 0 .. NP-2, every STRIDE-th chunk)
//                      | last panel: k == 0: row-bias x2, k < OPC: epilogue operands of HPC half-blocks x 8*NLT each
// and the waits derived from it.  NLT / NST = tensors loaded / stored per output element by the epilogue.
// With a single panel per tile (NP == 1) there is no room to defer: the epilogue stores at once, and the first chunks of
// the next tile see >= 64 younger stores (the count saturates at the 6-bit maximum).
template <int NCH, int NP, int NLT, int NST, bool HP, int NXL = PanelGeo<NCH>::NX>
struct Sched {
  static constexpr int NX = NXL;            // activation-panel loads per chunk 0 (0: the activations already sit in LDS)
  // Deferring keeps the previous tile's results (64*NST registers) alive next to this tile's operands (64*NLT): with
  // three or more tensors the wave runs out of its 256 architectural VGPRs, and a compiler spill of a register whose load
  // is still in flight reads garbage (seen as wrong Y2 rows in the first 8 rows of a tile) - those kernels store at once.
  static constexpr bool DEFER = NP >= 2 && NLT + NST <= 2;
  static constexpr int OPC = NCH / 2;       // the operand loads end with chunk OPC-1 of the last panel (NCH/2 chunks to land)
  // Operand loads (64*NLT per tile, in half-block order): with one tensor they fit one per MFMA in chunks 0..OPC-1 of the
  // last panel; with two tensors and at least two panels they start a panel earlier (no stores ride in those kernels), so
  // that an MFMA never has more than one of them behind it (two per MFMA stalled the K loop by ~6 k cycles per tile).
  static constexpr int NOPS = 64 * NLT;
  static constexpr bool OP_EARLY = NLT == 2 && NP >= 2;
  static constexpr int OP_G0 = OP_EARLY ? (NP - 2) * NCH : (NP - 1) * NCH;          // first global chunk carrying operand loads
  static constexpr int OP_NCHK = (NP - 1) * NCH + OPC - OP_G0;                       // number of chunks carrying them
  static constexpr int OP_PER = NOPS == 0 ? 0 : (NOPS + OP_NCHK - 1) / OP_NCHK;      // operand loads per chunk
  // [begin, end) of the operand-load indices issued in chunk k of panel p
  static constexpr int op_begin(int p, int k) {
    const int g = p * NCH + k - OP_G0;
    if (NOPS == 0 || g < 0 || g >= OP_NCHK) return 0;
    return g * OP_PER < NOPS ? g * OP_PER : NOPS;
  }
  static constexpr int op_end(int p, int k) {
    const int g = p * NCH + k - OP_G0;
    if (NOPS == 0 || g < 0 || g >= OP_NCHK) return 0;
    return (g + 1) * OP_PER < NOPS ? (g + 1) * OP_PER : NOPS;
  }
  static constexpr int XW = NCH - 3;        // chunk that moves the next panel registers -> LDS
  static constexpr int BAR = NCH - 2;       // chunk that holds the barrier
  static constexpr int STRIDE = DEFER ? (NP - 1) * NCH / 8 : 1;   // chunks between deferred half-block stores
  // half-block whose deferred stores ride in chunk k of panel p (-1: none)
  static constexpr int store_hb(int p, int k) {
    if (!DEFER || !HP || p >= NP - 1) return -1;   // HP: there is a previous tile whose results wait in registers
    const int g = p * NCH + k;
    return (g % STRIDE == 0 && g / STRIDE < 8) ? g / STRIDE : -1;
  }
  static constexpr int extras(int p, int k) {
    const bool last = p == NP - 1;
    return (k == 0 ? NX : 0) + (store_hb(p, k) >= 0 ? 8 * NST : 0) + (last && k == 0 ? 2 : 0) + (op_end(p, k) - op_begin(p, k));
  }
  // operations younger than B(c) when chunk c of panel p waits for it.  B(c) was issued first thing in chunk c - BDEPTH;
  // chunks before 0 belong to the previous panel (the previous tile's last one for p == 0)
  static constexpr int vm_frag(int p, int c) {
    if (!DEFER && p == 0 && c < BDEPTH) return 63;
    int n = 0;
    for (int k = c - BDEPTH; k < c; ++k) {
      const bool prev = k < 0;
      if (k > c - BDEPTH) n += 2;
      n += extras(prev ? (p + NP - 1) % NP : p, prev ? k + NCH : k);
    }
    return n > 63 ? 63 : n;
  }
  // operations younger than the activation loads of chunk 0 when chunk XW moves them to LDS (before its own B issue)
  static constexpr int vm_panel(int p) {
    int n = extras(p, 0) - NX;
    for (int k = 1; k < XW; ++k) n += 2 + extras(p, k);
    return n > 63 ? 63 : n;
  }
  // operations younger than the last epilogue operand when the epilogue starts
  static constexpr int vm_epi() { return 2 * (NCH - OPC); }
};

struct WideState {
  const float* bcur[2];   // packed-weight pointers of the two 32-column blocks at chunk 0 of the current panel
  const float* bnxt[2];   // ... of the next panel (next tile's first panel after the last one)
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

// ---------------------------------------------------------------------------------------------------------------------
// Epilogue.  The accumulator of a 32x32 block puts column l&31 and rows (r&3) + 8(r>>2) + 4(l>>5) in lane l, so operands
// and results move as dwords: one instruction = two full 128-byte lines.  Addressing is the saddr form (row base in
// SGPRs, per-lane column offset in one VGPR per tensor): no vector-ALU work per access.
// F1 / F2:  EPI_ACT: F1 = score seed Y2, F2 = per-row scale (sigma column);  EPI_DACT: F1 = additive Q;  EPI_CHAIN: unused
// ---------------------------------------------------------------------------------------------------------------------
template <int EPI, int ACT, bool F1, bool F2>
struct WideEpi {
  static constexpr int NLT = EPI == EPI_ACT ? (F2 ? 1 : 0) : EPI == EPI_DACT ? (F1 ? 2 : 1) : 2;   // tensors loaded per element
  static constexpr int NST = EPI == EPI_ACT ? (F1 ? 2 : 1) : EPI == EPI_DACT ? 1 : 2;              // tensors stored per element
  static constexpr bool SIGMA_OPERAND = EPI == EPI_ACT;   // its one loaded operand (F2) is the per-row sigma
  static constexpr bool CHAIN = EPI == EPI_CHAIN;

  const LinArgs& a;
  unsigned vY, vY2, vL0, vL1, vRS, vC;   // per-lane byte offsets
  float bcol[2], wsig[2], wfc[2];
  int colw_loaded;

  __device__ __forceinline__ WideEpi(const LinArgs& a_, int lane) : a(a_), colw_loaded(-1) {
    const int l31 = lane & 31, hh = lane >> 5;
    vY = (unsigned)((4 * hh * a.ldY + l31) * 4);
    vY2 = NST == 2 ? (unsigned)((4 * hh * (a.Y2 ? a.ldY2 : a.ldY) + l31) * 4) : 0u;   // absent Y2 (fused chains): Y again
    vL0 = (EPI != EPI_ACT) ? (unsigned)((4 * hh * a.ldS + l31) * 4) : 0u;
    vL1 = (EPI == EPI_CHAIN) ? (unsigned)((4 * hh * a.ldR + l31) * 4) : (EPI == EPI_DACT && F1) ? (unsigned)((4 * hh * a.ldQ + l31) * 4) : 0u;
    vRS = (unsigned)(16 * hh);
    vC = (unsigned)(l31 * 4);
    bcol[0] = bcol[1] = wsig[0] = wsig[1] = wfc[0] = wfc[1] = 0.f;
  }

  // column-only operands (bias, sigma weight, fc weight of the score seed): once per column panel
  __device__ __forceinline__ void column_operands(int colw, int lane) {
    if (EPI != EPI_ACT || colw == colw_loaded) return;
    colw_loaded = colw;
    const int l31 = lane & 31;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = colw + 32 * j + l31;
      bcol[j] = a.bias ? a.bias[col] : 0.f;
      wsig[j] = (F2 && a.rowscale_w) ? a.rowscale_w[col] : 0.f;
      wfc[j] = (F1 && a.R) ? a.R[col] : 0.f;
    }
  }

  // group row-bias of the tile (two columns per lane); a dummy in-bounds load when there is none keeps the counts static
  template <int JJ>
  __device__ __forceinline__ void issue_rowbias_one(float& rbj, int row0, int colw) const {
    const float* p = (EPI == EPI_ACT && a.rowbias) ? a.rowbias + (size_t)(row0 / a.rows_per_group) * a.rowbias_ld + colw
                                                     : a.Y + (size_t)row0 * a.ldY + colw;
    gload1<128 * JJ>(rbj, vC, p);
  }

  // operand loads of half-block HB = 4*J + 2*I + H (rows 32I + 16H + {0..3, 8..11} + 4hh, columns 32J + l31)
  template <int HB>
  __device__ __forceinline__ void issue_operands(float* l0, float* l1, int row0, int colw) const {
    if (NLT == 0) return;
    constexpr int J = HB >> 2, I = (HB >> 1) & 1, H = HB & 1;
    const int r0 = row0 + 32 * I + 16 * H;
    const int c0 = colw + 32 * J;
    if (EPI == EPI_ACT) {   // sigma of the row
      const float* p = (a.rowscale ? a.rowscale : a.src[0].x) + r0;
      gload1<0>(l0[0], vRS, p); gload1<4>(l0[1], vRS, p); gload1<8>(l0[2], vRS, p); gload1<12>(l0[3], vRS, p);
      gload1<32>(l0[4], vRS, p); gload1<36>(l0[5], vRS, p); gload1<40>(l0[6], vRS, p); gload1<44>(l0[7], vRS, p);
      return;
    }
    const int ld0 = a.ldS;
    const float* p0 = a.S + (size_t)r0 * ld0 + c0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      gload1<0>(l0[e], vL0, p0);
      p0 += (e == 3) ? (size_t)5 * ld0 : (size_t)ld0;
    }
    if (NLT == 2) {
      const float* T = EPI == EPI_CHAIN ? a.R : a.Q;
      const int ld1 = EPI == EPI_CHAIN ? a.ldR : a.ldQ;
      const float* p1 = T + (size_t)r0 * ld1 + c0;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        gload1<0>(l1[e], vL1, p1);
        p1 += (e == 3) ? (size_t)5 * ld1 : (size_t)ld1;
      }
    }
  }

  // results replace the operands in place: l0[e] <- Y, l1[e] <- Y2
  template <int HB>
  __device__ __forceinline__ void math(const f32x16& acc16, float* l0, float* l1, float brow, float& csum) const {
    constexpr int J = HB >> 2, H = HB & 1;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = acc16[8 * H + e];
      float y, y2 = 0.f;
      if (EPI == EPI_ACT) {
        // a fused chain instantiates F1/F2 for all its layers; a layer without the operand skips it (uniform select)
        y = act_fwd<ACT>((F2 && a.rowscale) ? __builtin_fmaf(l0[e], wsig[J], v + brow) : v + brow);
        if (F1) y2 = a.Y2 ? -wfc[J] * act_d1<ACT>(y) : y;
      } else if (EPI == EPI_DACT) {
        y = v * act_d1<ACT>(l0[e]) + (F1 ? l1[e] : 0.f);
      } else {
        const float em = (ACT == ACT_SOFTPLUS) ? fast_exp(-l0[e]) : 0.f;   // 1 - s without cancellation
        y = v * act_d1<ACT>(l0[e]);
        y2 = v * l1[e] * em;
      }
      csum += y;
      l0[e] = y;
      if (NST == 2) l1[e] = y2;
    }
  }

  template <int HB>
  __device__ __forceinline__ void stores(const float* y, const float* y2, int row0, int colw) const {
    constexpr int J = HB >> 2, I = (HB >> 1) & 1, H = HB & 1;
    const int r0 = row0 + 32 * I + 16 * H;
    const int c0 = colw + 32 * J;
    float* py = a.Y + (size_t)r0 * a.ldY + c0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      gstore1(vY, y[e], py);
      py += (e == 3) ? (size_t)5 * a.ldY : (size_t)a.ldY;
    }
    if (NST == 2) {
      const int ld2 = a.Y2 ? a.ldY2 : a.ldY;
      float* p2 = (a.Y2 ? a.Y2 : a.Y) + (size_t)r0 * ld2 + c0;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        gstore1(vY2, y2[e], p2);
        p2 += (e == 3) ? (size_t)5 * ld2 : (size_t)ld2;
      }
    }
  }

  __device__ __forceinline__ void store_all(const float* y, const float* y2, int row0, int colw) const {
    stores<0>(y + 0, y2 + 0, row0, colw); stores<1>(y + 8, y2 + 8, row0, colw); stores<2>(y + 16, y2 + 16, row0, colw);
    stores<3>(y + 24, y2 + 24, row0, colw); stores<4>(y + 32, y2 + 32, row0, colw); stores<5>(y + 40, y2 + 40, row0, colw);
    stores<6>(y + 48, y2 + 48, row0, colw); stores<7>(y + 56, y2 + 56, row0, colw);
  }

  template <int VM, bool STORE_NOW>
  __device__ __forceinline__ void run(f32x16 (&acc)[2][2], float* l0, float* l1, float (&rb)[2], int lane, int row0, int colw, int tile_row) const {
    // one wait for everything the epilogue reads (issued >= NCH/2 chunks ago)
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(rb[0]), "+v"(rb[1]) : "n"(VM) : "memory");
    if (NLT >= 1) { wait16<VM>(l0); wait16<VM>(l0 + 16); wait16<VM>(l0 + 32); wait16<VM>(l0 + 48); }
    if (NLT == 2) { wait16<VM>(l1); wait16<VM>(l1 + 16); wait16<VM>(l1 + 32); wait16<VM>(l1 + 48); }
    const bool has_rb = EPI == EPI_ACT && a.rowbias != nullptr;
    const float br0 = bcol[0] + (has_rb ? rb[0] : 0.f), br1 = bcol[1] + (has_rb ? rb[1] : 0.f);
    float csum[2] = {0.f, 0.f};
    math<0>(acc[0][0], l0 + 0, l1 + 0, br0, csum[0]);
    math<1>(acc[0][0], l0 + 8, l1 + 8, br0, csum[0]);
    math<2>(acc[1][0], l0 + 16, l1 + 16, br0, csum[0]);
    math<3>(acc[1][0], l0 + 24, l1 + 24, br0, csum[0]);
    math<4>(acc[0][1], l0 + 32, l1 + 32, br1, csum[1]);
    math<5>(acc[0][1], l0 + 40, l1 + 40, br1, csum[1]);
    math<6>(acc[1][1], l0 + 48, l1 + 48, br1, csum[1]);
    math<7>(acc[1][1], l0 + 56, l1 + 56, br1, csum[1]);
    if (STORE_NOW) store_all(l0, l1, row0, colw);
    if (a.colsum != nullptr) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float c2 = csum[j] + __shfl_xor(csum[j], 32);
        if (lane < 32) a.colsum[(size_t)tile_row * a.Nout + colw + 32 * j + lane] = c2;
      }
    }
  }
};

// everything a chunk needs besides the register arrays
template <int NCH, class EPI_T>
struct PanelCtx {
  const WideState& s;
  const EPI_T& epi;
  unsigned raddr, raddr_next, waddr_next, bvoff;
  int tid, row0, colw, prev_row0;   // prev_row0 < 0: no previous tile (nothing to store yet)
  const LinArgs* prev;              // layer whose results the deferred stores carry (the same layer outside of fused chains)
  unsigned pvY, pvY2;               // its per-lane store offsets
};

// One chunk of panel P: wait for its fragments, then 16 MFMAs with the chunk's memory instructions spread evenly behind
// them (an MFMA keeps the pipe busy for 64 cycles while the wave is free to issue; with one wave per SIMD nobody else
// would fill a gap, and more than a handful of instructions behind one MFMA is a gap).  The instructions of a chunk, in
// the issue order Sched assumes:
//   2 fragment reads A(c+1) | 2 weight loads B(c+3) | c == 0: NX panel loads | NX panel LDS writes (c == XW)
//   | 8*NST deferred stores | last panel: 2 row-bias loads, HPC*8*NLT epilogue-operand loads
// XLDS: the layer's whole activation tile (64 rows x K, row stride FC_TLD floats) already sits in LDS (fused chains:
// written there by the previous layer's epilogue) - no panel loads, no ring, no barrier inside the K loop.
constexpr int FC_TLD = 260;
template <int C, int P, int NCH, int NP, bool HP, class EPI_T, bool XLDS = false>
struct ChunkOps {
  using SC = Sched<NCH, NP, EPI_T::NLT, EPI_T::NST, HP, XLDS ? 0 : PanelGeo<NCH>::NX>;
  static constexpr int NX = PanelGeo<NCH>::NX;
  static constexpr int NLT = EPI_T::NLT, NST = EPI_T::NST;
  static constexpr bool LAST = P == NP - 1;
  static constexpr int SHB = SC::store_hb(P, C);
  static constexpr int n_a = (XLDS && LAST && C == NCH - 1) ? 0 : 2;   // no next chunk inside this layer's tile
  static constexpr int n_b = 2, n_x = (C == 0 && !XLDS) ? NX : 0, n_w = (C == SC::XW && !XLDS) ? NX : 0, n_st = SHB >= 0 ? 8 * NST : 0,
                       n_rb = (LAST && C == 0) ? 2 : 0, n_op = SC::op_end(P, C) - SC::op_begin(P, C);
  static constexpr int o_a = 0, o_b = o_a + n_a, o_x = o_b + n_b, o_w = o_x + n_x, o_st = o_w + n_w, o_rb = o_st + n_st,
                       o_op = o_rb + n_rb, total = o_op + n_op;
  static constexpr int PER = (total + 15) / 16;   // instructions behind each MFMA

  // running scalar pointers of the store / operand streams (bumped by one row per instruction)
  struct Ptrs {
    float* py;
    float* py2;
    const float* p0;
    const float* p1;
  };

  template <int K>
  static __device__ __forceinline__ void op(f32x4 (&A)[2][2], f32x4 (&B)[BDEPTH + 1][2], f32x4 (&xv)[NX], float* l0, float* l1, float (&rb)[2],
                                            const PanelCtx<NCH, EPI_T>& x, Ptrs& q) {
    using PG = PanelGeo<NCH>;
    const LinArgs& a = x.epi.a;
    if constexpr (K < o_b) {   // fragment read of the next chunk (chunk 0 of the next panel after the last one)
      constexpr int i = K - o_a;
      if constexpr (XLDS) lds_read4<(P * NCH + C + 1) * 32 + i * 32 * FC_TLD * 4>(A[(C + 1) & 1][i], x.raddr);
      else if constexpr (C + 1 < NCH) lds_read4<(C + 1) * 32 + i * 32 * WLDW * 4>(A[(C + 1) & 1][i], x.raddr);
      else lds_read4<i * 32 * WLDW * 4>(A[0][i], x.raddr_next);
    } else if constexpr (K < o_x) {
      constexpr int j = K - o_b;
      constexpr int CB = C + BDEPTH, slot = CB % (BDEPTH + 1), CC = CB < NCH ? CB : CB - NCH;
      const float* base = CB < NCH ? x.s.bcur[j] : x.s.bnxt[j];
      gload4<(CC & 3) * 1024>(B[slot][j], x.bvoff, base + (CC >> 2) * 1024);
    } else if constexpr (K < o_w) {
      constexpr int u = K - o_x;
      const unsigned voff = (unsigned)(((x.tid / PG::C4) * x.s.ldnxt + (x.tid % PG::C4) * 4) * 4);
      gload4<0>(xv[u], voff, x.s.xnxt + (size_t)u * PG::RPP * x.s.ldnxt);
    } else if constexpr (K < o_st) {
      constexpr int u = K - o_w;
      if constexpr (u == 0) wait_panel<SC::vm_panel(P), NX>(xv);
      lds_write4<u * PG::RPP * WLDW * 4>(x.waddr_next, xv[u]);
    } else if constexpr (K < o_rb) {
      constexpr int k = K - o_st, tns = k / 8, e = k % 8;
      constexpr int J = SHB >> 2, I = (SHB >> 1) & 1, H = SHB & 1;
      const LinArgs& pa = *x.prev;
      if constexpr (tns == 0) {
        if constexpr (e == 0) q.py = pa.Y + (size_t)(x.prev_row0 + 32 * I + 16 * H) * pa.ldY + x.colw + 32 * J;
        gstore1(x.pvY, l0[8 * SHB + e], q.py);
        q.py += (e == 3) ? (size_t)5 * pa.ldY : (size_t)pa.ldY;
      } else {
        const int ld2 = pa.Y2 ? pa.ldY2 : pa.ldY;
        if constexpr (e == 0) q.py2 = (pa.Y2 ? pa.Y2 : pa.Y) + (size_t)(x.prev_row0 + 32 * I + 16 * H) * ld2 + x.colw + 32 * J;
        gstore1(x.pvY2, l1[8 * SHB + e], q.py2);
        q.py2 += (e == 3) ? (size_t)5 * ld2 : (size_t)ld2;
      }
    } else if constexpr (K < o_op) {
      constexpr int k = K - o_rb;
      if constexpr (k == 0) x.epi.template issue_rowbias_one<0>(rb[0], x.row0, x.colw);
      else x.epi.template issue_rowbias_one<1>(rb[1], x.row0, x.colw);
    } else {
      constexpr int k = K - o_op;
      constexpr int idx = SC::op_begin(P, C) + k;                         // index in the tile's operand-load sequence
      constexpr int HB = idx / (8 * NLT), tns = (idx / 8) % NLT, e = idx % 8;
      constexpr int J = HB >> 2, I = (HB >> 1) & 1, H = HB & 1;
      constexpr bool fresh = e == 0 || k == 0;                            // first load of its stream in this chunk: full address
      constexpr int roff = 32 * I + 16 * H + (e & 3) + 8 * (e >> 2);      // row of element e inside the tile (without the lane part)
      if constexpr (EPI_T::SIGMA_OPERAND) {   // EPI_ACT with a per-row scale: sigma of the row
        constexpr int off = ((e & 3) + 8 * (e >> 2)) * 4;
        gload1<off>(l0[8 * HB + e], x.epi.vRS, (a.rowscale ? a.rowscale : a.src[0].x) + x.row0 + 32 * I + 16 * H);
      } else if constexpr (tns == 0) {
        if constexpr (fresh) q.p0 = a.S + (size_t)(x.row0 + roff) * a.ldS + x.colw + 32 * J;
        gload1<0>(l0[8 * HB + e], x.epi.vL0, q.p0);
        q.p0 += (e == 3) ? (size_t)5 * a.ldS : (size_t)a.ldS;
      } else {
        const float* T = EPI_T::CHAIN ? a.R : a.Q;
        const int ld1 = EPI_T::CHAIN ? a.ldR : a.ldQ;
        if constexpr (fresh) q.p1 = T + (size_t)(x.row0 + roff) * ld1 + x.colw + 32 * J;
        gload1<0>(l1[8 * HB + e], x.epi.vL1, q.p1);
        q.p1 += (e == 3) ? (size_t)5 * ld1 : (size_t)ld1;
      }
    }
  }

  // the instructions behind MFMA S
  template <int S, int R = 0>
  static __device__ __forceinline__ void slot(f32x4 (&A)[2][2], f32x4 (&B)[BDEPTH + 1][2], f32x4 (&xv)[NX], float* l0, float* l1, float (&rb)[2],
                                              const PanelCtx<NCH, EPI_T>& x, Ptrs& q) {
    if constexpr (R < PER && S * PER + R < total) {
      op<S * PER + R>(A, B, xv, l0, l1, rb, x, q);
      slot<S, R + 1>(A, B, xv, l0, l1, rb, x, q);
    }
  }

  template <int S>
  static __device__ __forceinline__ void steps(f32x16 (&acc)[2][2], f32x4 (&A)[2][2], f32x4 (&B)[BDEPTH + 1][2], f32x4 (&xv)[NX], float* l0,
                                               float* l1, float (&rb)[2], const PanelCtx<NCH, EPI_T>& x, Ptrs& q) {
    if constexpr (S < 16) {
      constexpr int kq = S >> 2, i = (S >> 1) & 1, j = S & 1, bslot = C % (BDEPTH + 1);
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[C & 1][i][kq], B[bslot][j][kq], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      slot<S>(A, B, xv, l0, l1, rb, x, q);
      __builtin_amdgcn_sched_barrier(0);
      steps<S + 1>(acc, A, B, xv, l0, l1, rb, x, q);
    }
  }

  static __device__ __forceinline__ void run(f32x16 (&acc)[2][2], f32x4 (&A)[2][2], f32x4 (&B)[BDEPTH + 1][2], f32x4 (&xv)[NX], float* l0,
                                             float* l1, float (&rb)[2], const PanelCtx<NCH, EPI_T>& x) {
    constexpr int bslot = C % (BDEPTH + 1);
    wait_frag<SC::vm_frag(P, C)>(A[C & 1][0], A[C & 1][1], B[bslot][0], B[bslot][1]);
    if (C == SC::BAR && !XLDS) __builtin_amdgcn_s_barrier();   // the wait above covered this wave's ds_writes (lgkmcnt(0))
    __builtin_amdgcn_sched_barrier(0);
    Ptrs q{nullptr, nullptr, nullptr, nullptr};
    steps<0>(acc, A, B, xv, l0, l1, rb, x, q);
  }
};

template <int P, int NCH, int NP, bool HP, class EPI_T, bool XLDS = false>
__device__ __forceinline__ void panel(f32x16 (&acc)[2][2], f32x4 (&A)[2][2], f32x4 (&B)[BDEPTH + 1][2], f32x4 (&xv)[PanelGeo<NCH>::NX],
                                      float* l0, float* l1, float (&rb)[2], const PanelCtx<NCH, EPI_T>& x) {
  ChunkOps<0, P, NCH, NP, HP, EPI_T, XLDS>::run(acc, A, B, xv, l0, l1, rb, x);
  ChunkOps<1, P, NCH, NP, HP, EPI_T, XLDS>::run(acc, A, B, xv, l0, l1, rb, x);
  ChunkOps<2, P, NCH, NP, HP, EPI_T, XLDS>::run(acc, A, B, xv, l0, l1, rb, x);
  ChunkOps<3, P, NCH, NP, HP, EPI_T, XLDS>::run(acc, A, B, xv, l0, l1, rb, x);
  if constexpr (NCH == 8) {
    ChunkOps<4, P, NCH, NP, HP, EPI_T, XLDS>::run(acc, A, B, xv, l0, l1, rb, x);
    ChunkOps<5, P, NCH, NP, HP, EPI_T, XLDS>::run(acc, A, B, xv, l0, l1, rb, x);
    ChunkOps<6, P, NCH, NP, HP, EPI_T, XLDS>::run(acc, A, B, xv, l0, l1, rb, x);
    ChunkOps<7, P, NCH, NP, HP, EPI_T, XLDS>::run(acc, A, B, xv, l0, l1, rb, x);
  }
}

// panels P .. NP-1 of one tile, straight-line: values defined by the asm loads must never meet at a control-flow join
// (the compiler would reconcile them with register copies - of registers whose loads are still in flight)
template <int P, int NCH, int NP, bool HP, class EPI_T, class PB, class PX>
__device__ __forceinline__ void tile_panels(f32x16 (&acc)[2][2], f32x4 (&A)[2][2], f32x4 (&B)[BDEPTH + 1][2], f32x4 (&xv)[PanelGeo<NCH>::NX],
                                            float* l0, float* l1, float (&rb)[2], WideState& s, const EPI_T& epi, int& buf, unsigned rlane,
                                            unsigned wlane, unsigned bvoff, int tid, int row0, int colw, int prev_row0, int tile, int tnext,
                                            const PB& panel_b, const PX& panel_x) {
  if constexpr (P < NP) {
    constexpr bool lastp = P + 1 == NP;
    const int tn = lastp ? tnext : tile;
    constexpr int pn = lastp ? 0 : P + 1;
    s.bnxt[0] = panel_b(tn, pn, 0);
    s.bnxt[1] = panel_b(tn, pn, 1);
    s.xnxt = panel_x(tn, pn, s.ldnxt);
    const int bnext = buf + 1 == NBUF ? 0 : buf + 1;
    const PanelCtx<NCH, EPI_T> x{s, epi, rlane + buf * WBUF_BYTES, rlane + bnext * WBUF_BYTES, wlane + bnext * WBUF_BYTES, bvoff, tid, row0, colw, prev_row0,
                                 &epi.a, epi.vY, epi.vY2};
    panel<P, NCH, NP, HP>(acc, A, B, xv, l0, l1, rb, x);
    s.bcur[0] = s.bnxt[0];
    s.bcur[1] = s.bnxt[1];
    buf = bnext;
    tile_panels<P + 1, NCH, NP, HP>(acc, A, B, xv, l0, l1, rb, s, epi, buf, rlane, wlane, bvoff, tid, row0, colw, prev_row0, tile, tnext, panel_b, panel_x);
  }
}

// NP = K panels per tile (compile time: the tile body is one basic block).  grid % ncp == 0, so a workgroup keeps its
// column panel and the column-only epilogue operands are loaded once.
template <int NCH, int NP, int EPI, int ACT, bool F1, bool F2>
__global__ __launch_bounds__(256, 1) void linear_wide_kernel(const LinArgs a, int ntiles, int ncp) {
  using PG = PanelGeo<NCH>;
  using EPI_T = WideEpi<EPI, ACT, F1, F2>;
  using SC = Sched<NCH, NP, EPI_T::NLT, EPI_T::NST, true>;
  constexpr int NX = PG::NX;
  static_assert(NCH > BDEPTH && NCH % (BDEPTH + 1) == 0 && NCH % 2 == 0, "ring slots must line up across panels");
  static_assert(!SC::DEFER || ((NP - 1) * NCH) % 8 == 0, "deferred stores: one half-block every STRIDE chunks");
  __shared__ float lds[NBUF * WBM * WLDW];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;

  // panel sequence over the (at most two) sources; every K is a multiple of 8 * NCH
  const int np0 = a.src[0].K / (8 * NCH);

  const unsigned lds0 = (unsigned)(uintptr_t)(lds_f32*)lds;
  const unsigned bvoff = (unsigned)lane * 16u;
  const unsigned rlane = lds0 + (unsigned)((l31 * WLDW + hh * 4) * 4);                         // fragment reads
  const unsigned wlane = lds0 + (unsigned)(((tid / PG::C4) * WLDW + (tid % PG::C4) * 4) * 4);   // panel stores

  const int cp = (int)blockIdx.x % ncp;
  const int colw = (cp * 8 + wave * 2) * 32;
  EPI_T epi(a, lane);
  epi.column_operands(colw, lane);

  // panel p of tile t: weight and activation pointers (wave-uniform)
  auto panel_b = [&](int tile, int p, int j) -> const float* {
    const int nb0 = cp * 8 + wave * 2;
    const bool s1 = p >= np0;
    const float* wp = s1 ? a.src[1].wp : a.src[0].wp;
    const int kch = (s1 ? a.src[1].K : a.src[0].K) >> 3;
    const int q = s1 ? p - np0 : p;
    return wp + ((size_t)(nb0 + j) * kch + (size_t)q * NCH) * 256;
  };
  auto panel_x = [&](int tile, int p, int& ld) -> const float* {
    const int row0 = (tile / ncp) * WBM;
    const bool s1 = p >= np0;
    ld = s1 ? a.src[1].ld : a.src[0].ld;
    const int q = s1 ? p - np0 : p;
    return (s1 ? a.src[1].x : a.src[0].x) + (size_t)row0 * ld + (size_t)q * (8 * NCH);
  };

  f32x4 A[2][2], B[BDEPTH + 1][2], xv[NX];
  float l0[64], l1[64], rb[2];
  WideState s;
  int tile = blockIdx.x;
  // ---- prologue: first panel into LDS buffer 0, weight fragments of chunks 0 .. BDEPTH-1 and fragment set 0 in flight
  {
    int ld0;
    const float* x0 = panel_x(tile, 0, ld0);
    issue_panel_loads<NCH, NX>(xv, x0, ld0, tid);
    s.bcur[0] = s.bnxt[0] = panel_b(tile, 0, 0);
    s.bcur[1] = s.bnxt[1] = panel_b(tile, 0, 1);
    issue_b<0, NCH>(B, s, bvoff);
    issue_b<1, NCH>(B, s, bvoff);
    issue_b<2, NCH>(B, s, bvoff);
    wait_panel<0, NX>(xv);   // everything landed (the first tile's waits may then be as loose as any later tile's)
    store_panel<NCH, NX>(xv, wlane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    issue_a<0>(A, rlane);
  }
  int buf = 0;
  int prev_row0 = -1;
#ifdef ARDAE_STAMPS
  unsigned long long t_k = 0, t_e = 0;
  const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
#endif
  // one tile: K loop (with the previous tile's stores riding along when HP) + epilogue arithmetic
  auto do_tile = [&](auto hp_tag) {
    constexpr bool HP = decltype(hp_tag)::value;
#ifdef ARDAE_STAMPS
    const unsigned long long T0 = __builtin_amdgcn_s_memtime();
#endif
    const int tr = tile / ncp;
    const int row0 = tr * WBM;
    const int tnext = tile + (int)gridDim.x < ntiles ? tile + (int)gridDim.x : tile;   // none: re-touch this tile (never used)
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    tile_panels<0, NCH, NP, HP>(acc, A, B, xv, l0, l1, rb, s, epi, buf, rlane, wlane, bvoff, tid, row0, colw, prev_row0, tile, tnext, panel_b, panel_x);
#ifdef ARDAE_STAMPS
    const unsigned long long T1 = __builtin_amdgcn_s_memtime();
    t_k += T1 - T0;
#endif
    // Nothing may be in flight across the epilogue: it is compiler-scheduled code under register pressure, and a spill or
    // copy of a register whose load has not landed would read garbage.  The next tile's first fragments (issued 1-3
    // chunks ago) are therefore waited for here; the epilogue's own operands landed long ago.
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                 : "+v"(A[0][0]), "+v"(A[0][1]), "+v"(B[0][0]), "+v"(B[0][1]), "+v"(B[1][0]), "+v"(B[1][1]), "+v"(B[2][0]), "+v"(B[2][1])
                 :
                 : "memory");
#ifdef ARDAE_DBG_NOEPI
    {
      float sum = rb[0];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
      if (sum == 12345.678f) a.Y[tid] = sum;
    }
#else
    epi.template run<0, !SC::DEFER>(acc, l0, l1, rb, lane, row0, colw, tr);
    if (SC::DEFER) prev_row0 = row0;
#endif
#ifdef ARDAE_STAMPS
    t_e += __builtin_amdgcn_s_memtime() - T1;
#endif
  };
  // the first tile has no predecessor whose stores could ride in its K loop: its own copy of the tile body
  do_tile(std::false_type{});
  for (tile += gridDim.x; tile < ntiles; tile += gridDim.x) do_tile(std::integral_constant<bool, SC::DEFER>{});
#ifndef ARDAE_DBG_NOEPI
  if (SC::DEFER && prev_row0 >= 0) epi.store_all(l0, l1, prev_row0, colw);
#endif
  // the last panel prefetched a (dummy) next panel: drain before the registers die
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#ifdef ARDAE_STAMPS
  if (a.tile_loss != nullptr && lane == 0) {
    unsigned long long* o = reinterpret_cast<unsigned long long*>(a.tile_loss) + ((size_t)blockIdx.x * 4 + wave) * 4;
    o[0] = t_k; o[1] = t_e; o[2] = __builtin_amdgcn_s_memtime() - t_begin; o[3] = t_begin;
  }
#endif
}

int wide_grid(int ntiles, int ncp);

template <int NCH, int NP, int EPI, int ACT, bool F1, bool F2>
int launch_wide(const LinArgs& a, hipStream_t st) {
  const int ncp = a.Nout / 256;
  const int ntiles = (a.M / WBM) * ncp;
  const int grid = wide_grid(ntiles, ncp);
  if (g_prof_enabled) {
    char name[96];
    snprintf(name, sizeof(name), "linear_wide_kernel<%d, %d, %d, %d, %d, %d>", NCH, NP, EPI, ACT, (int)F1, (int)F2);
    double ksum = 0;
    for (int s = 0; s < a.nsrc; ++s) ksum += a.src[s].K;
    double tensors = 1.0 + (a.Y2 ? 1 : 0) + ((EPI == EPI_DACT || EPI == EPI_CHAIN) ? 1 : 0) + ((EPI == EPI_CHAIN) ? 1 : 0) +
                     ((EPI == EPI_DACT && a.Q) ? 1 : 0);
    prof_begin(st, name, 2.0 * a.M * (double)a.Nout * ksum, 4.0 * ((double)a.M * ksum + tensors * a.M * (double)a.Nout + ksum * a.Nout));
  }
  hipLaunchKernelGGL((linear_wide_kernel<NCH, NP, EPI, ACT, F1, F2>), dim3(grid), dim3(256), 0, st, a, ntiles, ncp);
  prof_end(st);
  ARDAE_LAUNCH_CHECK();
  return 0;
}


// explicit instantiations live in linear_wide_inst_*.hip
#define ARDAE_WIDE_FOR_GEOS(X, EPI, ACT, F1, F2) \
  X(8, 4, EPI, ACT, F1, F2)                       \
  X(4, 1, EPI, ACT, F1, F2)
#define ARDAE_WIDE_FOR_ACT_FLAGS(X, ACT)          \
  ARDAE_WIDE_FOR_GEOS(X, EPI_ACT, ACT, false, false) \
  ARDAE_WIDE_FOR_GEOS(X, EPI_ACT, ACT, false, true)  \
  ARDAE_WIDE_FOR_GEOS(X, EPI_ACT, ACT, true, false)  \
  ARDAE_WIDE_FOR_GEOS(X, EPI_ACT, ACT, true, true)
#define ARDAE_WIDE_FOR_DACT_FLAGS(X, ACT)          \
  ARDAE_WIDE_FOR_GEOS(X, EPI_DACT, ACT, false, false) \
  ARDAE_WIDE_FOR_GEOS(X, EPI_DACT, ACT, true, false)
#define ARDAE_WIDE_EXTERN(NCH, NP, EPI, ACT, F1, F2) extern template int launch_wide<NCH, NP, EPI, ACT, F1, F2>(const LinArgs&, hipStream_t);
#define ARDAE_WIDE_INSTANTIATE(NCH, NP, EPI, ACT, F1, F2) template int launch_wide<NCH, NP, EPI, ACT, F1, F2>(const LinArgs&, hipStream_t);
#ifndef ARDAE_WIDE_INST_TU
ARDAE_WIDE_FOR_ACT_FLAGS(ARDAE_WIDE_EXTERN, ACT_NONE)
ARDAE_WIDE_FOR_ACT_FLAGS(ARDAE_WIDE_EXTERN, ACT_RELU)
ARDAE_WIDE_FOR_ACT_FLAGS(ARDAE_WIDE_EXTERN, ACT_SOFTPLUS)
ARDAE_WIDE_FOR_DACT_FLAGS(ARDAE_WIDE_EXTERN, ACT_NONE)
ARDAE_WIDE_FOR_DACT_FLAGS(ARDAE_WIDE_EXTERN, ACT_RELU)
ARDAE_WIDE_FOR_DACT_FLAGS(ARDAE_WIDE_EXTERN, ACT_SOFTPLUS)
ARDAE_WIDE_FOR_GEOS(ARDAE_WIDE_EXTERN, EPI_CHAIN, ACT_SOFTPLUS, false, false)
#endif

}  // namespace wide
}  // namespace ardae
