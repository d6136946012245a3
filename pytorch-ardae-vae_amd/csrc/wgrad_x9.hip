// Weight gradients of the big N-row problems with the FP32 products formed on the BF16 matrix cores (gfx950, round 4):
//     dW[o][i] = sum_pairs sum_m G[m][o] X[m][i],   O % 256 == 0, I % 256 == 0, M % 16 == 0.
//
// Why.  v_mfma_f32_32x32x2_f32 runs on the vector ALU's FP32 lanes: 64 cycles for 2 k, no co-issue with v_* instructions
// (scratch/mfma/samewave.hip) - wgrad_wide_kernel sits at 0.86 of that peak and cannot go further.  The BF16 matrix cores do
// 16 k in 32 cycles (v_mfma_f32_32x32x16_bf16: 2027 TFLOP/s sustained, and up to four v_* instructions per MFMA ride along for free:
// scratch/mfma/bf16x6.hip), i.e. 16 x the k per cycle.  An fp32 number is the EXACT sum of three bf16 numbers
//     x = x_h + x_m + x_l        (x_h = the top 16 bits of x, x_m = the top 16 bits of x - x_h, x_l = x - x_h - x_m: 8 + 8 + 8 significand
//                                 bits, every subtraction exact - the 24-bit significand is CUT into three pieces, no rounding anywhere)
// so g * x = sum over the NINE piece products g_p x_q, each of which is exact in fp32 (8 x 8 bits), accumulated in fp32 inside the
// MFMA - the product itself is formed without any rounding, which an fp32 FMA also guarantees and nothing weaker would.  Nine MFMAs
// of 16 k in 32 cycles each against eight of 2 k in 64: 9 / 16 of the matrix time, measured at least as accurate as the FP32 MFMA
// chain (32 x 32 x 256 products against float64: max 4.4e-7 / rms 7.2e-8 of the result scale, FP32 MFMA 6.1e-7 / 8.7e-8;
// tests/test_wgrad_gpu.py::test_wgrad_x9_* hold the kernel to the fp32 kernels' tolerance).
//
// How.  One workgroup per CU owns a 256 (o) x 256 (i) tile over a contiguous slice of the concatenated (pair, row) range, four
// waves = one per SIMD, each a 128 x 128 block = 4 x 4 accumulators (256 accumulator registers) - the geometry of
// wgrad_wide_kernel, so both operands still cross HBM exactly once per tile.  What differs is the operand path:
//   * staging: thread t owns COLUMN t of the G and of the X chunk (16 rows): sixteen coalesced global_load_dword per operand (a
//     wave reads 256 contiguous bytes of one row per instruction), two chunks ahead of their use;
//   * split: the thread's 8 consecutive rows of a column are exactly one lane's slice of an MFMA operand (8 consecutive k).  It cuts
//     them into the three bf16 planes (v_and + v_sub twice per element, one v_perm per pair and plane: ~5.5 vector-ALU instructions per
//     element, riding between the MFMAs) and writes three 16-byte fragments per 8 rows into LDS at [operand][plane][k group][column]: consecutive lanes,
//     consecutive 16 bytes - conflict-free for the writes and for the fragment reads (one ds_read_b128 per operand block and plane);
//   * a k-step = the whole chunk: 24 fragment reads feed 16 block pairs x 9 piece products = 144 MFMAs, issued piece-pair by
//     piece-pair (smallest first) over the 16 independent accumulators, one s_barrier per chunk, two LDS buffers of 48 KiB;
//   * bias / sigma-column sums come from the staged fp32 values of the thread's column (no LDS reduction needed any more).
// Partial tiles, the fixed-order split reduction and everything around the launch are shared with wgrad_wide.hip.
#include <stdlib.h>

#include <type_traits>

#include "profile.h"
#include "wgrad.h"
#include "wgrad_wide_tiles.h"

namespace ardae {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

constexpr int XT = 256;                       // tile edge
constexpr int XRC = 16;                       // rows per chunk = k of one MFMA
constexpr int X_PLANE_BYTES = 2 * XT * 16;    // [k group (2)][column (256)] x 16 bytes
constexpr int X_OP_BYTES = 3 * X_PLANE_BYTES;
constexpr int X_BUF_BYTES = 2 * X_OP_BYTES;   // G, X: 48 KiB

// s_nop 4: a scalar base the compiler restored with v_readlane needs 5 wait states before a VMEM instruction reads it (linear_wide_kernel.h)
// The staged rows are loaded with inline asm (hipcc sinks ordinary prefetch loads to their first use) into the SAME registers trip after trip:
// the operand is read-write, so every definition is tied to its predecessor's register and the loop's back edge needs no copy - a copy of
// a register whose load is still in flight would read garbage (tools/check_kernel_registers.py looks for exactly that).
// s_nop 4: a scalar base the compiler restored with v_readlane needs 5 wait states before a VMEM instruction reads it (linear_wide_kernel.h)
__device__ __forceinline__ void xload1(float& dst, unsigned voff, const float* sbase) {
  asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2" : "+v"(dst) : "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ void xload4(f32x4& dst, unsigned voff, const float* sbase) {
  asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "+v"(dst) : "v"(voff), "s"(sbase) : "memory");
}
// Loads retire in order: "at most N newer ones may still be in flight" lands everything older.  The wait names NO register: a read-write
// operand would let the compiler give the waited-for value another register - through a copy placed BEFORE the wait, i.e. a copy of a
// register whose load is still in flight (seen in the first version of this kernel; tools/check_kernel_registers.py).  Instead the first
// instructions that touch a freshly landed register are asm volatile statements too (which the compiler keeps in order behind the wait)
// and hand ordinary values on: the top half (first piece of the cut) and a copy for everything else.
template <int N>
__device__ __forceinline__ void xwait() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }
__device__ __forceinline__ void take_landed(const float& loaded, unsigned& top, float& copy) {
  asm volatile("v_and_b32 %0, 0xffff0000, %2\n\tv_mov_b32 %1, %2" : "=&v"(top), "=&v"(copy) : "v"(loaded));
}
__device__ __forceinline__ void take_landed(const f32x4& loaded, f32x4& copy) {
  asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
               : "=&v"(copy[0]), "=&v"(copy[1]), "=&v"(copy[2]), "=&v"(copy[3])
               : "v"(loaded[0]), "v"(loaded[1]), "v"(loaded[2]), "v"(loaded[3]));
}

// One value cut into its three bf16 pieces, as fp32 bit patterns whose low halves are zero: h = the top 16 bits of x (sign, exponent, 7
// mantissa bits), m = the top 16 bits of x - h, l = x - h - m (at most 8 significant bits are left: its low half is zero by itself).
// Both subtractions are exact, so h + m + l == x bit for bit; the pieces share x's sign.  (Pieces that fall into the denormal range - |x| below
// ~1e-30 - lose their low half here when l is a denormal: less than 2^-133 |other operand| per product, far below an fp32 underflow;
// tests/test_abi.py::test_three_piece_cut_is_exact.)
struct Cut3 { unsigned h, m, l; };
__device__ __forceinline__ Cut3 cut3(float x) {
  Cut3 c;
  c.h = __float_as_uint(x) & 0xffff0000u;
  const float r1 = x - __uint_as_float(c.h);
  c.m = __float_as_uint(r1) & 0xffff0000u;
  c.l = __float_as_uint(r1 - __uint_as_float(c.m));
  return c;
}
// two pieces (rows 2 q and 2 q + 1 of an operand slice) -> one dword of a bf16 plane: the high halves of `even` and `odd`
__device__ __forceinline__ unsigned pack_hi(unsigned even, unsigned odd) { return __builtin_amdgcn_perm(odd, even, 0x07060302u); }
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// f(integral_constant<0>) ... f(integral_constant<N - 1>): every index a compile-time constant (register arrays must never be indexed
// by a loop variable the optimiser has yet to unroll: they would be demoted to scratch memory)
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}
constexpr int X9_PA[9] = {2, 1, 2, 0, 2, 1, 0, 1, 0}, X9_PB[9] = {2, 2, 1, 2, 0, 1, 1, 0, 0};      // (plane of G, plane of X), smallest product first

struct X9Chunk {                 // where a chunk's rows live (wave-uniform)
  const float* g; const float* x; const float* rs;
  int ldg, ldx;
  float fb, fr;                  // 1.0 when the chunk contributes to the bias / sigma-weighted column sums
};

__global__ __launch_bounds__(256, 1) void wgrad_x9_kernel(const WwBatchDev batch) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];      // 2 x X_BUF_BYTES
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  const int wo = wave >> 1, wi = wave & 1;
  const int ti = (int)blockIdx.x / batch.splits, split = (int)blockIdx.x - ti * batch.splits;
  // (a copy in registers: read through the reference, every field would be fetched from the kernel-argument segment again after each
  // asm statement with a memory clobber - a chain of scalar-load latencies at the top of every trip)
  const WwTile T = batch.t[ti];

  // this workgroup's slice of the concatenated (pair, row) range, in chunks of 16 rows
  const int cpp = T.M / XRC;
  const int ctot = cpp * T.npairs;
  const int cps = (ctot + batch.splits - 1) / batch.splits;
  const int c_begin = split * cps;
  const int c_end = c_begin + cps < ctot ? c_begin + cps : ctot;

  auto chunk_at = [&](int ch) {
    X9Chunk c;
    const int cc = ch < c_end ? ch : c_end - 1;       // past the end: re-touch the last chunk (never used)
    const int pr = cc >= cpp ? 1 : 0;
    const int m0 = (cc - pr * cpp) * XRC;
    c.ldg = pr ? T.ldG[1] : T.ldG[0];
    c.ldx = pr ? T.ldX[1] : T.ldX[0];
    c.g = (pr ? T.G[1] : T.G[0]) + (size_t)m0 * c.ldg + T.o0;
    c.x = (pr ? T.X[1] : T.X[0]) + (size_t)m0 * c.ldx + T.i0;
    const bool vec = T.want_vec && pr == T.bias_pair && ch < c_end;
    c.fb = vec ? 1.f : 0.f;
    c.fr = (vec && T.rowscale) ? 1.f : 0.f;
    c.rs = T.rowscale ? T.rowscale + m0 : c.g;        // (no sigma: any readable address, weighted by fr = 0)
    return c;
  };
  // thread t stages column t: rows 0 .. 15 of the chunk (a wave's load = 256 contiguous bytes of one row)
  // A chunk's 36 loads in THE order every trip re-issues them in: G rows 0 .. 15 with the sigma quad q behind row 4 q + 3, then X rows
  // 0 .. 15 (loads retire in order, so "the value I am about to use has landed" is a compile-time vmcnt count: see the trip below)
  auto issue_loads = [&](const X9Chunk& c, float (&g)[XRC], float (&x)[XRC], f32x4 (&rs)[4]) {
    const float* pg = c.g;
    const float* px = c.x;
#pragma unroll
    for (int j = 0; j < XRC; ++j) {
      xload1(g[j], (unsigned)tid * 4u, pg); pg += c.ldg;       // (one 64-bit scalar add per row)
      if ((j & 3) == 3) xload4(rs[j >> 2], 0u, c.rs + (j & ~3));
    }
#pragma unroll
    for (int j = 0; j < XRC; ++j) { xload1(x[j], (unsigned)tid * 4u, px); px += c.ldx; }
  };
  float bsum = 0.f, rsum = 0.f;
  // the three fragments of one operand's k group (rows 8 kg .. 8 kg + 7 of this thread's column)
  auto write_group = [&](const u32x4& h, const u32x4& m, const u32x4& l, int op, int kg, unsigned char* buf) {
    unsigned char* p = buf + op * X_OP_BYTES + (kg * XT + tid) * 16;
    *reinterpret_cast<u32x4*>(p) = h;
    *reinterpret_cast<u32x4*>(p + X_PLANE_BYTES) = m;
    *reinterpret_cast<u32x4*>(p + 2 * X_PLANE_BYTES) = l;
  };
  auto stage_group = [&](const float (&v)[XRC], int op, int kg, unsigned char* buf) {
    u32x4 h, m, l;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const Cut3 e = cut3(v[8 * kg + 2 * q]), o = cut3(v[8 * kg + 2 * q + 1]);
      h[q] = pack_hi(e.h, o.h); m[q] = pack_hi(e.m, o.m); l[q] = pack_hi(e.l, o.l);
    }
    write_group(h, m, l, op, kg, buf);
  };

  f32x16 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  if (c_begin < c_end) {
    float gc[XRC], xc[XRC];
    f32x4 rsc[4];
#pragma unroll
    for (int j = 0; j < XRC; ++j) gc[j] = xc[j] = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) rsc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    // ---- prologue: chunk c_begin -> buffer 0 (latency exposed once per launch); chunk c_begin + 1 requested into the staging registers
    {
      const X9Chunk c0 = chunk_at(c_begin);
      issue_loads(c0, gc, xc, rsc);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int j = 0; j < XRC; ++j) asm volatile("" : "+v"(gc[j]), "+v"(xc[j]));
#pragma unroll
      for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(rsc[q]));
      {
        float s0 = 0.f, r0 = 0.f;
#pragma unroll
        for (int j = 0; j < XRC; ++j) { s0 += gc[j]; r0 += gc[j] * rsc[j >> 2][j & 3]; }
        bsum += s0 * c0.fb; rsum += r0 * c0.fr;
      }
      stage_group(gc, 0, 0, lds); stage_group(gc, 0, 1, lds); stage_group(xc, 1, 0, lds); stage_group(xc, 1, 1, lds);
      issue_loads(chunk_at(c_begin + 1), gc, xc, rsc);
      __syncthreads();
    }
    int buf = 0;
    // Fragments: lane (row / column l31 of the block, k group hh), one ds_read_b128 per operand block and plane, read ONCE per chunk and
    // kept in registers (96): re-reading two planes per product would need 64 bytes per clock of LDS bandwidth per CU.
    bf16x8 Af[3][4], Bf[3][4];
    auto read_plane = [&](auto pp, const unsigned char* from) {
      constexpr int p = decltype(pp)::value;
      static_for<4>([&](auto aa) {
        constexpr int a = decltype(aa)::value;
        Af[p][a] = *reinterpret_cast<const bf16x8*>(from + p * X_PLANE_BYTES + (hh * XT + wo * 128 + a * 32 + l31) * 16);
      });
      static_for<4>([&](auto bb) {
        constexpr int b = decltype(bb)::value;
        Bf[p][b] = *reinterpret_cast<const bf16x8*>(from + X_OP_BYTES + p * X_PLANE_BYTES + (hh * XT + wi * 128 + b * 32 + l31) * 16);
      });
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    read_plane(I2{}, lds);      // the first chunk's l and m planes (every later chunk's are read in the tail of the trip before it)
    read_plane(I1{}, lds);
    // One chunk per trip.  The trip's 144 MFMAs - the nine piece products (plane of G, plane of X), smallest first: (l,l) (m,l) (l,m) (h,l)
    // (l,h) (m,m) (h,m) (m,h) (h,h) - run as ONE stream; everything else sits in a GAP between two MFMAs (sched_barrier pins the program
    // order), because only there the vector ALU, the LDS and the memory pipes work beside the matrix cores, and only while a gap's
    // instructions take less than the MFMA's 32 cycles (about four v_* or one memory instruction: scratch/mfma/bf16x6.hip; in-kernel cycle
    // stamps, round 4: 8200 cycles per trip with loads, cuts and fragment reads bunched, 4608 are MFMA issue slots).  Gap by gap:
    //   the staged chunk (ch + 1, in gc / xc: 16 rows of G, 16 of X per thread) is cut pair by pair, seven gaps per pair of rows:
    //     wait + cut first half of row 2 p (and its share of the bias column sum) | second half (and of the sigma column sum) | RE-LOAD
    //     the register with row 2 p of chunk ch + 2 | the same three for row 2 p + 1 | pack the pair into the three planes' dwords;
    //     behind every fourth pair (one k group of an operand) three gaps write its three fragments to the other LDS buffer;
    //   a row's register is re-loaded the moment it has been cut, so a load has a whole trip to land and needs no second register set:
    //     loads retire in order and every trip issues the same 36 in the same order, so when row e is cut exactly 35 newer loads may be in
    //     flight (its sigma quad: 31) - a compile-time vmcnt;
    //   gap 0 also requests the chunk's h-plane fragments (first needed by product 3; l and m planes are in registers already);
    //   gap 124: s_barrier - chunk ch + 1 is complete in the other buffer, nobody reads this one any more;
    //   gap 125 / 128: its l- and m-plane fragments replace this chunk's (dead since products 4 / 7): the next trip starts on registers.
    X9Chunk c1 = chunk_at(c_begin + 1);
#pragma unroll 1
    for (int ch = c_begin; ch < c_end; ++ch) {
      unsigned char* cur = lds + buf * X_BUF_BYTES;
      unsigned char* oth = lds + (buf ^ 1) * X_BUF_BYTES;
      const X9Chunk c2 = chunk_at(ch + 2);
      const float* pgn = c2.g;
      const float* pxn = c2.x;
      u32x4 sh, sm, sl;
      Cut3 ce{0u, 0u, 0u}, co{0u, 0u, 0u};
      float r1 = 0.f, vv = 0.f, cs = 0.f, cr = 0.f;
      f32x4 rsv = {0.f, 0.f, 0.f, 0.f};
      static_for<144>([&](auto nn) {
        constexpr int n = decltype(nn)::value, s = n >> 4, idx = n & 15, a = idx >> 2, b = idx & 3;
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Af[X9_PA[s]][a], Bf[X9_PB[s]][b], acc[a][b], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (n == 0) {
          read_plane(I0{}, cur);
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (n < 124) {
          constexpr int grp = n / 31, r = n % 31;            // k group grp = (operand grp / 2, rows 8 (grp % 2) ..): 4 pairs x 7 gaps + 3 writes
          if constexpr (r < 28) {
            constexpr int pr = r / 7, slot = r % 7, e = 8 * grp + 2 * pr + (slot >= 3 ? 1 : 0), op = e >> 4, j = e & 15;
            float& v = op ? xc[j] : gc[j];
            if constexpr (slot == 0 || slot == 3) {          // first half of the cut: h, and what is left
              xwait<35>();
              Cut3& c = slot == 0 ? ce : co;
              take_landed(v, c.h, vv);
              r1 = vv - __uint_as_float(c.h);
              if constexpr (op == 0) cs += vv;
            } else if constexpr (slot == 1 || slot == 4) {   // second half: m and l (both subtractions are exact)
              Cut3& c = slot == 1 ? ce : co;
              c.m = __float_as_uint(r1) & 0xffff0000u;
              c.l = __float_as_uint(r1 - __uint_as_float(c.m));
              if constexpr (op == 0) {
                if constexpr ((j & 3) == 0) { xwait<31>(); take_landed(rsc[j >> 2], rsv); }
                cr += vv * rsv[j & 3];
              }
            } else if constexpr (slot == 2 || slot == 5) {   // the register is free: row j of chunk ch + 2 (and the sigma quad behind row 4 q + 3)
              if constexpr (op == 0) {
                xload1(v, (unsigned)tid * 4u, pgn); pgn += c2.ldg;
                if constexpr ((j & 3) == 3) xload4(rsc[j >> 2], 0u, c2.rs + (j & ~3));
              } else {
                xload1(v, (unsigned)tid * 4u, pxn); pxn += c2.ldx;
              }
            } else {                                         // the pair's dword of each plane
              sh[pr] = pack_hi(ce.h, co.h); sm[pr] = pack_hi(ce.m, co.m); sl[pr] = pack_hi(ce.l, co.l);
            }
          } else {
            constexpr int pl = r - 28;
            unsigned char* p = oth + (grp >> 1) * X_OP_BYTES + ((grp & 1) * XT + tid) * 16 + pl * X_PLANE_BYTES;
            *reinterpret_cast<u32x4*>(p) = pl == 0 ? sh : pl == 1 ? sm : sl;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (n == 124) {
          __syncthreads();
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (n == 125) {
          read_plane(I2{}, oth);
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (n == 128) {
          read_plane(I1{}, oth);
          __builtin_amdgcn_sched_barrier(0);
        }
      });
      bsum += cs * c1.fb; rsum += cr * c1.fr;
      buf ^= 1;
      c1 = c2;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the re-loads of the last trip: nothing uses them)
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
  if (T.want_vec && T.i0 == 0) {
    T.partial_vec[((size_t)split * 2 + 0) * T.O + T.o0 + tid] = bsum;
    T.partial_vec[((size_t)split * 2 + 1) * T.O + T.o0 + tid] = rsum;
  }
}

}  // namespace

bool wgrad_x9_eligible(const WgradProblem& p) {
  static const bool off = debug_knob("ARDAE_WGRAD_X9") && atoi(debug_knob("ARDAE_WGRAD_X9")) == 0;
  if (off || p.O % XT || p.I % XT || p.M % XRC || p.M < 64 * 32) return false;
  for (int k = 0; k < p.npairs; ++k)
    if ((int64_t)p.ldG[k] * XRC * 4 >= (int64_t)1 << 31 || (int64_t)p.ldX[k] * XRC * 4 >= (int64_t)1 << 31) return false;
  return true;
}

int launch_wgrad_x9(const WwBatchDev& b, hipStream_t st) {
  static const bool attr_set = [] {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_x9_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * X_BUF_BYTES) == hipSuccess;
  }();
  ARDAE_CHECK_ARG(attr_set, "wgrad_x9: cannot reserve %d bytes of LDS", 2 * X_BUF_BYTES);
  hipLaunchKernelGGL(wgrad_x9_kernel, dim3(b.ntiles * b.splits), dim3(256), 2 * X_BUF_BYTES, st, b);
  return 0;
}

}  // namespace ardae
