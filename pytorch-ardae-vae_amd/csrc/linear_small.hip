// Latency-optimised fused linear kernel for the per-image (B-row) problems: the encoder trunk, the context encoder, the
// decoder, the sigma = 0 score pass of the VAE update and their backward chains (models/layers.py:501-515 on a few
// hundred rows).  A step has ~60 of these launches, each a dependent link of a chain, each a few MFLOP: what they cost
// is their own critical path, not throughput.  The 32 x 128 tiling of linear_kernel gives every wave the whole K
// (128 MFMAs = 3.4 us at K = 256) behind an LDS staging round trip; here
//   * one workgroup owns ONE 32 x 32 output block and its four waves split K (32 MFMAs each at K = 256), so a 512 x 256
//     layer runs on 128 CUs instead of 32;
//   * nothing is staged: every wave loads its own A fragments (one float4 per lane per 8-deep chunk, rows are at least
//     32 B contiguous) and packed weight fragments straight into registers, eight chunks of both in flight at once -
//     one memory round trip per 64 k, double buffered for longer K (784-wide first layer, concat inputs);
//   * the epilogue's operands (saved activations, per-image biases) are requested before the K loop;
//   * the four partial accumulators meet in LDS (16 KB) and wave w finishes rows 8w .. 8w+7 of the block, so the epilogue
//     and its stores are spread over the whole workgroup.
// Same operator, same packed-weight image and same epilogue arithmetic (order of the K summation aside) as linear_kernel.
#include <stdlib.h>

#include <algorithm>

#include "elementwise.h"
#include "linear.h"
#include "profile.h"

namespace ardae {
namespace {

constexpr int SB = 8;   // chunks (of 8 k) per register batch

struct SmallSrc {
  const float* x; const float* wp;
  int ld, K, kchunks;
  bool vec;
};

template <int EPI, int ACT>
__device__ __forceinline__ void small_block(const LinArgs& a, int bx, int by, float (*red)[16][64]) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int row0 = bx * 32, nb = by;
  const int arow = min(row0 + l31, a.M - 1);

  // ---- epilogue operands first: they are independent of the K loop
  const int col_raw = nb * 32 + l31;
  const bool cok = col_raw < a.Nout;
  const int col = min(col_raw, a.Nout - 1);
  int rowv[4];
  bool ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int raw = row0 + 8 * wave + 4 * hh + i;
    ok[i] = cok && raw < a.M;
    rowv[i] = min(raw, a.M - 1);
  }
  float o1[4], o2[4], bcol = 0.f, wsig = 0.f, wv = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) o1[i] = 0.f, o2[i] = 0.f;
  if (EPI == EPI_ACT) {
    if (a.bias) bcol = a.bias[col];
    if (a.rowbias) {
#pragma unroll
      for (int i = 0; i < 4; ++i) o1[i] = a.rowbias[(size_t)(rowv[i] / a.rows_per_group) * a.rowbias_ld + col];
    }
    if (a.rowscale) {
      wsig = a.rowscale_w[col];
#pragma unroll
      for (int i = 0; i < 4; ++i) o2[i] = a.rowscale[rowv[i]];
    }
    if (a.Y2) wv = a.R[col];
  } else if (EPI == EPI_DACT) {
#pragma unroll
    for (int i = 0; i < 4; ++i) o1[i] = a.S[(size_t)rowv[i] * a.ldS + col];
    if (a.Q) {
#pragma unroll
      for (int i = 0; i < 4; ++i) o2[i] = a.Q[(size_t)rowv[i] * a.ldQ + col];
    }
  } else {   // EPI_CHAIN
#pragma unroll
    for (int i = 0; i < 4; ++i) o1[i] = a.S[(size_t)rowv[i] * a.ldS + col];
#pragma unroll
    for (int i = 0; i < 4; ++i) o2[i] = a.R[(size_t)rowv[i] * a.ldR + col];
  }

  // ---- K loop over the flat chunk list of both sources; wave w owns a contiguous quarter
  SmallSrc s0, s1;
  s0.x = a.src[0].x; s0.wp = a.src[0].wp; s0.ld = a.src[0].ld; s0.K = a.src[0].K; s0.kchunks = (s0.K + 7) >> 3;
  s0.vec = ((s0.ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(s0.x) & 15) == 0);
  s1 = s0;
  int n1 = 0;
  if (a.nsrc > 1) {
    s1.x = a.src[1].x; s1.wp = a.src[1].wp; s1.ld = a.src[1].ld; s1.K = a.src[1].K; s1.kchunks = (s1.K + 7) >> 3;
    s1.vec = ((s1.ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(s1.x) & 15) == 0);
    n1 = s1.kchunks;
  }
  const int n0 = s0.kchunks, T = n0 + n1;
  const int cbeg = (T * wave) >> 2, cend = (T * (wave + 1)) >> 2;

  auto load_chunk = [&](int c, f32x4& av, f32x4& bv) {
    const bool live = c < cend;
    const int cc = live ? c : cend - 1;       // clamped: a dead chunk re-reads a valid address and contributes A = 0
    const bool second = cc >= n0;
    const SmallSrc& s = second ? s1 : s0;
    const int kc = second ? cc - n0 : cc;
    const int k = kc * 8 + 4 * hh;
    bv = *reinterpret_cast<const f32x4*>(s.wp + ((size_t)nb * s.kchunks + kc) * 256 + lane * 4);
    const float* p = s.x + (size_t)arow * s.ld + k;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (s.vec && k + 4 <= s.K) {
      v = *reinterpret_cast<const f32x4*>(p);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (k + j < s.K) v[j] = p[j];
    }
    if (!live) v = f32x4{0.f, 0.f, 0.f, 0.f};
    av = v;
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  f32x4 a0[SB], b0[SB], a1[SB], b1[SB];
  if (cbeg < cend) {
#pragma unroll
    for (int u = 0; u < SB; ++u) load_chunk(cbeg + u, a0[u], b0[u]);
    for (int c = cbeg; c < cend; c += 2 * SB) {
      const bool more1 = c + SB < cend;
      if (more1) {
#pragma unroll
        for (int u = 0; u < SB; ++u) load_chunk(c + SB + u, a1[u], b1[u]);
      }
#pragma unroll
      for (int u = 0; u < SB; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u][q], b0[u][q], acc, 0, 0, 0);
      if (!more1) break;
      if (c + 2 * SB < cend) {
#pragma unroll
        for (int u = 0; u < SB; ++u) load_chunk(c + 2 * SB + u, a0[u], b0[u]);
      }
#pragma unroll
      for (int u = 0; u < SB; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u][q], b1[u][q], acc, 0, 0, 0);
    }
  }

  // ---- the four partial sums meet in LDS; wave w finishes accumulator registers 4w .. 4w+3
  //      (rows 8w + 4hh + {0..3} of the block, column l31: every store instruction covers two full 128-B lines)
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
  __syncthreads();
  float v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
    v[i] = (red[0][4 * wave + i][lane] + red[1][4 * wave + i][lane]) + (red[2][4 * wave + i][lane] + red[3][4 * wave + i][lane]);

  if (EPI == EPI_ACT) {
    float y[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i] = act_fwd<ACT>(v[i] + bcol + o1[i] + o2[i] * wsig);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (ok[i]) a.Y[(size_t)rowv[i] * a.ldY + col] = y[i];
    if (a.Y2) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (ok[i]) a.Y2[(size_t)rowv[i] * a.ldY2 + col] = -wv * act_d1<ACT>(y[i]);
    }
  } else if (EPI == EPI_DACT) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (ok[i]) a.Y[(size_t)rowv[i] * a.ldY + col] = v[i] * act_d1<ACT>(o1[i]) + o2[i];
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float em = act_ratio<ACT>(o1[i]);   // s'/s (softplus: 1 - s without cancellation)
      if (ok[i]) {
        a.Y[(size_t)rowv[i] * a.ldY + col] = v[i] * act_d1<ACT>(o1[i]);
        a.Y2[(size_t)rowv[i] * a.ldY2 + col] = v[i] * o2[i] * em;
      }
    }
  }
}

template <int EPI, int ACT>
__global__ __launch_bounds__(256) void linear_small_kernel(const LinArgs a) {
  __shared__ float red[4][16][64];
  small_block<EPI, ACT>(a, blockIdx.x, blockIdx.y, red);
}
// TWO independent problems of one epilogue kind in one launch (blockIdx.z picks): links of two dependent chains that do not depend
// on each other - the context encoder and the input encoder of the sigma = 0 score pass (models/graddae/mlp.py:426-431) - cost one
// launch latency per level instead of two
template <int EPI, int ACT>
__global__ __launch_bounds__(256) void linear_small_pair_kernel(const LinArgs a0, const LinArgs a1) {
  const LinArgs& a = blockIdx.z ? a1 : a0;
  if ((int)blockIdx.x * 32 >= a.M || (int)blockIdx.y * 32 >= a.Nout) return;      // the grid covers the larger problem
  __shared__ float red[4][16][64];
  small_block<EPI, ACT>(a, blockIdx.x, blockIdx.y, red);
}

// ---------------------------------------------------------------------------------------------------------------------
// A whole dependent CHAIN of per-image layers in ONE launch.  The per-image (B-row) passes of a step - the sigma = 0 score pass of
// the VAE update, encoder trunk, context encoder and their backward chains - are ~40 launches on the step's critical path, each
// 1 us of MFMA inside 5.5 us (8.5 us at 512 rows) of launch-to-launch latency.  Here the workgroups of a launch walk the LEVELS of a
// chain: a level holds one or two independent problems, workgroup (rb, j) computes 32 x 32 block j of row block rb at every level and
// then waits until all blocks of ITS ROW BLOCK at that level are done (a layer is row-local in its input: block (rb, *) of level l + 1
// reads rows rb of level l only) - a counter per row block, released / acquired at agent scope, instead of a kernel boundary.
// Row block rb's workgroups are placed on ONE XCD (block index i runs on XCD i % 8 - scratch/mfma/hwid.hip - so i = (rb / 8) * 8 nj +
// j * 8 + rb % 8 puts all nj blocks of row block rb on XCD rb % 8): their hand-over stays inside that XCD's L2.
// Deadlock freedom: the grid is at most a few hundred workgroups of 256 threads / 16 KiB LDS and waits only for workgroups of its own
// launch; other kernels finish without it, so every workgroup is scheduled eventually.
constexpr int SC_MAXLEV = 16, SC_MAXPROB = 20;     // the argument block stays under the 4 KiB kernel-argument limit
struct ScProblem {            // what small_block reads of ardae_linear_args, 104 bytes
  const float *x0, *wp0, *x1, *wp1, *bias, *rowbias, *rowscale, *rowscale_w, *S, *Q, *R;
  float *Y, *Y2;
  int ld0, K0, ld1, K1, nsrc, Nout, act, epi, rowbias_ld, rows_per_group, ldS, ldQ, ldR, ldY, ldY2;
};
struct ScLevel { int np, nblk0, nblk; };        // problems, 32-column blocks of problem 0, of both
struct ScArgs {
  int nlev, M, nj, fast, nrb; // nj: workgroups per row block (max blocks of a level); fast: XCD-local hand-over allowed; nrb: row blocks
  unsigned* cnt;              // [3 nrb] zeroed before the launch: level counters | XCC masks | arrivals
  ScLevel lv[SC_MAXLEV];
  ScProblem pr[SC_MAXPROB];
  int first[SC_MAXLEV];       // index of level l's first problem in pr
};

__device__ __forceinline__ LinArgs sc_args(const ScProblem& q, int M) {
  LinArgs a;
  a.M = M; a.Nout = q.Nout; a.nsrc = q.nsrc; a.act = q.act;
  a.src[0].x = q.x0; a.src[0].ld = q.ld0; a.src[0].K = q.K0; a.src[0].wp = q.wp0;
  a.src[1].x = q.x1; a.src[1].ld = q.ld1; a.src[1].K = q.K1; a.src[1].wp = q.wp1;
  a.bias = q.bias; a.rowbias = q.rowbias; a.rowbias_ld = q.rowbias_ld; a.rows_per_group = q.rows_per_group;
  a.rowscale = q.rowscale; a.rowscale_w = q.rowscale_w;
  a.S = q.S; a.ldS = q.ldS; a.R = q.R; a.ldR = q.ldR; a.Q = q.Q; a.ldQ = q.ldQ;
  a.sigma = nullptr; a.eps = nullptr; a.ldeps = 0; a.scale = 0.f;
  a.Y = q.Y; a.ldY = q.ldY; a.Y2 = q.Y2; a.ldY2 = q.ldY2; a.colsum = nullptr; a.tile_loss = nullptr;
  return a;
}

__global__ __launch_bounds__(256) void linear_small_chain_kernel(const ScArgs c) {
  __shared__ float red[4][16][64];
  const int i = (int)blockIdx.x, nj = c.nj;
  const int rb = (i / (8 * nj)) * 8 + (i & 7), j = (i / 8) % nj;     // see above: row block rb's workgroups share an XCD
  if (rb * 32 >= c.M) return;
  // Is the placement what the fast hand-over assumes?  Every workgroup of the row block reports its XCC id (atomics only, no data
  // involved) and waits for the others: one XCC -> the hand-over may stay inside that XCD's L2 (vmcnt + L1 invalidate); anything else
  // (another dispatch order, a partitioned device) -> agent-scope release / acquire (L2 write-back + invalidate: correct, slow).
  bool fast = false;
  {
    __shared__ unsigned s_mask;
    if (threadIdx.x == 0) {
      unsigned xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      __hip_atomic_fetch_or(c.cnt + c.nrb + rb, 1u << (xcc & 15), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(c.cnt + 2 * c.nrb + rb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(c.cnt + 2 * c.nrb + rb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nj) __builtin_amdgcn_s_sleep(1);
      s_mask = __hip_atomic_load(c.cnt + c.nrb + rb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    fast = c.fast && __builtin_popcount(s_mask) == 1;
  }
  unsigned target = 0;
  for (int l = 0; l < c.nlev; ++l) {
    const ScLevel lv = c.lv[l];
    if (j < lv.nblk) {
      const bool second = j >= lv.nblk0;
      const LinArgs a = sc_args(c.pr[c.first[l] + (second ? 1 : 0)], c.M);
      const int cb = second ? j - lv.nblk0 : j;
      const int key = a.act * 4 + (int)(c.pr[c.first[l] + (second ? 1 : 0)].epi);
      switch (key) {
        case ACT_NONE * 4 + EPI_ACT: small_block<EPI_ACT, ACT_NONE>(a, rb, cb, red); break;
        case ACT_RELU * 4 + EPI_ACT: small_block<EPI_ACT, ACT_RELU>(a, rb, cb, red); break;
        case ACT_SOFTPLUS * 4 + EPI_ACT: small_block<EPI_ACT, ACT_SOFTPLUS>(a, rb, cb, red); break;
        case ACT_NONE * 4 + EPI_DACT: small_block<EPI_DACT, ACT_NONE>(a, rb, cb, red); break;
        case ACT_RELU * 4 + EPI_DACT: small_block<EPI_DACT, ACT_RELU>(a, rb, cb, red); break;
        default: small_block<EPI_DACT, ACT_SOFTPLUS>(a, rb, cb, red); break;
      }
    }
    if (l + 1 == c.nlev) break;                     // the kernel boundary orders the last level
    target += (unsigned)lv.nblk;
    // release: this workgroup's stores are visible at agent scope before its arrival is; acquire: nothing of the next level is read
    // before every block of this row block has arrived
    // (every wave releases its OWN stores - a workgroup barrier does not wait for another wave's stores to reach the L2)
    if (fast) {
      // all workgroups of the row block share an XCD (verified above), so the hand-over only has to reach that XCD's L2: the L1 is
      // write-through (a store is complete when the L2 has it: vmcnt), and the consumer drops its CU's L1 lines before reading
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0) {
        if (j < lv.nblk) __hip_atomic_fetch_add(c.cnt + rb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(c.cnt + rb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
      }
      __syncthreads();
      asm volatile("buffer_inv sc0" ::: "memory");
    } else {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __syncthreads();
      if (threadIdx.x == 0) {
        if (j < lv.nblk) __hip_atomic_fetch_add(c.cnt + rb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(c.cnt + rb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
      }
      __syncthreads();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
  }
}

template <int EPI, int ACT>
int launch_small_pair(const LinArgs& a0, const LinArgs& a1, hipStream_t st) {
  if (g_prof_enabled) {
    char name[64];
    snprintf(name, sizeof(name), "linear_small_pair_kernel<%d, %d>", EPI, ACT);
    double fl = 0, by = 0;
    for (const LinArgs* a : {&a0, &a1}) {
      double ksum = 0;
      for (int s = 0; s < a->nsrc; ++s) ksum += a->src[s].K;
      fl += 2.0 * a->M * (double)a->Nout * ksum;
      by += 4.0 * ((double)a->M * ksum + 2.0 * a->M * (double)a->Nout + ksum * a->Nout);
    }
    prof_begin(st, name, fl, by);
  }
  const dim3 grid(std::max(ceil_div(a0.M, 32), ceil_div(a1.M, 32)), std::max(ceil_div(a0.Nout, 32), ceil_div(a1.Nout, 32)), 2);
  hipLaunchKernelGGL((linear_small_pair_kernel<EPI, ACT>), grid, dim3(256), 0, st, a0, a1);
  prof_end(st);
  ARDAE_LAUNCH_CHECK();
  return 0;
}

template <int EPI, int ACT>
int launch_small(const LinArgs& a, hipStream_t st) {
  if (g_prof_enabled) {
    char name[64];
    snprintf(name, sizeof(name), "linear_small_kernel<%d, %d>", EPI, ACT);
    double ksum = 0;
    for (int s = 0; s < a.nsrc; ++s) ksum += a.src[s].K;
    const double tensors = 1.0 + (a.Y2 ? 1 : 0) + ((EPI == EPI_DACT || EPI == EPI_CHAIN) ? 1 : 0) + ((EPI == EPI_CHAIN) ? 1 : 0) +
                           ((EPI == EPI_DACT && a.Q) ? 1 : 0);
    prof_begin(st, name, 2.0 * a.M * (double)a.Nout * ksum, 4.0 * ((double)a.M * ksum + tensors * a.M * (double)a.Nout + ksum * a.Nout));
  }
  hipLaunchKernelGGL((linear_small_kernel<EPI, ACT>), dim3(ceil_div(a.M, 32), ceil_div(a.Nout, 32)), dim3(256), 0, st, a);
  prof_end(st);
  ARDAE_LAUNCH_CHECK();
  return 0;
}

}  // namespace

// Per-image problems only: up to SMALL_MAX_TILES workgroups (beyond that the throughput tilings win), no per-tile side
// outputs (column sums, DAE-loss partials: those belong to N-row launches).  ARDAE_SMALL=0 switches the kernel off.
bool linear_small_eligible(const LinArgs& a, int epi) {
  static const bool on = !(debug_knob("ARDAE_SMALL") && atoi(debug_knob("ARDAE_SMALL")) == 0);
  static const int max_tiles = debug_knob("ARDAE_SMALL_MAX_TILES") ? atoi(debug_knob("ARDAE_SMALL_MAX_TILES")) : 512;
  if (!on || epi == EPI_DAE_LOSS || a.colsum != nullptr || a.tile_loss != nullptr) return false;
  if (a.M <= 0 || a.Nout <= 0) return false;
  if (epi == EPI_CHAIN && a.act == ACT_NONE) return false;
  return (int64_t)ceil_div(a.M, 32) * ceil_div(a.Nout, 32) <= max_tiles;
}

namespace {
bool sc_fill(const LinArgs& a, int epi, ScProblem& q) {
  if (!linear_small_eligible(a, epi) || a.nsrc < 1 || a.nsrc > 2 || !(epi == EPI_ACT || epi == EPI_DACT)) return false;
  if (a.act != ACT_NONE && a.act != ACT_RELU && a.act != ACT_SOFTPLUS) return false;
  q.x0 = a.src[0].x; q.wp0 = a.src[0].wp; q.ld0 = a.src[0].ld; q.K0 = a.src[0].K;
  q.x1 = a.nsrc > 1 ? a.src[1].x : a.src[0].x; q.wp1 = a.nsrc > 1 ? a.src[1].wp : a.src[0].wp; q.ld1 = a.nsrc > 1 ? a.src[1].ld : 0; q.K1 = a.nsrc > 1 ? a.src[1].K : 0;
  q.nsrc = a.nsrc; q.Nout = a.Nout; q.act = a.act; q.epi = epi;
  q.bias = a.bias; q.rowbias = a.rowbias; q.rowbias_ld = a.rowbias_ld; q.rows_per_group = a.rows_per_group; q.rowscale = a.rowscale; q.rowscale_w = a.rowscale_w;
  q.S = a.S; q.ldS = a.ldS; q.Q = a.Q; q.ldQ = a.ldQ; q.R = a.R; q.ldR = a.ldR; q.Y = a.Y; q.ldY = a.ldY; q.Y2 = a.Y2; q.ldY2 = a.ldY2;
  return true;
}
}  // namespace

// A chain of per-image levels (each one or two independent problems on the same M rows; level l + 1 reads outputs of levels <= l of
// its own rows only) in one launch.  counters: 3 ceil(M / 32) unsigned ints of scratch (cleared here with a fill launch).  Falls back to
// one launch per problem when a problem does not qualify for the split-K kernel.
int launch_linear_small_chain(const LinArgs* probs, const int* epis, const int* level_of, int nprob, float* counters, hipStream_t st) {
  ARDAE_CHECK_ARG(probs && epis && level_of && nprob >= 1 && counters, "linear_small_chain: bad arguments");
  static const bool on = !(debug_knob("ARDAE_SMALL_CHAIN") && atoi(debug_knob("ARDAE_SMALL_CHAIN")) == 0);
  ScArgs c;
  memset(&c, 0, sizeof(c));
  static_assert(sizeof(ScArgs) <= 4096, "kernel arguments");
  bool ok = on && nprob <= SC_MAXPROB;
  const int M = probs[0].M;
  int nlev = 0;
  for (int i = 0; ok && i < nprob; ++i) {
    const int l = level_of[i];
    ok = probs[i].M == M && l >= 0 && l < SC_MAXLEV && (i == 0 ? l == 0 : (l == level_of[i - 1] || l == level_of[i - 1] + 1)) && sc_fill(probs[i], epis[i], c.pr[i]);
    if (!ok) break;
    if (l + 1 > nlev) { nlev = l + 1; c.first[l] = i; c.lv[l].np = 0; }
    ok = c.lv[l].np < 2;
    if (!ok) break;
    const int nb = ceil_div(probs[i].Nout, 32);
    if (c.lv[l].np == 0) c.lv[l].nblk0 = nb;
    c.lv[l].nblk += nb;
    c.lv[l].np += 1;
    if (c.lv[l].nblk > c.nj) c.nj = c.lv[l].nblk;
  }
  const int nrb = ceil_div(M, 32);
  ok = ok && nlev >= 2 && (int64_t)nrb * c.nj <= 2048;
  if (!ok) {
    for (int i = 0; i < nprob; ++i) ARDAE_TRY(launch_linear(probs[i], epis[i], st));
    return 0;
  }
  c.nlev = nlev; c.M = M; c.cnt = reinterpret_cast<unsigned*>(counters);
  static const bool fast = !(debug_knob("ARDAE_SMALL_CHAIN_FAST") && atoi(debug_knob("ARDAE_SMALL_CHAIN_FAST")) == 0);
  c.fast = fast ? 1 : 0;
  c.nrb = nrb;
  ARDAE_TRY(launch_fill(counters, 3 * (size_t)nrb, 0.f, st));
  if (g_prof_enabled) {
    double fl = 0, by = 0;
    for (int i = 0; i < nprob; ++i) {
      double ksum = 0;
      for (int s = 0; s < probs[i].nsrc; ++s) ksum += probs[i].src[s].K;
      fl += 2.0 * M * (double)probs[i].Nout * ksum;
      by += 4.0 * ((double)M * ksum + 2.0 * M * (double)probs[i].Nout + ksum * probs[i].Nout);
    }
    char name[64];
    snprintf(name, sizeof(name), "linear_small_chain_kernel x%d", nlev);
    prof_begin(st, name, fl, by);
  }
  const int groups = ceil_div(nrb, 8);
  hipLaunchKernelGGL(linear_small_chain_kernel, dim3(groups * 8 * c.nj), dim3(256), 0, st, c);
  prof_end(st);
  ARDAE_LAUNCH_CHECK();
  return 0;
}

// two independent per-image problems: one launch when both qualify for the split-K kernel with the same epilogue / activation
// (instantiated for what the score pass needs: forward layers), two launches of launch_linear otherwise
int launch_linear_pair(const LinArgs& a0, const LinArgs& a1, int epi, hipStream_t st) {
  if (epi == EPI_ACT && a0.act == a1.act && linear_small_eligible(a0, epi) && linear_small_eligible(a1, epi) && !a0.Y2 && !a1.Y2) {
    if (a0.act == ACT_SOFTPLUS) return launch_small_pair<EPI_ACT, ACT_SOFTPLUS>(a0, a1, st);
    if (a0.act == ACT_RELU) return launch_small_pair<EPI_ACT, ACT_RELU>(a0, a1, st);
  }
  ARDAE_TRY(launch_linear(a0, epi, st));
  return launch_linear(a1, epi, st);
}

int launch_linear_small(const LinArgs& a, int epi, hipStream_t st) {
  switch (epi) {
    case EPI_ACT:
      if (a.act == ACT_NONE) return launch_small<EPI_ACT, ACT_NONE>(a, st);
      if (a.act == ACT_RELU) return launch_small<EPI_ACT, ACT_RELU>(a, st);
      if (a.act == ACT_SOFTPLUS) return launch_small<EPI_ACT, ACT_SOFTPLUS>(a, st);
      if (a.act == ACT_ELU) return launch_small<EPI_ACT, ACT_ELU>(a, st);
      if (a.act == ACT_TANH) return launch_small<EPI_ACT, ACT_TANH>(a, st);
      if (a.act == ACT_LEAKY) return launch_small<EPI_ACT, ACT_LEAKY>(a, st);
      break;
    case EPI_DACT:
      if (a.act == ACT_NONE) return launch_small<EPI_DACT, ACT_NONE>(a, st);
      if (a.act == ACT_RELU) return launch_small<EPI_DACT, ACT_RELU>(a, st);
      if (a.act == ACT_SOFTPLUS) return launch_small<EPI_DACT, ACT_SOFTPLUS>(a, st);
      if (a.act == ACT_ELU) return launch_small<EPI_DACT, ACT_ELU>(a, st);
      if (a.act == ACT_TANH) return launch_small<EPI_DACT, ACT_TANH>(a, st);
      if (a.act == ACT_LEAKY) return launch_small<EPI_DACT, ACT_LEAKY>(a, st);
      break;
    case EPI_CHAIN:
      if (a.act == ACT_SOFTPLUS) return launch_small<EPI_CHAIN, ACT_SOFTPLUS>(a, st);
      if (a.act == ACT_RELU) return launch_small<EPI_CHAIN, ACT_RELU>(a, st);
      if (a.act == ACT_ELU) return launch_small<EPI_CHAIN, ACT_ELU>(a, st);
      if (a.act == ACT_TANH) return launch_small<EPI_CHAIN, ACT_TANH>(a, st);
      if (a.act == ACT_LEAKY) return launch_small<EPI_CHAIN, ACT_LEAKY>(a, st);
      break;
  }
  ARDAE_CHECK_ARG(false, "linear_small: unsupported epilogue/activation combination (epi=%d act=%d)", epi, a.act);
  return -1;
}

}  // namespace ardae
