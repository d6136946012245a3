// Warp-specialised persistent variant of the fused linear kernel for the big N-row layers (gfx950).
//
// Why: with every wave doing load -> MFMA -> epilogue, the co-resident workgroups of a CU run their phases in
// lockstep (in-kernel stamps: K loop 140k cycles for 33k cycles of MFMA issue, matrix pipe idle ~50 %).  Here the
// two waves of each SIMD have complementary roles (MI355X_MICROARCH "two waves per SIMD": matrix beside memory):
//
//   waves 0-3 (consumers, one per SIMD): nothing but LDS fragment reads, L2 weight-fragment loads (register ring, three
//       8-deep K chunks ahead) and v_mfma_f32_32x32x2_f32; after a tile they drop the accumulators into an LDS staging
//       buffer and go straight on to the next tile;
//   waves 4-7 (producers, one per SIMD): stream the NEXT tile's activations HBM -> registers -> LDS and run the
//       PREVIOUS tile's epilogue from the staging buffer (saved-activation loads, softplus / derivative math, 16-byte
//       coalesced stores), i.e. every HBM access of the kernel.
//
// One workgroup (512 threads) per CU, persistent over 32-row tiles; activation tile and staging buffer are double
// buffered, one barrier per tile.  Same operator and arguments as linear_kernel (linear.hip); launch_linear() routes
// here when the shape allows (single source, K <= 256, Nout <= 256, 16-byte aligned operands, no column sums).
#include "linear.h"
#include "profile.h"

namespace ardae {
namespace {

constexpr int WS_BM = 32;          // rows per tile
constexpr int WS_KMAX = 256;
constexpr int WS_LDX = WS_KMAX + 4;   // activation tile row stride (floats): conflict-free ds_read_b128 fragments
constexpr int WS_LDC = 256 + 4;       // staging row stride

constexpr int WS_THREADS = 768;       // 4 consumer waves + 8 producer waves (3 waves per SIMD)
constexpr int WS_PROD = WS_THREADS - 256;

// Eligibility (linear_ws_eligible) guarantees: M % 32 == 0, K % 32 == 0, K <= 256, Nout % 4 == 0, 16-byte aligned
// operands -> no ragged tiles, so the hot paths below are branch-free (a divergent or data-dependent branch around a
// load makes hipcc fall back to s_waitcnt vmcnt(0) at every join, which serialised the first version on HBM latency).
template <int EPI, int ACT, int RING>
__global__ __launch_bounds__(WS_THREADS, 3) void linear_ws_kernel(const LinArgs a, int ntiles) {
  __shared__ float Xs[2][WS_BM * WS_LDX];
  __shared__ float Cs[2][WS_BM * WS_LDC];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool consumer = wave < 4;
  const int K = a.src[0].K, ld = a.src[0].ld;
  const float* __restrict__ X = a.src[0].x;
  const float* __restrict__ wp = a.src[0].wp;
  const int nch = K >> 3;                 // 8-deep K chunks (multiple of 4)
  const int nblk_total = (a.Nout + 31) >> 5;
  const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;

  // ------------------------------------------------------------------ producer state (ptid = 0..511)
  const int ptid = tid - 256;
  const int c4n = K >> 2;                         // float4 per activation row: 8..64, a power of two or 3*2^k... (K%32==0)
  constexpr int XV = (WS_BM * (WS_KMAX / 4) + WS_PROD - 1) / WS_PROD;   // 4 float4 per producer thread per tile
  // activation tile: float4 index idx = ptid + 512 u -> (row, col4); indices past the tile are clamped onto row 31 (they
  // re-load and re-store a valid element: same address, same value - no branch)
  int xoff_g[XV], xoff_l[XV];
#pragma unroll
  for (int u = 0; u < XV; ++u) {
    const int idx = ptid + WS_PROD * u;
    int r = idx / c4n;
    const int c = (idx - r * c4n) << 2;
    r = min(r, WS_BM - 1);
    xoff_g[u] = r * ld + c;
    xoff_l[u] = r * WS_LDX + c;
  }
  f32x4 xv[XV];
  // epilogue mapping: rows erow + 8j (j = 0..3), columns ecol .. ecol+3
  constexpr int EV = WS_BM / (WS_PROD / 64);
  const int ecol = min((ptid & 63) << 2, a.Nout - 4);   // clamped: threads past Nout duplicate the last float4 (benign)
  const int erow = ptid >> 6;
  f32x4 sv[EV], qv[EV];
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f}, wsig4 = bias4, wfc4 = bias4;
  const bool has_rb = EPI == EPI_ACT && a.rowbias != nullptr, has_rs = EPI == EPI_ACT && a.rowscale != nullptr;
  const bool has_y2 = EPI == EPI_ACT && a.Y2 != nullptr, has_q = EPI == EPI_DACT && a.Q != nullptr;
  if (!consumer && EPI == EPI_ACT) {
    if (a.bias) bias4 = *reinterpret_cast<const f32x4*>(a.bias + ecol);
    if (a.rowscale_w) wsig4 = *reinterpret_cast<const f32x4*>(a.rowscale_w + ecol);
    if (has_y2) wfc4 = *reinterpret_cast<const f32x4*>(a.R + ecol);
  }

  // ------------------------------------------------------------------ consumer state
  const int l31 = lane & 31, hh = lane >> 5;
  const int nb0 = wave * 2;                        // two 32-column blocks per consumer wave
  const bool cons_active = consumer && nb0 < nblk_total;
  const float* bptr[2] = {nullptr, nullptr};
  if (consumer) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int nb = min(nb0 + j, nblk_total - 1);
      bptr[j] = wp + (size_t)nb * nch * 256 + lane * 4;
    }
  }

  // ------------------------------------------------------------------ pipeline
  //   step i:  consumers  MFMA(tile i) from Xs[i&1] -> accumulators -> Cs[i&1]
  //            producers  request X(tile i+1); epilogue(tile i-1) from Cs[(i-1)&1] with operands requested in step i-1;
  //                       request the operands of tile i; X(tile i+1) -> Xs[(i+1)&1]
  if (!consumer) {
    const float* xt = X + (size_t)blockIdx.x * WS_BM * ld;
#pragma unroll
    for (int u = 0; u < XV; ++u) xv[u] = *reinterpret_cast<const f32x4*>(xt + xoff_g[u]);
#pragma unroll
    for (int u = 0; u < XV; ++u) *reinterpret_cast<f32x4*>(&Xs[0][xoff_l[u]]) = xv[u];
  }
  __syncthreads();
#ifdef ARDAE_STAMPS
  unsigned long long acc_a = 0, acc_b = 0, acc_c = 0, acc_d = 0;
#define WS_STAMP(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#else
#define WS_STAMP(v)
#endif
  for (int i = 0; i <= my_tiles; ++i) {
    const int tile = blockIdx.x + i * gridDim.x;
    WS_STAMP(ts0);
    if (consumer) {
      if (i < my_tiles && cons_active) {
        const float* xb = &Xs[i & 1][l31 * WS_LDX + hh * 4];
        f32x16 acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        // weight-fragment ring: RING register sets, loads issued RING-1 chunks ahead of their MFMAs (the producers keep the
        // CU's vector-memory pipeline busy with HBM streams, so an L2 hit can take several thousand cycles to come back);
        // nch % RING == 0, and the prefetches past the end re-read the last chunk (clamped) instead of branching
        f32x4 bw[RING][2], aw[2];
#pragma unroll
        for (int u = 0; u < RING - 1; ++u)
#pragma unroll
          for (int j = 0; j < 2; ++j) bw[u][j] = *reinterpret_cast<const f32x4*>(bptr[j] + (size_t)min(u, nch - 1) * 256);
        aw[0] = *reinterpret_cast<const f32x4*>(xb);
        for (int kc = 0; kc < nch; kc += RING) {
#pragma unroll
          for (int u = 0; u < RING; ++u) {
            const int kpre = min(kc + u + RING - 1, nch - 1);
#pragma unroll
            for (int j = 0; j < 2; ++j) bw[(u + RING - 1) % RING][j] = *reinterpret_cast<const f32x4*>(bptr[j] + (size_t)kpre * 256);
            aw[(u + 1) & 1] = *reinterpret_cast<const f32x4*>(xb + min(kc + u + 1, nch - 1) * 8);
            // pin the prefetches here: without the barrier hipcc sinks each weight load down to its first use
            // (load; s_waitcnt vmcnt(0); mfma - no lookahead at all)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
              for (int j = 0; j < 2; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[u & 1][q], bw[u][j][q], acc[j], 0, 0, 0);
          }
        }
        WS_STAMP(ts1);
#ifdef ARDAE_STAMPS
        acc_a += ts1 - ts0;
#endif
        // accumulators -> staging (row = (r&3) + 8(r>>2) + 4hh, col = 32 nb + l31)
        float* cb = &Cs[i & 1][(4 * hh) * WS_LDC + nb0 * 32 + l31];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) cb[((r & 3) + 8 * (r >> 2)) * WS_LDC + j * 32] = acc[j][r];
      }
    } else {
      const bool more = i + 1 < my_tiles;
      const int tnext = more ? tile + (int)gridDim.x : tile;          // clamped: the drain steps re-request a valid tile
      const float* xt = X + (size_t)min(tnext, ntiles - 1) * WS_BM * ld;
#pragma unroll
      for (int u = 0; u < XV; ++u) xv[u] = *reinterpret_cast<const f32x4*>(xt + xoff_g[u]);
      if (i >= 1) {
        // ---- epilogue of tile i-1 (operands sv/qv were requested during step i-1)
        const int row0 = (tile - (int)gridDim.x) * WS_BM;
        const float* cbuf = Cs[(i - 1) & 1];
        f32x4 v[EV], rb4[EV];
        float rs[EV];
#pragma unroll
        for (int j = 0; j < EV; ++j) v[j] = *reinterpret_cast<const f32x4*>(&cbuf[(erow + 8 * j) * WS_LDC + ecol]);
        if (has_rb) {
#pragma unroll
          for (int j = 0; j < EV; ++j)
            rb4[j] = *reinterpret_cast<const f32x4*>(a.rowbias + (size_t)((row0 + erow + 8 * j) / a.rows_per_group) * a.rowbias_ld + ecol);
        } else {
#pragma unroll
          for (int j = 0; j < EV; ++j) rb4[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (has_rs) {
#pragma unroll
          for (int j = 0; j < EV; ++j) rs[j] = a.rowscale[row0 + erow + 8 * j];
        } else {
#pragma unroll
          for (int j = 0; j < EV; ++j) rs[j] = 0.f;
        }
        f32x4 y[EV], y2[EV];
#pragma unroll
        for (int j = 0; j < EV; ++j) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (EPI == EPI_ACT) {
              y[j][q] = act_fwd<ACT>(v[j][q] + bias4[q] + rb4[j][q] + rs[j] * wsig4[q]);
              y2[j][q] = -wfc4[q] * act_d1<ACT>(y[j][q]);
            } else if (EPI == EPI_DACT) {
              y[j][q] = v[j][q] * act_d1<ACT>(sv[j][q]) + qv[j][q];
            } else {
              const float em = (ACT == ACT_SOFTPLUS) ? fast_exp(-sv[j][q]) : 0.f;
              y[j][q] = v[j][q] * act_d1<ACT>(sv[j][q]);
              y2[j][q] = v[j][q] * qv[j][q] * em;
            }
          }
        }
#pragma unroll
        for (int j = 0; j < EV; ++j) *reinterpret_cast<f32x4*>(a.Y + (size_t)(row0 + erow + 8 * j) * a.ldY + ecol) = y[j];
        if (EPI == EPI_CHAIN || has_y2) {
#pragma unroll
          for (int j = 0; j < EV; ++j) *reinterpret_cast<f32x4*>(a.Y2 + (size_t)(row0 + erow + 8 * j) * a.ldY2 + ecol) = y2[j];
        }
      }
      WS_STAMP(tp1);
      if (EPI != EPI_ACT) {
        // ---- request the operands of tile i (used by the epilogue of step i+1; they land while we wait at the barrier)
        const int row0 = min(tile, ntiles - 1) * WS_BM;
#pragma unroll
        for (int j = 0; j < EV; ++j) sv[j] = *reinterpret_cast<const f32x4*>(a.S + (size_t)(row0 + erow + 8 * j) * a.ldS + ecol);
        if (EPI == EPI_CHAIN) {
#pragma unroll
          for (int j = 0; j < EV; ++j) qv[j] = *reinterpret_cast<const f32x4*>(a.R + (size_t)(row0 + erow + 8 * j) * a.ldR + ecol);
        } else if (has_q) {
#pragma unroll
          for (int j = 0; j < EV; ++j) qv[j] = *reinterpret_cast<const f32x4*>(a.Q + (size_t)(row0 + erow + 8 * j) * a.ldQ + ecol);
        } else {
#pragma unroll
          for (int j = 0; j < EV; ++j) qv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
      // X(tile i+1) -> the other activation buffer (its previous contents were consumed in step i-1)
#pragma unroll
      for (int u = 0; u < XV; ++u) *reinterpret_cast<f32x4*>(&Xs[(i + 1) & 1][xoff_l[u]]) = xv[u];
#ifdef ARDAE_STAMPS
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const unsigned long long tp2 = __builtin_amdgcn_s_memtime();
      acc_a += tp1 - ts0; acc_b += tp2 - tp1;
#endif
    }
    WS_STAMP(ts2);
    __syncthreads();
#ifdef ARDAE_STAMPS
    const unsigned long long ts3 = __builtin_amdgcn_s_memtime();
    acc_c += ts3 - ts2; acc_d += ts3 - ts0;
#endif
  }
#ifdef ARDAE_STAMPS
  if (a.tile_loss != nullptr && lane == 0) {
    unsigned long long* o = reinterpret_cast<unsigned long long*>(a.tile_loss) + ((size_t)blockIdx.x * 12 + wave) * 4;
    o[0] = acc_a; o[1] = acc_b; o[2] = acc_c; o[3] = acc_d;
  }
#endif
}

// =====================================================================================================================
// WS2: both MFMA operands through LDS.  The first specialised kernel above streams its weight fragments from L2 inside
// the MFMA wave; those loads queue behind the producers' HBM streams in the CU's vector-memory pipeline (measured: MFMA
// loop 32k cycles per 32-row tile vs 16.4k ideal with 14 loads in flight).  Here the MFMA waves touch LDS only:
//   * tile 64 rows x 256 cols, K panels of 32; activation panel [64][36] and weight panel [256][36] in a 2-slot LDS ring;
//   * 8 producer waves fetch panel g+3 into registers while panel g is multiplied (two register sets = two steps of
//     latency tolerance), write panel g+1 into the free slot, and run the previous tile's epilogue from the staging
//     buffer in slices of 2 float4 per thread per step, its operands requested one step ahead;
//   * 4 consumer waves (64 rows x 64 cols each, 64 accumulators): ds_read_b128 fragments + MFMA, nothing else;
//   * one barrier per panel step.
constexpr int W2_BM = 64, W2_LDA = 36, W2_LDC = 260;
constexpr int W2_A_SLOT = W2_BM * W2_LDA;
constexpr int W2_B_SLOT = 256 * W2_LDA;
constexpr int W2_SMEM = 2 * W2_A_SLOT + 2 * W2_B_SLOT + W2_BM * W2_LDC;   // 39,680 floats = 158,720 B

// Roles (12 waves): 0-3 consumers; 4-5 load the EVEN panels, 6-7 the ODD panels; 8-11 run the epilogue.  vmcnt retires
// in order and hipcc falls back to vmcnt(0) whenever it cannot count, so every role keeps exactly ONE kind of request in
// flight: a loader pair owns every second panel (requested two steps before it is written to LDS, nothing else
// outstanding when it waits), the epilogue waves only ever wait for their own operand loads of the previous step.
template <int EPI, int ACT>
__global__ __launch_bounds__(WS_THREADS, 3) void linear_ws2_kernel(const LinArgs a, int ntiles) {
  __shared__ float smem[W2_SMEM];
  float* const As = smem;
  float* const Bs = smem + 2 * W2_A_SLOT;
  float* const Cs = Bs + 2 * W2_B_SLOT;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int role = wave < 4 ? 0 : (wave < 6 ? 1 : (wave < 8 ? 2 : 3));   // 0 consumer, 1 loader(even), 2 loader(odd), 3 epilogue
  const int K = a.src[0].K, ld = a.src[0].ld;
  const float* __restrict__ X = a.src[0].x;
  const float* __restrict__ wp = a.src[0].wp;
  const int nch = K >> 3;                       // 8-deep chunks in K
  const int P = K >> 5;                         // panel steps per tile (even, >= 6: checked by the launcher)
  const int nblk_total = (a.Nout + 31) >> 5;
  const int bid = blockIdx.x, grid = gridDim.x;
  const int T = (ntiles - bid + grid - 1) / grid;
  const int G = T * P;

  // ---------------------------------------------------------------- loader state (ltid = 0..127 within its pair)
  const int ltid = tid & 127;
  constexpr int LA = 4, LB = 16;                // float4 per loader thread: activation panel 512, weight panel 2048
  // float4 index q = ltid + 128 u.  Activation panel: row q>>3 = (ltid>>3) + 16u, float4 ltid&7.  Weight panel: lane q&63 =
  // ltid&63, chunk (q>>6)&3 = (ltid>>6) + 2(u&1), column block q>>8 = u>>1  -> every per-u term is a compile-time constant
  // or wave-uniform, so only two base offsets live in registers.
  const int ln = ltid & 63;
  const int a_gbase = (ltid >> 3) * ld + ((ltid & 7) << 2);
  const int a_lbase = (ltid >> 3) * W2_LDA + ((ltid & 7) << 2);
  const int b_gbase = ((ltid >> 6) * 64 + ln) * 4;
  const int b_lbase = (ln & 31) * W2_LDA + (ltid >> 6) * 8 + 4 * (ln >> 5);
  f32x4 ra[LA], rb[LB];
  auto issue = [&](int g) {
    g = min(g, G - 1);                                                 // past the end: re-request the last panel (no branch)
    const int i = g / P, p = g - i * P;
    const float* xt = X + (size_t)(bid + i * grid) * W2_BM * ld + p * 32 + a_gbase;
#pragma unroll
    for (int u = 0; u < LA; ++u) ra[u] = *reinterpret_cast<const f32x4*>(xt + (size_t)(16 * u) * ld);
    const float* wt = wp + (size_t)p * 1024 + b_gbase;                 // 4 chunks x 256 floats per panel
#pragma unroll
    for (int u = 0; u < LB; ++u) {
      const int nb = min(u >> 1, nblk_total - 1);
      rb[u] = *reinterpret_cast<const f32x4*>(wt + ((size_t)nb * nch + 2 * (u & 1)) * 256);
    }
  };
  auto stash = [&](int slot) {
    float* ad = &As[slot * W2_A_SLOT + a_lbase];
#pragma unroll
    for (int u = 0; u < LA; ++u) *reinterpret_cast<f32x4*>(ad + 16 * u * W2_LDA) = ra[u];
    float* bd = &Bs[slot * W2_B_SLOT + b_lbase];
#pragma unroll
    for (int u = 0; u < LB; ++u) *reinterpret_cast<f32x4*>(bd + (u >> 1) * 32 * W2_LDA + 2 * (u & 1) * 8) = rb[u];
  };

  // ---------------------------------------------------------------- epilogue state (etid = 0..255)
  // item j (0..15) of a tile = row (etid>>6) + 4j, columns ecol..ecol+3; PER items per step in steps 1..ceil(16/PER)
  const int etid = tid - 512;
  constexpr int PER = 3;
  const int ecol = min((etid & 63) << 2, a.Nout - 4);
  const int erow = etid >> 6;
  f32x4 sv[PER], qv[PER];
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f}, wsig4 = bias4, wfc4 = bias4;
  const bool has_rb = EPI == EPI_ACT && a.rowbias != nullptr, has_rs = EPI == EPI_ACT && a.rowscale != nullptr;
  const bool has_y2 = EPI == EPI_ACT && a.Y2 != nullptr, has_q = EPI == EPI_DACT && a.Q != nullptr;
  if (role == 3 && EPI == EPI_ACT) {
    if (a.bias) bias4 = *reinterpret_cast<const f32x4*>(a.bias + ecol);
    if (a.rowscale_w) wsig4 = *reinterpret_cast<const f32x4*>(a.rowscale_w + ecol);
    if (has_y2) wfc4 = *reinterpret_cast<const f32x4*>(a.R + ecol);
  }
  auto operands = [&](int row0, int j0) {                              // request S and Q/R of items j0 .. j0+PER-1 (clamped to 15)
    if (EPI == EPI_ACT) return;
#pragma unroll
    for (int jj = 0; jj < PER; ++jj) {
      const size_t row = (size_t)(row0 + erow + 4 * min(j0 + jj, 15));
      sv[jj] = *reinterpret_cast<const f32x4*>(a.S + row * a.ldS + ecol);
      if (EPI == EPI_CHAIN) qv[jj] = *reinterpret_cast<const f32x4*>(a.R + row * a.ldR + ecol);
      else if (has_q) qv[jj] = *reinterpret_cast<const f32x4*>(a.Q + row * a.ldQ + ecol);
      else qv[jj] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto finish = [&](int row0, int j0) {                                // epilogue of items j0 .. min(j0+PER,16)-1 (operands in sv/qv)
#pragma unroll
    for (int jj = 0; jj < PER; ++jj) {
      const int j = j0 + jj;
      if (j < 16) {
        const int r = erow + 4 * j;
        const size_t row = (size_t)(row0 + r);
        const f32x4 v = *reinterpret_cast<const f32x4*>(&Cs[r * W2_LDC + ecol]);
        f32x4 y, y2;
        if (EPI == EPI_ACT) {
          f32x4 pre = v + bias4;
          if (has_rb) pre += *reinterpret_cast<const f32x4*>(a.rowbias + (size_t)((row0 + r) / a.rows_per_group) * a.rowbias_ld + ecol);
          if (has_rs) pre += a.rowscale[row0 + r] * wsig4;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            y[q] = act_fwd<ACT>(pre[q]);
            y2[q] = -wfc4[q] * act_d1<ACT>(y[q]);
          }
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float d = act_d1<ACT>(sv[jj][q]);
            if (EPI == EPI_DACT) {
              y[q] = v[q] * d + qv[jj][q];
            } else {
              const float em = (ACT == ACT_SOFTPLUS) ? fast_exp(-sv[jj][q]) : 0.f;
              y[q] = v[q] * d;
              y2[q] = v[q] * qv[jj][q] * em;
            }
          }
        }
        *reinterpret_cast<f32x4*>(a.Y + row * a.ldY + ecol) = y;
        if (EPI == EPI_CHAIN || has_y2) *reinterpret_cast<f32x4*>(a.Y2 + row * a.ldY2 + ecol) = y2;
      }
    }
  };
  constexpr int ESTEPS = (16 + PER - 1) / PER;                         // 6 epilogue steps (p = 1..6) per tile
  auto epilogue_step = [&](int g) {
    const int i = g / P, p = g - i * P;
    if (i < 1) return;
    const int prow0 = (bid + (i - 1) * grid) * W2_BM;                  // previous tile of this workgroup
    if (p >= 1 && p <= ESTEPS) finish(prow0, PER * (p - 1));
    if (p < ESTEPS) operands(prow0, PER * p);                          // for step p+1
  };

  // ---------------------------------------------------------------- consumer state
  const int l31 = lane & 31, hh = lane >> 5;
  const bool cons_active = role == 0 && wave * 2 < nblk_total;
  f32x16 acc[2][2];
  auto dump = [&]() {                                                  // accumulators -> staging
    float* cb = &Cs[(4 * hh) * W2_LDC + wave * 64 + l31];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) cb[(i * 32 + (r & 3) + 8 * (r >> 2)) * W2_LDC + j * 32] = acc[i][j][r];
  };
  auto consumer_step = [&](int g) {
    if (!cons_active) return;
    const int i = g / P, p = g - i * P;
    if (p == 0) {
      if (i >= 1) dump();
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;
    }
    const float* ap = &As[(g & 1) * W2_A_SLOT + l31 * W2_LDA + hh * 4];
    const float* bp = &Bs[(g & 1) * W2_B_SLOT + (wave * 64 + l31) * W2_LDA + hh * 4];
    f32x4 af[2][2], bf[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      af[0][x] = *reinterpret_cast<const f32x4*>(ap + x * 32 * W2_LDA);
      bf[0][x] = *reinterpret_cast<const f32x4*>(bp + x * 32 * W2_LDA);
    }
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      if (kc < 3) {
#pragma unroll
        for (int x = 0; x < 2; ++x) {
          af[(kc + 1) & 1][x] = *reinterpret_cast<const f32x4*>(ap + x * 32 * W2_LDA + (kc + 1) * 8);
          bf[(kc + 1) & 1][x] = *reinterpret_cast<const f32x4*>(bp + x * 32 * W2_LDA + (kc + 1) * 8);
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
          for (int y = 0; y < 2; ++y)
            acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kc & 1][x][q], bf[kc & 1][y][q], acc[x][y], 0, 0, 0);
    }
  };

  // ---------------------------------------------------------------- pipeline
  // One loop per role (same number of barriers in each): the roles' register live ranges stay disjoint, so the kernel's
  // allocation is the maximum over the roles, not their union.
#ifdef ARDAE_STAMPS
  unsigned long long w_work = 0, w_wait = 0, w_x0 = 0, w_x1 = 0;
#define W2_BEGIN() const unsigned long long t0_ = __builtin_amdgcn_s_memtime()
#define W2_MID() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long t1_ = __builtin_amdgcn_s_memtime()
#define W2_END() { const unsigned long long t2_ = __builtin_amdgcn_s_memtime(); w_work += t1_ - t0_; w_wait += t2_ - t1_; }
#else
#define W2_BEGIN()
#define W2_MID()
#define W2_END()
#endif
  if (role == 0) {
    __syncthreads();
    for (int g = 0; g < G; ++g) {
      W2_BEGIN();
      consumer_step(g);
      W2_MID();
      __syncthreads();
      W2_END();
    }
    if (cons_active) dump();                    // drain: the last tile's accumulators
    __syncthreads();
  } else if (role == 3) {
    __syncthreads();
    for (int g = 0; g < G; ++g) {
      W2_BEGIN();
      epilogue_step(g);
      W2_MID();
      __syncthreads();
      W2_END();
    }
    __syncthreads();
    const int row0 = (bid + (T - 1) * grid) * W2_BM;     // ... and its whole epilogue
#pragma unroll 1
    for (int j0 = 0; j0 < 16; j0 += PER) {
      operands(row0, j0);
      finish(row0, j0);
    }
  } else {
    const int par = role - 1;                   // parity of the panels this loader pair owns
    if (par == 0) {                             // panel 0 now, panel 2 in flight
      issue(0);
      stash(0);
      issue(2);
    } else {
      issue(1);                                 // panel 1 in flight
    }
    __syncthreads();
    for (int g = 0; g < G; ++g) {
      W2_BEGIN();
      if (((g + 1) & 1) == par) {               // this pair owns panel g+1: write it to its slot, request panel g+3
#ifdef ARDAE_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long ta = __builtin_amdgcn_s_memtime();
#endif
        stash((g + 1) & 1);
#ifdef ARDAE_STAMPS
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long tb = __builtin_amdgcn_s_memtime();
        w_x0 += ta - t0_; w_x1 += tb - ta;
#endif
        issue(g + 3);
      }
      W2_MID();
      __syncthreads();
      W2_END();
    }
    __syncthreads();
  }
#ifdef ARDAE_STAMPS
  if (a.tile_loss != nullptr && lane == 0) {
    unsigned long long* o = reinterpret_cast<unsigned long long*>(a.tile_loss) + ((size_t)blockIdx.x * 12 + wave) * 4;
    o[0] = w_work; o[1] = w_wait; o[2] = (unsigned long long)G; o[3] = (w_x0 << 32) | (w_x1 & 0xffffffffull);
  }
#endif
}

template <int EPI, int ACT>
int launch_ws2_t(const LinArgs& a, hipStream_t st) {
  const int ntiles = a.M / W2_BM;
  const int grid = ntiles < 256 ? ntiles : 256;
  if (g_prof_enabled) {
    char name[96];
    snprintf(name, sizeof(name), "linear_ws2_kernel<%d, %d>", EPI, ACT);
    const double K = a.src[0].K;
    double tensors = 1.0 + (a.Y2 ? 1 : 0) + ((EPI == EPI_DACT || EPI == EPI_CHAIN) ? 1 : 0) + ((EPI == EPI_CHAIN) ? 1 : 0) +
                     ((EPI == EPI_DACT && a.Q) ? 1 : 0);
    prof_begin(st, name, 2.0 * a.M * (double)a.Nout * K, 4.0 * ((double)a.M * K + tensors * a.M * (double)a.Nout + K * a.Nout));
  }
  hipLaunchKernelGGL((linear_ws2_kernel<EPI, ACT>), dim3(grid), dim3(WS_THREADS), 0, st, a, ntiles);
  prof_end(st);
  ARDAE_LAUNCH_CHECK();
  return 0;
}

template <int EPI, int ACT>
int launch_ws_t(const LinArgs& a, hipStream_t st) {
  const int ntiles = ceil_div(a.M, WS_BM);
  const int grid = ntiles < 256 ? ntiles : 256;
  if (g_prof_enabled) {
    char name[96];
    snprintf(name, sizeof(name), "linear_ws_kernel<%d, %d, %d>", EPI, ACT, (a.src[0].K & 63) == 0 ? 8 : 4);
    const double K = a.src[0].K;
    double tensors = 1.0 + (a.Y2 ? 1 : 0) + ((EPI == EPI_DACT || EPI == EPI_CHAIN) ? 1 : 0) + ((EPI == EPI_CHAIN) ? 1 : 0) +
                     ((EPI == EPI_DACT && a.Q) ? 1 : 0);
    prof_begin(st, name, 2.0 * a.M * (double)a.Nout * K, 4.0 * ((double)a.M * K + tensors * a.M * (double)a.Nout + K * a.Nout));
  }
  if ((a.src[0].K & 63) == 0)
    hipLaunchKernelGGL((linear_ws_kernel<EPI, ACT, 8>), dim3(grid), dim3(WS_THREADS), 0, st, a, ntiles);
  else
    hipLaunchKernelGGL((linear_ws_kernel<EPI, ACT, 4>), dim3(grid), dim3(WS_THREADS), 0, st, a, ntiles);
  prof_end(st);
  ARDAE_LAUNCH_CHECK();
  return 0;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

bool linear_ws_eligible(const LinArgs& a, int epi) {
  if (!(epi == EPI_ACT || epi == EPI_DACT || epi == EPI_CHAIN)) return false;
  if (a.nsrc != 1 || a.colsum != nullptr) return false;
  if (a.M < 4096) return false;                                   // per-image problems stay on the small-M geometry
  if ((a.M % WS_BM) != 0 || (a.src[0].K & 31) || a.Nout > 256 || a.Nout <= 32 || (a.Nout & 3)) return false;
  if ((a.src[0].ld & 3) || !aligned16(a.src[0].x) || (a.ldY & 3) || !aligned16(a.Y)) return false;
  if (a.Y2 && ((a.ldY2 & 3) || !aligned16(a.Y2))) return false;
  if (epi == EPI_ACT) {
    if (a.bias && !aligned16(a.bias)) return false;
    if (a.rowbias && ((a.rowbias_ld & 3) || !aligned16(a.rowbias))) return false;
    if (a.rowscale_w && !aligned16(a.rowscale_w)) return false;
    if (a.Y2 && !aligned16(a.R)) return false;
  } else {
    if ((a.ldS & 3) || !aligned16(a.S)) return false;
    if (epi == EPI_CHAIN && ((a.ldR & 3) || !aligned16(a.R))) return false;
    if (epi == EPI_DACT && a.Q && ((a.ldQ & 3) || !aligned16(a.Q))) return false;
  }
  return true;
}

bool linear_ws2_eligible(const LinArgs& a, int epi) {
  if (!linear_ws_eligible(a, epi)) return false;
  const int K = a.src[0].K;
  return (a.M % W2_BM) == 0 && (K & 63) == 0 && K >= 192;
}

int launch_linear_ws2(const LinArgs& a, int epi, hipStream_t st) {
  switch (epi) {
    case EPI_ACT:
      if (a.act == ACT_NONE) return launch_ws2_t<EPI_ACT, ACT_NONE>(a, st);
      if (a.act == ACT_RELU) return launch_ws2_t<EPI_ACT, ACT_RELU>(a, st);
      if (a.act == ACT_SOFTPLUS) return launch_ws2_t<EPI_ACT, ACT_SOFTPLUS>(a, st);
      break;
    case EPI_DACT:
      if (a.act == ACT_NONE) return launch_ws2_t<EPI_DACT, ACT_NONE>(a, st);
      if (a.act == ACT_RELU) return launch_ws2_t<EPI_DACT, ACT_RELU>(a, st);
      if (a.act == ACT_SOFTPLUS) return launch_ws2_t<EPI_DACT, ACT_SOFTPLUS>(a, st);
      break;
    case EPI_CHAIN:
      if (a.act == ACT_SOFTPLUS) return launch_ws2_t<EPI_CHAIN, ACT_SOFTPLUS>(a, st);
      break;
  }
  ARDAE_CHECK_ARG(false, "linear_ws2: unsupported epilogue/activation combination (epi=%d act=%d)", epi, a.act);
  return -1;
}

int launch_linear_ws(const LinArgs& a, int epi, hipStream_t st) {
  switch (epi) {
    case EPI_ACT:
      if (a.act == ACT_NONE) return launch_ws_t<EPI_ACT, ACT_NONE>(a, st);
      if (a.act == ACT_RELU) return launch_ws_t<EPI_ACT, ACT_RELU>(a, st);
      if (a.act == ACT_SOFTPLUS) return launch_ws_t<EPI_ACT, ACT_SOFTPLUS>(a, st);
      break;
    case EPI_DACT:
      if (a.act == ACT_NONE) return launch_ws_t<EPI_DACT, ACT_NONE>(a, st);
      if (a.act == ACT_RELU) return launch_ws_t<EPI_DACT, ACT_RELU>(a, st);
      if (a.act == ACT_SOFTPLUS) return launch_ws_t<EPI_DACT, ACT_SOFTPLUS>(a, st);
      break;
    case EPI_CHAIN:
      if (a.act == ACT_SOFTPLUS) return launch_ws_t<EPI_CHAIN, ACT_SOFTPLUS>(a, st);
      break;
  }
  ARDAE_CHECK_ARG(false, "linear_ws: unsupported epilogue/activation combination (epi=%d act=%d)", epi, a.act);
  return -1;
}

}  // namespace ardae
