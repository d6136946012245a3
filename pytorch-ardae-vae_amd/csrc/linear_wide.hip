// Dispatcher of the software-pipelined wide linear kernel (template code: linear_wide_kernel.h; the kernels are
// instantiated per epilogue / activation in linear_wide_inst_*.hip so that the build parallelises).
#include <string.h>

#include "linear_wide_kernel.h"

namespace ardae {
namespace wide {

int wide_grid(int ntiles, int ncp) {
  static const char* genv = debug_knob("ARDAE_WIDE_GRID");   // experiments: resident workgroups
  int g = genv ? atoi(genv) : 256;
  g -= g % ncp;
  return ntiles < g ? ntiles : g;
}

}  // namespace wide

namespace {

using namespace wide;

bool al16w(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// supported K layouts (the wave's weight slab, 32 NJ columns x K, must fit its 256 AGPRs): one source of K = 256 (four 64-wide
// panels) or K = 32 (one 32-wide panel) with two column blocks per wave (workgroup = 256 columns), or K = 512 (eight panels)
// with one column block per wave (workgroup = 128 columns) - the N-row layers of the reference's MLP models at h_dim 256 / 512,
// z_dim 32; anything else runs on linear_kernel
int wide_panels(const LinArgs& a, int& nch, int& nj) {
  if (a.nsrc != 1) return 0;
  if (a.src[0].K == 256) { nch = 8; nj = 2; return 4; }
  if (a.src[0].K == 32) { nch = 4; nj = 2; return 1; }
  if (a.src[0].K == 512) { nch = 8; nj = 1; return 8; }
  if (a.src[0].K == 1024) { nch = 8; nj = 1; return 16; }   // rolling slab window (instantiated for config #5's epilogues only: linear_wide_eligible)
  return 0;
}

template <int EPI, int ACT, bool F1, bool F2>
int launch_wide_nch(const LinArgs& a, hipStream_t st) {
  int nch = 0, nj = 0;
  const int np = wide_panels(a, nch, nj);
  if (np == 16) {
    if constexpr (ACT == ACT_SOFTPLUS && !F1 && (EPI == EPI_ACT || (EPI == EPI_DACT && !F2))) return launch_wide<8, 16, 1, EPI, ACT, F1, F2>(a, st);
    ARDAE_CHECK_ARG(false, "linear_wide: K = 1024 is instantiated for softplus ACT / DACT-without-Q epilogues only");
  }
  if (np == 8) return launch_wide<8, 8, 1, EPI, ACT, F1, F2>(a, st);
  if (nch == 4) return launch_wide<4, 1, 2, EPI, ACT, F1, F2>(a, st);
  return launch_wide<8, 4, 2, EPI, ACT, F1, F2>(a, st);
}

template <int EPI, int ACT>
int launch_wide_flags(const LinArgs& a, hipStream_t st) {
  if constexpr (EPI == EPI_ACT) {
    const bool y2 = a.Y2 != nullptr, rs = a.rowscale != nullptr;
    if (y2 && rs) return launch_wide_nch<EPI, ACT, true, true>(a, st);
    if (y2) return launch_wide_nch<EPI, ACT, true, false>(a, st);
    if (rs) return launch_wide_nch<EPI, ACT, false, true>(a, st);
    return launch_wide_nch<EPI, ACT, false, false>(a, st);
  } else if constexpr (EPI == EPI_DACT) {
    if (a.Q) return launch_wide_nch<EPI, ACT, true, false>(a, st);
    return launch_wide_nch<EPI, ACT, false, false>(a, st);
  } else {
    return launch_wide_nch<EPI, ACT, false, false>(a, st);
  }
}

}  // namespace

bool linear_wide_eligible(const LinArgs& a, int epi) {
  int nch = 0, nj = 0;
  if (wide_panels(a, nch, nj) == 0) return false;
  const int wgcols = 128 * nj;                                      // columns per workgroup
  if (a.M <= 0 || (a.M % WBM) || a.Nout <= 0 || (a.Nout % wgcols)) return false;
  if ((int64_t)(a.M / WBM) * (a.Nout / wgcols) < 128) return false;   // small problems: the 32 x 128 tiling fills the chip better
  for (int s = 0; s < a.nsrc; ++s) {
    if (a.src[s].K % 32 || (a.src[s].ld & 3) || !al16w(a.src[s].x)) return false;
    if ((int64_t)a.src[s].ld * a.M * 4 >= (int64_t)1 << 32) return false;   // 32-bit buffer offsets
  }
  if (epi == EPI_DAE_LOSS) return false;   // Nout = z_dim there (narrow geometry)
  if (a.src[0].K == 1024) {               // the instantiated K = 1024 epilogues: softplus forward layers (no score seed), DACT without Q
    if (a.act != ACT_SOFTPLUS) return false;
    if (!((epi == EPI_ACT && !a.Y2) || (epi == EPI_DACT && !a.Q))) return false;
  }
  // instantiated for the activations of the shipped recipes; elu / tanh / leaky_relu layers run on the generic kernel
  if (a.act != ACT_NONE && a.act != ACT_RELU && a.act != ACT_SOFTPLUS) return false;
  if (epi == EPI_CHAIN && a.act != ACT_SOFTPLUS) return false;
  if (256 % (a.Nout / wgcols)) return false;   // a workgroup keeps its column panel
  if (epi == EPI_ACT && a.rowbias) {
    // a tile must not meet two images: groups that are multiples of the tile height, or whole groups of >= one tile laid out as
    // "group tiles" (the kernel's tpg mode; per-tile column sums would count the doubly computed rows twice, so not with colsum)
    if (a.rows_per_group <= 0) return false;
    if (a.rows_per_group % WBM && (a.rows_per_group < WBM || a.M % a.rows_per_group || a.colsum)) return false;
  }
  if (epi == EPI_ACT && a.rowscale && !a.rowscale_w) return false;
  // 32-bit per-lane byte offsets in the epilogue
  const int64_t ldmax = std::max<int64_t>({a.ldY, a.ldY2, a.ldS, a.ldR, a.ldQ});
  if (ldmax * a.M * 4 >= (int64_t)1 << 32) return false;   // every epilogue tensor behind a 32-bit buffer offset
  return true;
}

// A run of consecutive row-local N-row layers as ONE layer-major launch (linear_wide_layers_kernel): K = Nout = 256, one epilogue kind,
// one activation, the same optional operands in every layer (forward: bias only; DACT: all with Q or none; CHAIN), every layer reading
// its predecessor's Y.  ARDAE_WIDE_LAYERS=0: off (A/B).
bool linear_wide_layers_eligible(const LinArgs* L, int nl, int epi) {
  static const bool on = !(debug_knob("ARDAE_WIDE_LAYERS") && atoi(debug_knob("ARDAE_WIDE_LAYERS")) == 0);
  if (!on || nl < 2 || nl > WIDE_MAXL || !(epi == EPI_ACT || epi == EPI_DACT || epi == EPI_CHAIN)) return false;
  const LinArgs& f = L[0];
  if (f.act != ACT_SOFTPLUS && !(f.act == ACT_RELU && epi != EPI_CHAIN)) return false;
  for (int l = 0; l < nl; ++l) {
    const LinArgs& a = L[l];
    if (!linear_wide_eligible(a, epi)) return false;
    if (a.M != f.M || a.Nout != 256 || a.nsrc != 1 || a.src[0].K != 256 || a.act != f.act || !a.Y) return false;
    if (l > 0 && (a.src[0].x != L[l - 1].Y || a.src[0].ld != L[l - 1].ldY)) return false;
    if (epi == EPI_ACT && (a.rowbias || a.rowscale || a.Y2)) return false;             // plain forward layers
    if (epi == EPI_DACT && ((a.Q != nullptr) != (f.Q != nullptr))) return false;
    if (epi == EPI_CHAIN && (!a.Y2 || !a.R)) return false;
    // a layer must not write what a LATER layer of the run still reads as an operand from another workgroup's rows: all operands are
    // row-local (S, Q, R rows of the tile), so only the chaining above matters
  }
  return true;
}

int launch_linear_wide_layers(const LinArgs* L, int nl, int epi, hipStream_t st) {
  ARDAE_CHECK_ARG(linear_wide_layers_eligible(L, nl, epi), "linear_wide_layers: run not eligible");
  const int act = L[0].act;
  if (epi == EPI_ACT) return act == ACT_SOFTPLUS ? launch_wide_layers<8, 4, 2, EPI_ACT, ACT_SOFTPLUS, false, false>(L, nl, st)
                                                 : launch_wide_layers<8, 4, 2, EPI_ACT, ACT_RELU, false, false>(L, nl, st);
  if (epi == EPI_CHAIN) return launch_wide_layers<8, 4, 2, EPI_CHAIN, ACT_SOFTPLUS, false, false>(L, nl, st);
  if (L[0].Q) return act == ACT_SOFTPLUS ? launch_wide_layers<8, 4, 2, EPI_DACT, ACT_SOFTPLUS, true, false>(L, nl, st)
                                         : launch_wide_layers<8, 4, 2, EPI_DACT, ACT_RELU, true, false>(L, nl, st);
  return act == ACT_SOFTPLUS ? launch_wide_layers<8, 4, 2, EPI_DACT, ACT_SOFTPLUS, false, false>(L, nl, st)
                             : launch_wide_layers<8, 4, 2, EPI_DACT, ACT_RELU, false, false>(L, nl, st);
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
