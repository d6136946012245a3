// Dispatcher of the layer-chain kernel (template code: linear_chain_kernel.h, instantiated in linear_chain_inst_*.hip).
#include "linear_chain_kernel.h"

namespace ardae {
namespace wide {

int chain_grid(int ntiles) { return ntiles < 256 ? ntiles : 256; }

}  // namespace wide

using namespace wide;

namespace {
bool al16c(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// most row tiles per workgroup for which one launch per chain beats one launch per layer (per-layer launches amortise their
// fixed cost over the tiles and keep the NEXT tile's panel loads in flight; the chain kernel pays an exposed input load per tile)
int chain_max_tiles() {
  static const int v = debug_knob("ARDAE_CHAIN_MAX_TILES") ? atoi(debug_knob("ARDAE_CHAIN_MAX_TILES")) : 1024;
  return v;
}
// ... and for the chains whose last layer stages the next tile (EPI_ACT, EPI_DACT without Q: linear_chain_kernel's PF)
int chain_pf_max_tiles() {
  static const int v = debug_knob("ARDAE_CHAIN_PF_MAX_TILES") ? atoi(debug_knob("ARDAE_CHAIN_PF_MAX_TILES")) : 1024;
  return v;
}
}  // namespace

// A chain: nl >= 2 layers of one epilogue kind, K = Nout = 256, every layer's input = the previous layer's Y, on whole 64-row
// tiles; EPI_ACT: row bias / sigma term on the first layer only, score seed (Y2) on the last only; EPI_DACT: Q on all layers or
// none; small row counts only (chain_max_tiles).  ARDAE_CHAIN_MAX_TILES=0 switches the kernel off.
bool linear_chain_eligible(const LinArgs* L, int nl, int epi) {
  if (nl < 2 || nl > CH_MAXL || !(epi == EPI_ACT || epi == EPI_DACT || epi == EPI_CHAIN)) return false;
  const int M = L[0].M;
  const bool pf = epi == EPI_ACT || (epi == EPI_DACT && L[0].Q == nullptr);
  if (M <= 0 || (M % WBM) || M / WBM < 128 || M / WBM > (pf ? std::max(chain_pf_max_tiles(), chain_max_tiles()) : chain_max_tiles())) return false;
  for (int l = 0; l < nl; ++l) {
    const LinArgs& a = L[l];
    if (!linear_wide_eligible(a, epi)) return false;
    if (a.M != M || a.Nout != 256 || a.nsrc != 1 || a.src[0].K != 256 || a.act != L[0].act || !a.Y || a.ldY < 256) return false;
    if (l > 0 && (a.src[0].x != L[l - 1].Y || a.src[0].ld != L[l - 1].ldY)) return false;
    if (a.act != ACT_SOFTPLUS && a.act != ACT_RELU) return false;
    if (epi == EPI_ACT) {
      if ((a.rowbias || a.rowscale) && l != 0) return false;
      if (a.rowbias && (a.rows_per_group <= 0 || a.rows_per_group % WBM)) return false;
      if (a.rowscale && !a.rowbias) return false;          // instantiated together (W_1 of the energy network)
      if (a.rowbias && !a.rowscale) return false;
      if (a.Y2 && l != nl - 1) return false;
      if (a.Y2 && !L[0].rowbias) return false;             // the seed output exists on the energy stack only
      if (a.colsum) return false;
    } else if (epi == EPI_DACT) {
      if ((a.Q != nullptr) != (L[0].Q != nullptr) || a.colsum) return false;
    } else {
      if (a.act != ACT_SOFTPLUS || !a.Y2 || !a.R) return false;
      if (a.colsum && l != nl - 1) return false;
    }
    if (!al16c(a.src[0].wp)) return false;
  }
  return true;
}

int launch_linear_chain(const LinArgs* L, int nl, int epi, hipStream_t st) {
  ARDAE_CHECK_ARG(linear_chain_eligible(L, nl, epi), "linear_chain: shape not eligible");
  ChainArgs ca;
  memset(&ca, 0, sizeof(ca));
  ca.nl = nl; ca.M = L[0].M;
  for (int l = 0; l < nl; ++l) ca.L[l] = L[l];
  const int act = L[0].act;
  if (epi == EPI_ACT) {
    const bool f2 = L[0].rowbias != nullptr, f1 = L[nl - 1].Y2 != nullptr;
    if (act == ACT_SOFTPLUS) {
      if (f2 && f1) return launch_chain<EPI_ACT, ACT_SOFTPLUS, true, true, false>(ca, st);
      if (f2) return launch_chain<EPI_ACT, ACT_SOFTPLUS, true, false, false>(ca, st);
      return launch_chain<EPI_ACT, ACT_SOFTPLUS, false, false, false>(ca, st);
    }
    if (f2 && f1) return launch_chain<EPI_ACT, ACT_RELU, true, true, false>(ca, st);
    if (f2) return launch_chain<EPI_ACT, ACT_RELU, true, false, false>(ca, st);
    return launch_chain<EPI_ACT, ACT_RELU, false, false, false>(ca, st);
  }
  if (epi == EPI_DACT) {
    const bool fq = L[0].Q != nullptr;
    if (act == ACT_SOFTPLUS) return fq ? launch_chain<EPI_DACT, ACT_SOFTPLUS, false, false, true>(ca, st) : launch_chain<EPI_DACT, ACT_SOFTPLUS, false, false, false>(ca, st);
    return fq ? launch_chain<EPI_DACT, ACT_RELU, false, false, true>(ca, st) : launch_chain<EPI_DACT, ACT_RELU, false, false, false>(ca, st);
  }
  return launch_chain<EPI_CHAIN, ACT_SOFTPLUS, false, false, false>(ca, st);
}

}  // namespace ardae
