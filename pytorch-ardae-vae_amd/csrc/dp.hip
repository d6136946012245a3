// Data-parallel gradient exchange inside the C ABI (SURVEY 8(b) `dp_allreduce_flat`, 8(e), kernel-inventory row K11): one RCCL
// communicator per rank, created once, and ONE in-place mean all-reduce per flat gradient buffer (2.17 MB cDAE, 3.36 MB model at config
// #2 - latency-bound messages over xGMI), issued on the caller's stream like every other entry point, so it can be CAPTURED into the
// step's HIP graph beside the kernels (RCCL launches its kernel into a capturing stream).  The reference has no collectives
// (single process, single device); what the mean over ranks has to reproduce is its batch-mean losses
// (ivae_ardae.py:771,804: `cdae_loss.backward()`, `model_loss.backward()` over the whole batch).
//
// RCCL is bound at RUN time (dlopen), not at link time: the library must load on a box without RCCL and in a process that already holds a
// copy (PyTorch ships its own librccl.so; one RCCL per process is enough, so a resident copy is reused before anything is loaded).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "ardae_hip.h"
#include "common.h"

namespace ardae {
namespace {

struct Rccl {
  void* handle = nullptr;
  char where[256] = "";
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;
  decltype(&ncclCommUserRank) CommUserRank = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;
  bool ok = false;
};

const Rccl& rccl() {
  static const Rccl r = [] {
    Rccl q;
    // a copy that is already resident (PyTorch's, loaded under the name its own libraries ask for) first; then the system's
    const char* resident[] = {"librccl.so", "librccl.so.1"};
    for (const char* n : resident)
      if (!q.handle) q.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    const char* fresh[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : fresh)
      if (!q.handle) q.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!q.handle) return q;
#define ARDAE_RCCL_SYM(field, sym) q.field = reinterpret_cast<decltype(q.field)>(dlsym(q.handle, #sym))
    ARDAE_RCCL_SYM(GetUniqueId, ncclGetUniqueId);
    ARDAE_RCCL_SYM(CommInitRank, ncclCommInitRank);
    ARDAE_RCCL_SYM(CommDestroy, ncclCommDestroy);
    ARDAE_RCCL_SYM(CommCount, ncclCommCount);
    ARDAE_RCCL_SYM(CommUserRank, ncclCommUserRank);
    ARDAE_RCCL_SYM(AllReduce, ncclAllReduce);
    ARDAE_RCCL_SYM(GetErrorString, ncclGetErrorString);
    ARDAE_RCCL_SYM(GetVersion, ncclGetVersion);
#undef ARDAE_RCCL_SYM
    q.ok = q.GetUniqueId && q.CommInitRank && q.CommDestroy && q.CommCount && q.CommUserRank && q.AllReduce && q.GetErrorString && q.GetVersion;
    Dl_info info;
    int ver = 0;
    if (q.ok && q.GetVersion(&ver) == ncclSuccess && dladdr(reinterpret_cast<void*>(q.AllReduce), &info) && info.dli_fname)
      snprintf(q.where, sizeof(q.where), "RCCL %d.%d.%d (%s)", ver / 10000, (ver / 100) % 100, ver % 100, info.dli_fname);
    return q;
  }();
  return r;
}

struct DpComm {
  uint32_t magic;
  ncclComm_t comm;
  int nranks, rank, device;
};
constexpr uint32_t DP_MAGIC = 0x41524443u;      // "ARDC": a stale or foreign handle is refused, not dereferenced further

#define ARDAE_RCCL(call)                                                                                                   \
  do {                                                                                                                     \
    ncclResult_t r__ = (call);                                                                                             \
    if (r__ != ncclSuccess) {                                                                                              \
      ::ardae::set_last_error("%s failed: %s (%s:%d)", #call, rccl().GetErrorString(r__), __FILE__, __LINE__);            \
      return 1000 + (int)r__;                                                                                              \
    }                                                                                                                      \
  } while (0)

int need_rccl(const char* who) {
  ARDAE_CHECK_ARG(rccl().ok, "%s: no usable librccl.so in this process or on the library path (dlopen: %s)", who, rccl().handle ? "symbols missing" : "not found");
  return 0;
}

}  // namespace
}  // namespace ardae

using namespace ardae;

extern "C" {

const char* ardae_dp_backend(void) { return rccl().ok ? rccl().where : ""; }

int ardae_dp_unique_id(void* host_id) {
  ARDAE_CHECK_ARG(host_id != nullptr, "ardae_dp_unique_id: host_id is NULL");
  ARDAE_TRY(need_rccl("ardae_dp_unique_id"));
  static_assert(sizeof(ncclUniqueId) == ARDAE_DP_UNIQUE_ID_BYTES, "unique id size");
  ncclUniqueId id;
  ARDAE_RCCL(rccl().GetUniqueId(&id));
  memcpy(host_id, &id, sizeof(id));
  return 0;
}

int ardae_dp_comm_create(const void* host_id, int nranks, int rank, void** comm_out) {
  ARDAE_CHECK_ARG(host_id && comm_out, "ardae_dp_comm_create: NULL argument");
  ARDAE_CHECK_ARG(nranks >= 1 && rank >= 0 && rank < nranks, "ardae_dp_comm_create: rank %d of %d", rank, nranks);
  *comm_out = nullptr;
  ARDAE_TRY(need_rccl("ardae_dp_comm_create"));
  int dev = -1;
  ARDAE_HIP(hipGetDevice(&dev));
  ncclUniqueId id;
  memcpy(&id, host_id, sizeof(id));
  ncclComm_t c = nullptr;
  ARDAE_RCCL(rccl().CommInitRank(&c, nranks, id, rank));      // collective: returns when all `nranks` processes have called it
  int n = 0, r = -1;
  ARDAE_RCCL(rccl().CommCount(c, &n));
  ARDAE_RCCL(rccl().CommUserRank(c, &r));
  if (n != nranks || r != rank) {
    rccl().CommDestroy(c);
    ARDAE_CHECK_ARG(false, "ardae_dp_comm_create: the communicator reports rank %d of %d, asked for %d of %d", r, n, rank, nranks);
  }
  DpComm* d = new DpComm{DP_MAGIC, c, nranks, rank, dev};
  *comm_out = d;
  return 0;
}

int ardae_dp_comm_query(void* comm, int* nranks, int* rank, int* device) {
  DpComm* d = static_cast<DpComm*>(comm);
  ARDAE_CHECK_ARG(d && d->magic == DP_MAGIC, "ardae_dp_comm_query: not a communicator handle");
  int n = 0, r = -1;
  ARDAE_RCCL(rccl().CommCount(d->comm, &n));          // what RCCL itself reports, not what the caller asked for
  ARDAE_RCCL(rccl().CommUserRank(d->comm, &r));
  if (nranks) *nranks = n;
  if (rank) *rank = r;
  if (device) *device = d->device;
  return 0;
}

int ardae_dp_comm_destroy(void* comm) {
  DpComm* d = static_cast<DpComm*>(comm);
  ARDAE_CHECK_ARG(d && d->magic == DP_MAGIC, "ardae_dp_comm_destroy: not a communicator handle");
  d->magic = 0;
  ncclResult_t r = rccl().CommDestroy(d->comm);
  delete d;
  if (r != ncclSuccess) {
    set_last_error("ncclCommDestroy failed: %s", rccl().GetErrorString(r));
    return 1000 + (int)r;
  }
  return 0;
}

int ardae_dp_allreduce_mean(void* comm, float* buf, size_t n, void* stream) {
  DpComm* d = static_cast<DpComm*>(comm);
  ARDAE_CHECK_ARG(d && d->magic == DP_MAGIC, "ardae_dp_allreduce_mean: not a communicator handle");
  ARDAE_CHECK_ARG(buf != nullptr && n >= 1, "ardae_dp_allreduce_mean: empty buffer");
  // in place; ncclAvg = sum over ranks, then one multiplication by 1 / nranks (exact for the power-of-two rank counts of a node).
  // One rank: RCCL returns the buffer as it is (nothing to add, nothing launched).
  ARDAE_RCCL(rccl().AllReduce(buf, buf, n, ncclFloat32, ncclAvg, d->comm, (hipStream_t)stream));
  return 0;
}

}  // extern "C"
