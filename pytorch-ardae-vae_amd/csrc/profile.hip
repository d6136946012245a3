// See profile.h.  Events are created lazily, kept in a pool and resolved (hipEventElapsedTime) only in
// ardae_profile_report, which synchronises; nothing here runs unless ardae_profile_enable(1) was called.
#include "profile.h"

#include <map>
#include <string>
#include <vector>

#include "ardae_hip.h"

namespace ardae {

bool g_prof_enabled = false;

namespace {
struct Rec {
  std::string name;
  hipEvent_t e0, e1;
  double flops, bytes;
};
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;

hipEvent_t get_event() {
  if (!g_pool.empty()) {
    hipEvent_t e = g_pool.back();
    g_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}
}  // namespace

void prof_begin_impl(hipStream_t st, const char* name, double flops, double bytes) {
  Rec r;
  r.name = name; r.flops = flops; r.bytes = bytes;
  r.e0 = get_event(); r.e1 = get_event();
  (void)hipEventRecord(r.e0, st);
  g_recs.push_back(r);
}
void prof_end_impl(hipStream_t st) {
  if (!g_recs.empty()) (void)hipEventRecord(g_recs.back().e1, st);
}

}  // namespace ardae

using namespace ardae;

extern "C" {

int ardae_profile_enable(int on) {
  g_prof_enabled = on != 0;
  return 0;
}

int ardae_profile_report(ardae_profile_entry* entries, int max_entries) {
  std::map<std::string, ardae_profile_entry> agg;
  for (auto& r : g_recs) {
    (void)hipEventSynchronize(r.e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, r.e0, r.e1);
    auto& e = agg[r.name];
    if (e.calls == 0) {
      memset(&e, 0, sizeof(e));
      strncpy(e.name, r.name.c_str(), sizeof(e.name) - 1);
    }
    e.calls += 1; e.total_ms += ms; e.flops += r.flops; e.bytes += r.bytes;
    g_pool.push_back(r.e0); g_pool.push_back(r.e1);
  }
  g_recs.clear();
  int n = 0;
  for (auto& kv : agg) {
    if (n < max_entries && entries) entries[n] = kv.second;
    ++n;
  }
  return n;
}

}  // extern "C"

namespace {
__global__ void debug_stamp_kernel(unsigned long long* slots, int slot) {
  if (threadIdx.x == 0 && blockIdx.x == 0) slots[slot] = __builtin_amdgcn_s_memrealtime();
}
}  // namespace

extern "C" int ardae_debug_stamp(unsigned long long* slots, int slot, void* stream) {
  ARDAE_CHECK_ARG(slots && slot >= 0, "debug_stamp: bad arguments");
  hipLaunchKernelGGL(debug_stamp_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, slots, slot);
  ARDAE_LAUNCH_CHECK();
  return 0;
}
