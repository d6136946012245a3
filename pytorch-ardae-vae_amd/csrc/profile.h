// Optional per-kernel timing with HIP events on the launch stream (bench.py's live roofline numbers).
// Disabled by default: when off, prof_begin/prof_end are a single predictable branch.
#pragma once
#include "common.h"

namespace ardae {

extern bool g_prof_enabled;
void prof_begin_impl(hipStream_t st, const char* name, double flops, double bytes);
void prof_end_impl(hipStream_t st);

inline void prof_begin(hipStream_t st, const char* name, double flops, double bytes) {
  if (g_prof_enabled) prof_begin_impl(st, name, flops, bytes);
}
inline void prof_end(hipStream_t st) {
  if (g_prof_enabled) prof_end_impl(st);
}

}  // namespace ardae
