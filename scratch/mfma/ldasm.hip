// Clean measurement of what a global_load_dwordx4 costs next to an MFMA stream (no address arithmetic in the loop:
// the loads use immediate offsets from one base pointer that advances by a power-of-two mask in SALU).
// PIPE 0: loads of iteration i are waited for immediately (vmcnt(0)) before its 16 MFMAs.
// PIPE 1: loads of iteration i+1 are issued before the 16 MFMAs of iteration i (second register set, vmcnt(NL)).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define GLOAD(dst, ptr, off) asm volatile("global_load_dwordx4 %0, %1, off offset:" #off : "=v"(dst) : "v"(ptr) : "memory")
template <int NL>
__device__ __forceinline__ void loads(f32x4 (&v)[4], const float* p) {
  if (NL >= 1) GLOAD(v[0], p, 0);
  if (NL >= 2) GLOAD(v[1], p, 1024);
  if (NL >= 3) GLOAD(v[2], p, 2048);
  if (NL >= 4) GLOAD(v[3], p, 3072);
}
__device__ __forceinline__ void mfma16(f32x16 (&acc)[4], const f32x4 (&v)[4]) {
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[q & 3][q >> 2], v[(q + 1) & 3][q >> 2], acc[q & 3], 0, 0, 0);
}
template <int NL, int PIPE>
__global__ __launch_bounds__(256) void k(float* out, const float* g, int iters, unsigned mask) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  f32x4 va[4], vb[4];
  for (int i = 0; i < 4; ++i) va[i] = vb[i] = f32x4{1.f, 2.f, 3.f, 4.f} * (float)(lane + i);
  const float* base = g + ((size_t)(blockIdx.x % 8) * 4 + wave) * 65536 + lane * 4;   // 256 KiB per wave region
  unsigned off = 0;
  if (PIPE) { loads<NL>(va, base); }
  for (int it = 0; it < iters; it += 2) {
    off = (off + 1024) & mask;
    if (PIPE) {
      loads<NL>(vb, base + off);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
    } else {
      loads<NL>(va, base + off);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    mfma16(acc, va);
    __builtin_amdgcn_sched_barrier(0);
    off = (off + 1024) & mask;
    if (PIPE) {
      loads<NL>(va, base + off);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
    } else {
      loads<NL>(vb, base + off);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    mfma16(acc, vb);
    __builtin_amdgcn_sched_barrier(0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0.f;
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NL, int PIPE>
void run(int blocks_per_cu, unsigned span_floats) {
  float *out, *g; (void)hipMalloc(&out, 1 << 24); (void)hipMalloc(&g, 33 * 65536 * 4); (void)hipMemset(g, 0, 33 * 65536 * 4);
  const int iters = 2048, grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<NL, PIPE><<<grid, 256>>>(out, g, iters, span_floats - 1); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) k<NL, PIPE><<<grid, 256>>>(out, g, iters, span_floats - 1);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double flop = (double)grid * 4 * iters * 16 * 4096.0;
  printf("NL %d pipe %d span %3u KiB waves/SIMD %d: %.3f ms  %.1f TFLOP/s\n", NL, PIPE, span_floats * 4 / 1024, blocks_per_cu, ms, flop / ms / 1e9);
  (void)hipFree(out); (void)hipFree(g);
}
template <int NL, int PIPE> void sweep(unsigned span) { for (int b = 1; b <= 4; ++b) run<NL, PIPE>(b, span); }
int main() {
  sweep<0, 0>(16384);
  sweep<2, 0>(1024); sweep<2, 1>(1024);     // 4 KiB per wave: L1
  sweep<2, 0>(16384); sweep<2, 1>(16384);   // 64 KiB per wave: L2
  sweep<4, 0>(16384); sweep<4, 1>(16384);
  return 0;
}
