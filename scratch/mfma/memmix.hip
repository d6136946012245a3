// What do streaming dword loads/stores cost when they ride in the SAME wave's instruction stream as the MFMAs, versus
// in a separate VALU-free mover wave on the same SIMD?  Per iteration: 16 MFMAs and NL loads + NS stores (256 B each).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int OFF> __device__ __forceinline__ void gl(float& d, unsigned voff, const float* s) {
  asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2 offset:%3" : "=v"(d) : "v"(voff), "s"(s), "n"(OFF) : "memory");
}
template <int OFF> __device__ __forceinline__ void gs(unsigned voff, float v, float* s) {
  asm volatile("s_nop 4\n\tglobal_store_dword %0, %1, %2 offset:%3" ::"v"(voff), "v"(v), "s"(s), "n"(OFF) : "memory");
}
// MODE 0: everything in the MFMA wave (4 waves / block).  MODE 1: 4 MFMA waves + 4 mover waves (8 waves / block).
template <int MODE, int NM>   // NM memory op pairs (1 load + 1 store) per 16 MFMAs, NM in {0, 2, 4, 8}
__global__ __launch_bounds__(MODE ? 512 : 256) void k(float* out, int iters, const float* x, float* y) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const unsigned voff = lane * 4;
  const size_t wbase = ((size_t)blockIdx.x * 4 + (wave & 3)) * (size_t)iters * NM * 64;
  const float* px = x + wbase;
  float* py = y + wbase;
  if (wave < 4) {
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const float av = (float)lane, bv = 0.5f;
    float v[8];
    for (int u = 0; u < 8; ++u) v[u] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        acc[q & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[q & 3], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0x4);
        if (MODE == 0 && NM > 0) {
          // stores of the values loaded in the previous iteration, then this iteration's loads (one op behind an MFMA)
          if (q < NM) { if (q == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(0) : "memory"); gs<0>(voff, v[q & 7], py + q * 64); }
          else if (q < 2 * NM) gl<0>(v[(q - NM) & 7], voff, px + (q - NM) * 64);
        }
        __builtin_amdgcn_sched_barrier(0x4);
      }
      px += NM * 64; py += NM * 64;
    }
    float s = v[0];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  } else if (NM > 0) {
    float v[8];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < NM; ++u) gl<0>(v[u], voff, px + u * 64);
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])::"memory");
#pragma unroll
      for (int u = 0; u < NM; ++u) gs<0>(voff, v[u], py + u * 64);
      px += NM * 64; py += NM * 64;
    }
  }
}
template <int MODE, int NM>
void run(float* out, float* x, float* y) {
  const int iters = 2048;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<MODE, NM><<<256, MODE ? 512 : 256>>>(out, iters, x, y); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) k<MODE, NM><<<256, MODE ? 512 : 256>>>(out, iters, x, y);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double bytes = 256.0 * 4 * iters * NM * 256 * 2;
  printf("%s  %d load+store pairs per 16 MFMAs: %.3f ms (MFMA-only ideal %.3f ms)  %.2f TB/s\n", MODE ? "mover waves " : "same wave   ", NM, ms,
         (double)iters * 16 * 64 / 2.4e6, bytes / ms / 1e9);
}
int main() {
  float *out, *x, *y; (void)hipMalloc(&out, 1 << 24);
  const size_t n = (size_t)256 * 4 * 2048 * 8 * 64 + 4096;
  if (hipMalloc(&x, n * 4) != hipSuccess || hipMalloc(&y, n * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
  (void)hipMemset(x, 0, n * 4);
  run<0, 0>(out, x, y); run<0, 2>(out, x, y); run<0, 4>(out, x, y); run<0, 8>(out, x, y);
  run<1, 2>(out, x, y); run<1, 4>(out, x, y); run<1, 8>(out, x, y);
  return 0;
}
