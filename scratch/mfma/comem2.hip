// Memory-streaming waves with NO vector-ALU instructions (saddr addressing, SALU pointer bumps, data stored as loaded)
// beside MFMA waves on the same SIMD.  VARIANT 1 adds one v_mul per element to the copy waves.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int OFF> __device__ __forceinline__ void gl(float& d, unsigned voff, const float* s) {
  asm volatile("global_load_dword %0, %1, %2 offset:%3" : "=v"(d) : "v"(voff), "s"(s), "n"(OFF) : "memory");
}
template <int OFF> __device__ __forceinline__ void gs(unsigned voff, float v, float* s) {
  asm volatile("global_store_dword %0, %1, %2 offset:%3" ::"v"(voff), "v"(v), "s"(s), "n"(OFF) : "memory");
}
template <int VARIANT>
__global__ __launch_bounds__(512) void k(float* out, int iters, const float* x, float* y, int chunks) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (wave < 4) {
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const float av = (float)lane, bv = 0.5f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[q & 3], 0, 0, 0);
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  } else {
    // each wave streams `chunks` blocks of 8 x 256 B
    const size_t wbase = ((size_t)blockIdx.x * 4 + (wave - 4)) * (size_t)chunks * 512;
    const float* px = x + wbase;
    float* py = y + wbase;
    const unsigned voff = lane * 4;
    for (int c = 0; c < chunks; ++c) {
      float v[8];
      gl<0>(v[0], voff, px); gl<256>(v[1], voff, px); gl<512>(v[2], voff, px); gl<768>(v[3], voff, px);
      gl<1024>(v[4], voff, px); gl<1280>(v[5], voff, px); gl<1536>(v[6], voff, px); gl<1792>(v[7], voff, px);
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])::"memory");
      if (VARIANT == 1) {
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] *= 2.f;
      }
      gs<0>(voff, v[0], py); gs<256>(voff, v[1], py); gs<512>(voff, v[2], py); gs<768>(voff, v[3], py);
      gs<1024>(voff, v[4], py); gs<1280>(voff, v[5], py); gs<1536>(voff, v[6], py); gs<1792>(voff, v[7], py);
      px += 512; py += 512;
    }
  }
}
template <int VARIANT>
void run(const char* name, int iters, int chunks, float* out, float* x, float* y) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<VARIANT><<<256, 512>>>(out, iters, x, y, chunks); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) k<VARIANT><<<256, 512>>>(out, iters, x, y, chunks);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double bytes = 256.0 * 4 * chunks * 2048 * 2;
  printf("var %d %-24s mfma iters %5d bytes %5.0f MB: %.3f ms  (MFMA ideal %.3f ms, %.2f TB/s)\n", VARIANT, name, iters, bytes / 1e6, ms,
         (double)iters * 16 * 64 / 2.4e6, bytes / ms / 1e9);
}
int main() {
  float *out, *x, *y; (void)hipMalloc(&out, 1 << 24);
  const size_t n = (size_t)256 << 20;
  (void)hipMalloc(&x, n * 4); (void)hipMalloc(&y, n * 4); (void)hipMemset(x, 0, n * 4);
  run<0>("MFMA only", 2048, 0, out, x, y);
  run<0>("copy only", 0, 256, out, x, y);
  run<0>("MFMA + copy", 2048, 256, out, x, y);
  run<1>("copy only", 0, 256, out, x, y);
  run<1>("MFMA + copy", 2048, 256, out, x, y);
  return 0;
}
