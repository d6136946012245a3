#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// mode 0: registers only; mode 1: A fragment re-read from LDS each chunk; mode 2: + B fragment from global (L2)
template <int MODE, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(float* out, const float* wp, int iters) {
  __shared__ float lds[64 * 68];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 64 * 68; i += blockDim.x) lds[i] = (float)(i % 7) * 0.01f;
  __syncthreads();
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  f32x4 av[2], bv[2];
  av[0] = av[1] = f32x4{1.f, 2.f, 3.f, 4.f} * (float)lane; bv[0] = bv[1] = f32x4{0.5f, 0.25f, 0.125f, 1.f};
  const float* ap = &lds[(lane & 31) * 68 + (lane >> 5) * 4];
  const float* bp = wp + lane * 4;
  for (int it = 0; it < iters; ++it) {
    if (MODE >= 1) { av[0] = *(const f32x4*)(ap + (it & 7) * 8); av[1] = *(const f32x4*)(ap + 32 * 68 + (it & 7) * 8); }
    if (MODE >= 2) { bv[0] = *(const f32x4*)(bp + (size_t)(it & 31) * 256); bv[1] = *(const f32x4*)(bp + (size_t)(32 + (it & 31)) * 256); }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x * 2 + y] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[x][q], bv[y][q], acc[x * 2 + y], 0, 0, 0);
  }
  float s = 0.f;
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE, int WAVES>
void run(const char* name, int blocks_per_cu) {
  float *out, *wp; hipMalloc(&out, 1 << 24); hipMalloc(&wp, 1 << 20); hipMemset(wp, 0, 1 << 20);
  const int iters = 4096, grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE, WAVES><<<grid, WAVES * 64>>>(out, wp, iters); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) k<MODE, WAVES><<<grid, WAVES * 64>>>(out, wp, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double flop = (double)grid * WAVES * iters * 16 * 4096.0;
  printf("%-34s waves/WG %d WG/CU %d: %.3f ms  %.1f TFLOP/s\n", name, WAVES, blocks_per_cu, ms, flop / ms / 1e9);
  hipFree(out); hipFree(wp);
}
int main() {
  run<0, 4>("registers only", 1);
  run<0, 4>("registers only", 2);
  run<0, 8>("registers only", 1);
  run<1, 4>("A frags from LDS", 1);
  run<2, 4>("A from LDS, B from L2", 1);
  run<2, 4>("A from LDS, B from L2", 3);
  return 0;
}
