// Where does the dispatcher put workgroup i?  768 workgroups, 3 fit per CU (LDS), each records XCC / SE / CU ids.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ __launch_bounds__(256) void k(unsigned* out) {
  __shared__ float pad[12000];
  pad[threadIdx.x] = 1.f;
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  // keep the block alive a little so that all 768 are co-resident
  for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(64);
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
  if (pad[(threadIdx.x + 1) & 255] == 7.f) out[0] = 0;
}
int main() {
  unsigned* d; (void)hipMalloc(&d, 768 * 8);
  k<<<768, 256>>>(d); (void)hipDeviceSynchronize();
  std::vector<unsigned> h(768 * 2); (void)hipMemcpy(h.data(), d, 768 * 8, hipMemcpyDeviceToHost);
  for (int b = 0; b < 768; ++b) {
    const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
    const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
    if (b < 40 || (b >= 254 && b < 268) || (b >= 510 && b < 524)) printf("wg %3d: xcc %u se %u sh %u cu %2u  (hw %08x)\n", b, xcc, se, sh, cu, hw);
  }
  // phase check: do b, b+256, b+512 share a CU?
  int same = 0;
  for (int b = 0; b < 256; ++b) {
    auto key = [&](int i) { return ((h[2 * i + 1] & 0xf) << 16) | (h[2 * i] & 0xff00); };
    if (key(b) == key(b + 256) && key(b) == key(b + 512)) ++same;
  }
  printf("b, b+256, b+512 on the same CU: %d of 256\n", same);
  return 0;
}
