// Which SIMD does each wave of an 8-wave workgroup land on?  (HW_ID[5:4] = SIMD_ID on gfx9)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k(int* out) {
  const int hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = hw;
}
int main() {
  int* d; hipMalloc(&d, 4 * 8 * 4);
  k<<<4, 512>>>(d);
  int h[32]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int b = 0; b < 4; ++b) {
    printf("block %d:", b);
    for (int w = 0; w < 8; ++w) printf(" w%d[simd %d wave %d cu %d raw %08x]", w, (h[b * 8 + w] >> 4) & 3, h[b * 8 + w] & 15, (h[b * 8 + w] >> 8) & 15, h[b*8+w]);
    printf("\n");
  }
  return 0;
}
