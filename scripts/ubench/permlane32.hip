// What v_permlane32_swap does with (old, src) on gfx950: prints both results for lane-id inputs.
//   hipcc --offload-arch=gfx950 -O2 scripts/ubench/permlane32.hip -o /tmp/permlane32 && /tmp/permlane32
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  const unsigned a = 100u + threadIdx.x, b = 200u + threadIdx.x;
  const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[threadIdx.x] = r[0];
  out[64 + threadIdx.x] = r[1];
  const float x = (float)threadIdx.x;
  const auto s = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, x), false, false);
  out[128 + threadIdx.x] = (unsigned)(__builtin_bit_cast(float, s[0]) + __builtin_bit_cast(float, s[1]));
}
int main() {
  unsigned* d; unsigned h[192];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int part = 0; part < 3; ++part) {
    printf("%s:", part == 0 ? "r[0] (old=100+lane)" : (part == 1 ? "r[1] (src=200+lane)" : "sum of both with old=src=lane"));
    for (int i = 0; i < 64; i += 8) printf(" [%d]=%u", i, h[part * 64 + i]);
    printf(" [33]=%u [63]=%u\n", h[part * 64 + 33], h[part * 64 + 63]);
  }
  return 0;
}
