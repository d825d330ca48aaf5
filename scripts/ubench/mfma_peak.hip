// Microbenchmark: what does v_mfma_f32_32x32x2_f32 sustain on this chip with random operands?
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
using f32x16 = float __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k(const float* __restrict__ in, float* __restrict__ out, int iters) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = in[(tid * 16 + i) & 0xffff]; b[i] = in[(tid * 16 + 8 + i) & 0xffff]; }
  f32x16 acc[NACC];
  for (int j = 0; j < NACC; ++j) for (int g = 0; g < 16; ++g) acc[j][g] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[(i + j) & 7], acc[j], 0, 0, 0);
  }
  float s = 0.f;
  for (int j = 0; j < NACC; ++j) for (int g = 0; g < 16; ++g) s += acc[j][g];
  out[tid] = s;
}

int main() {
  const int n = 1 << 16;
  std::vector<float> h(n);
  for (auto& x : h) x = (float)rand() / RAND_MAX * 2.f - 1.f;
  float *in, *out;
  hipMalloc(&in, n * 4); hipMalloc(&out, 1 << 24);
  hipMemcpy(in, h.data(), n * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int zero = 0; zero < 2; ++zero) {
    if (zero) hipMemset(in, 0, n * 4);
    for (int nacc = 1; nacc <= 4; nacc *= 2)
    for (int wps = 1; wps <= 4; ++wps) {
      const int blocks = 256 * wps;  // 4 waves per block -> wps waves per SIMD
      const int iters = 20000;
      auto run = [&](int it) {
        if (nacc == 1) k<1><<<blocks, 256>>>(in, out, it);
        else if (nacc == 2) k<2><<<blocks, 256>>>(in, out, it);
        else k<4><<<blocks, 256>>>(in, out, it);
      };
      run(100);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      run(iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double mfma_per_wave = (double)iters * 8 * nacc;
      printf("chains/wave %d  ", nacc);
      const double flops = mfma_per_wave * blocks * 4 * (2.0 * 32 * 32 * 2);
      const double cyc_per_simd = mfma_per_wave * wps * 64;  // if the pipe were always busy
      printf("%s operands, %d waves/SIMD: %.3f ms  %.1f TFLOP/s  implied clock if pipe 100%% busy %.3f GHz\n",
             zero ? "zero  " : "random", wps, ms, flops / ms / 1e9, cyc_per_simd / (ms * 1e-3) / 1e9);
    }
  }
  return 0;
}
