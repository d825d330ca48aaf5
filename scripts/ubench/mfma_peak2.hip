// Microbenchmark 2: same MFMA count, but every MFMA of a 64-instruction tile uses distinct A/B
// registers (as the projection kernel does), and the kernel reports shader cycles vs wall clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
using f32x16 = float __attribute__((ext_vector_type(16)));
using f32x4 = float __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(const f32x4* __restrict__ in, float* __restrict__ out, int iters,
                                            unsigned long long* clk) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  float b[2][32];
  f32x4 a[8];
  for (int i = 0; i < 16; ++i) {
    f32x4 x = in[(tid * 16 + i) & 0xfff];
    b[0][2 * i] = x[0]; b[0][2 * i + 1] = x[1]; b[1][2 * i] = x[2]; b[1][2 * i + 1] = x[3];
  }
  for (int i = 0; i < 8; ++i) a[i] = in[(tid * 8 + i + 77) & 0xfff];
  f32x16 acc[2];
  for (int j = 0; j < 2; ++j) for (int g = 0; g < 16; ++g) acc[j][g] = 0.f;
  const unsigned long long c0 = __builtin_readcyclecounter();
  const unsigned long long w0 = wall_clock64();
  float keep = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int t = 0; t < 2; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][c], b[t][4 * q + c], acc[t], 0, 0, 0);
    if (MODE == 1) {  // a short dependent VALU epilogue per tile, then restart the accumulators
      for (int t = 0; t < 2; ++t) { float m = 0.f; for (int g = 0; g < 16; ++g) m = fmaxf(m, acc[t][g]); keep += m; }
      for (int t = 0; t < 2; ++t) for (int g = 0; g < 16; ++g) acc[t][g] = 0.f;
    }
    if (MODE == 2) {  // quadratic-form style epilogue: fma chain acc * b
      for (int t = 0; t < 2; ++t) { float m = 0.f; for (int g = 0; g < 16; ++g) m = fmaf(acc[t][g], b[t][g], m); keep += m; }
      for (int t = 0; t < 2; ++t) for (int g = 0; g < 16; ++g) acc[t][g] = 0.f;
    }
    if (MODE == 3) {  // data-dependent branch on a uniform value + epilogue (as the item switch does)
      const int sel = __builtin_amdgcn_readfirstlane(it) % 3;
      if (sel == 0) { for (int t = 0; t < 2; ++t) { float m = 0.f; for (int g = 0; g < 16; ++g) m = fmaxf(m, acc[t][g]); keep += m; } }
      else if (sel == 1) { for (int t = 0; t < 2; ++t) { float m = 0.f; for (int g = 0; g < 16; ++g) m = fmaf(acc[t][g], b[t][g], m); keep += m; } }
      else { for (int t = 0; t < 2; ++t) { float m = 0.f; for (int g = 0; g < 16; ++g) m = fmaf(acc[t][g], acc[t][g], m); keep += m; } }
      for (int t = 0; t < 2; ++t) for (int g = 0; g < 16; ++g) acc[t][g] = 0.f;
    }
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  const unsigned long long w1 = wall_clock64();
  float s = keep;
  for (int j = 0; j < 2; ++j) for (int g = 0; g < 16; ++g) s += acc[j][g];
  out[tid] = s;
  if (tid == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

int main() {
  const int n = 1 << 12;
  std::vector<float> h(n * 4);
  for (auto& x : h) x = (float)rand() / RAND_MAX * 2.f - 1.f;
  f32x4* in; float* out; unsigned long long* clk;
  hipMalloc(&in, n * 16); hipMalloc(&out, 1 << 24); hipMalloc(&clk, 16);
  hipMemcpy(in, h.data(), n * 16, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 4; ++mode)
    for (int wps = 1; wps <= 2; ++wps) {
      const int blocks = 256 * wps, iters = 4000;
      auto launch = [&](int n_it) { if (mode == 0) k<0><<<blocks, 256>>>(in, out, n_it, clk); else if (mode == 1) k<1><<<blocks, 256>>>(in, out, n_it, clk); else if (mode == 2) k<2><<<blocks, 256>>>(in, out, n_it, clk); else k<3><<<blocks, 256>>>(in, out, n_it, clk); };
      launch(10);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      launch(iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long hc[2]; hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
      const double mfma = (double)iters * 64;
      printf("mode %d (%s), %d waves/SIMD: %.3f ms  %.1f TFLOP/s | wave0: %.1f shader cyc per MFMA, shader clock %.3f GHz (wall clock 100 MHz ticks %llu)\n",
             mode, mode == 0 ? "pure MFMA" : mode == 1 ? "max epilogue" : mode == 2 ? "fma epilogue" : "switch epilogue", wps, ms, mfma * blocks * 4 * 4096.0 / ms / 1e9,
             (double)hc[0] / mfma, (double)hc[0] / ((double)hc[1] * 10.0) , hc[1]);
    }
  return 0;
}
