// Microbenchmark 3: what does a VALU instruction cost next to v_mfma_f32_32x32x2_f32?
// Each "tile" = 64 MFMAs (2 accumulators) followed by NV VALU ops of a given kind on the
// finished accumulators (independent chains), with or without interleaving.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
using f32x16 = float __attribute__((ext_vector_type(16)));
using f32x4 = float __attribute__((ext_vector_type(4)));

template <int NV, int KIND>
__global__ __launch_bounds__(256, 1) void k(const f32x4* __restrict__ in, float* __restrict__ out, int iters,
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
  float s[8];
  for (int i = 0; i < 8; ++i) s[i] = (float)i;
  const unsigned long long c0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int t = 0; t < 2; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][c], b[t][4 * q + c], acc[t], 0, 0, 0);
    // NV VALU ops in 8 independent chains
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float x = acc[i & 1][(i >> 1) & 15];
      if (KIND == 0) s[i & 7] = fmaf(x, b[0][i & 31], s[i & 7]);        // v_fma_f32
      else if (KIND == 1) s[i & 7] = fmaxf(s[i & 7], x);               // v_max_f32
      else s[i & 7] = (x > s[i & 7]) ? x : s[(i + 1) & 7];             // v_cmp + v_cndmask
    }
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  float r = 0.f;
  for (int i = 0; i < 8; ++i) r += s[i];
  for (int j = 0; j < 2; ++j) for (int g = 0; g < 16; ++g) r += acc[j][g];
  out[tid] = r;
  if (tid == 0) clk[0] = c1 - c0;
}

template <int NV, int KIND>
void run(const f32x4* in, float* out, unsigned long long* clk, const char* name) {
  for (int wps = 1; wps <= 2; ++wps) {
    const int blocks = 256 * wps, iters = 2000;
    k<NV, KIND><<<blocks, 256>>>(in, out, 10, clk);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<NV, KIND><<<blocks, 256>>>(in, out, iters, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long hc; hipMemcpy(&hc, clk, 8, hipMemcpyDeviceToHost);
    printf("%-10s NV=%3d  %d waves/SIMD: %.3f ms  %.1f TFLOP/s(mfma) | wave0 %.0f cyc/tile (64 MFMA = 4096)\n", name, NV, wps,
           ms, (double)iters * 64 * blocks * 4 * 4096.0 / ms / 1e9, (double)hc / iters);
  }
}

int main() {
  const int n = 1 << 12;
  std::vector<float> h(n * 4);
  for (auto& x : h) x = (float)rand() / RAND_MAX * 2.f - 1.f;
  f32x4* in; float* out; unsigned long long* clk;
  hipMalloc(&in, n * 16); hipMalloc(&out, 1 << 24); hipMalloc(&clk, 16);
  hipMemcpy(in, h.data(), n * 16, hipMemcpyHostToDevice);
  run<0, 0>(in, out, clk, "none");
  run<32, 0>(in, out, clk, "fma");
  run<128, 0>(in, out, clk, "fma");
  run<256, 0>(in, out, clk, "fma");
  run<128, 1>(in, out, clk, "max");
  run<128, 2>(in, out, clk, "cmp+sel");
  return 0;
}
