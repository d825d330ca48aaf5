// Microbenchmark (round 6): waves that ALTERNATE between a burst of MFMAs and vector work on the burst's results -- the shape
// of every tile walk of this library -- W waves per SIMD.  Does the vector phase of one wave run under the bursts of the
// others (time -> max(MFMA, VALU)) or do the phases add up (time -> MFMA + VALU), as the counters of the W-in-LDS kernel say
// (MFMA busy 56 % + VALU active 40 % = the whole kernel)?
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_phases.hip -o scripts/ubench/mfma_phases
// MODE 0: the vector instructions read the accumulators (dependent on the burst) | 1: they read other registers (independent)
// | 2: dependent, but the first 8 of them are issued BEHIND THE NEXT burst's first MFMA (software pipelining by hand)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NM, int NV, int MODE>
__global__ void kern(const float* __restrict__ in, float* __restrict__ out, int iters, unsigned long long* clk, int stagger) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)in[(lane * 8 + i) & 1023]; b[i] = (_Float16)in[(lane * 8 + i + 512) & 1023]; }
  f32x16 c;
  for (int g = 0; g < 16; ++g) c[g] = 0.f;
  float s[8], o[16];
  for (int i = 0; i < 8; ++i) s[i] = in[lane + i];
  for (int i = 0; i < 16; ++i) o[i] = in[lane + 8 + i];
  if (stagger > 0) for (int i = 0; i < (wave >> 2) * stagger; ++i) __builtin_amdgcn_s_sleep(8);
  if (stagger == -1) {      // a static issue priority per wave of a SIMD
    if ((wave >> 2) == 0) __builtin_amdgcn_s_setprio(3);
    else if ((wave >> 2) == 1) __builtin_amdgcn_s_setprio(2);
    else if ((wave >> 2) == 2) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
  }
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (stagger == -2) __builtin_amdgcn_s_setprio(3);      // burst high, vector phase low
    if (stagger == -3) __builtin_amdgcn_s_setprio(0);      // burst low, vector phase high
#pragma unroll
    for (int u = 0; u < NM; ++u) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    if (stagger == -2) __builtin_amdgcn_s_setprio(0);
    if (stagger == -3) __builtin_amdgcn_s_setprio(3);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if constexpr (MODE == 1) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(s[i & 7]) : "v"(o[i & 15]), "v"(o[(i + 1) & 15]));
      else asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(s[i & 7]) : "v"(c[i & 15]), "v"(c[(i + 1) & 15]));
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  float r = 0.f;
  for (int i = 0; i < 8; ++i) r += s[i];
  for (int g = 0; g < 16; ++g) r += c[g];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (blockIdx.x == 0 && lane == 0) clk[wave] = c1 - c0;
}

static float* g_in; static float* g_out; static unsigned long long* g_clk;
template <int NM, int NV, int MODE>
void run(int wps, int stagger = 0) {
  const int iters = 4000;
  kern<NM, NV, MODE><<<256, 256 * wps>>>(g_in, g_out, 10, g_clk, stagger);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    kern<NM, NV, MODE><<<256, 256 * wps>>>(g_in, g_out, iters, g_clk, stagger);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  unsigned long long c[16]; hipMemcpy(c, g_clk, 128, hipMemcpyDeviceToHost);
  unsigned long long mx = 0; for (int i = 0; i < 4 * wps; ++i) mx = c[i] > mx ? c[i] : mx;
  const double per_iter = (double)mx / iters;
  printf("%2d MFMA + %3d v_max3 (%s)%s, %d waves/SIMD: %7.3f ms, %8.0f clocks per iteration of the slowest wave | pipe needs %5d, "
         "vector issue ~%5d, per SIMD\n", NM, NV, MODE == 1 ? "independent" : "on the results", stagger > 0 ? " staggered" : (stagger == -1 ? " static prio" : (stagger == -2 ? " burst prio 3" : (stagger == -3 ? " vector prio 3" : ""))), wps, best,
         per_iter, NM * 32 * wps, NV * 4 * wps);
}

int main() {
  std::vector<float> h(1 << 16);
  for (auto& x : h) x = (float)rand() / RAND_MAX * 2.f - 1.f;
  hipMalloc(&g_in, h.size() * 4); hipMalloc(&g_out, 256 * 1024 * 4); hipMalloc(&g_clk, 128);
  hipMemcpy(g_in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  for (int st = 0; st >= -3; --st) { run<12, 96, 0>(4, st); run<12, 96, 0>(3, st); run<12, 96, 0>(2, st); run<24, 192, 0>(2, st); run<12, 48, 0>(4, st); }
  return 0;
}
