// Developer micro-benchmark: where does a wave of lmi_quad_kernel<float, 5> (config 4: r = 20, n = k = 10) spend its
// cycles?  s_memtime at the kernel's phase boundaries (RAYEN_LQ_STAMPS), averaged over the waves, and the kernel time.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DRAYEN_LQ_STAMPS -I include -I rayen_amd/csrc scripts/ubench/lq_stamps.hip -o /tmp/lq_stamps
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "rayen_lmi_quad.h"

using namespace rayen;

int main(int argc, char** argv) {
  const int r = 20, R = 20, n = 10, k = 10;
  const int64_t B = argc > 1 ? atoll(argv[1]) : 16384;
  std::mt19937 gen(1);
  std::uniform_real_distribution<float> u(-1.f, 1.f);
  std::vector<float> host((size_t)n * R * R + k, 0.f);
  for (int a = 0; a < n; ++a)
    for (int i = 0; i < r; ++i)
      for (int j = 0; j <= i; ++j) {
        const float x = u(gen);
        host[((size_t)a * R + i) * R + j] = x;
        host[((size_t)a * R + j) * R + i] = x;
      }
  std::vector<float> hv((size_t)B * n);
  for (auto& x : hv) x = u(gen);
  float *image, *v, *y;
  int32_t* ids;
  hipMalloc(&image, host.size() * 4);
  hipMemcpy(image, host.data(), host.size() * 4, hipMemcpyHostToDevice);
  hipMalloc(&v, hv.size() * 4);
  hipMemcpy(v, hv.data(), hv.size() * 4, hipMemcpyHostToDevice);
  hipMalloc(&y, (size_t)B * k * 4);
  hipMalloc(&ids, 8);
  const int64_t elems = (int64_t)host.size();
  const size_t lds = 4 * (((elems + 3) & ~int64_t(3)) + 64 * (size_t)(n + 1));
  const bool mf = argc > 2 && atoi(argv[2]) != 0;      // second argument 1: S on the matrix cores (form_S_mfma)
  const int R4 = R / 4, NT = R4 * R4, ks = (n + 3) / 4;
  std::vector<float> hwm((size_t)ks * NT * 64, 0.f);
  for (int kk = 0; kk < ks; ++kk)
    for (int tau = 0; tau < NT; ++tau)
      for (int l = 0; l < 64; ++l) {
        const int mrow = l & 15, a = 4 * kk + (l >> 4);
        const int sub = lq::Mma16<float>::sub_of_row(mrow), idx = 4 * tau + lq::Mma16<float>::reg_of_row(mrow);
        const int row = 4 * (idx / R) + sub, c = idx % R;
        if (a < n && row < r && c < r) hwm[((size_t)kk * NT + tau) * 64 + l] = host[((size_t)a * R + row) * R + c];
      }
  hwm.insert(hwm.end(), host.begin() + (size_t)n * R * R, host.end());
  float* wm;
  hipMalloc(&wm, hwm.size() * 4);
  hipMemcpy(wm, hwm.data(), hwm.size() * 4, hipMemcpyHostToDevice);
  const int64_t gen_elems = (int64_t)n * R * R;
  const int64_t elems_mf = (int64_t)hwm.size();
  const size_t lds_mf = 4 * (((elems_mf + 3) & ~int64_t(3)) + ((64 * (size_t)(n + 1) + 3) & ~size_t(3)) + 4 * (size_t)(lq::quad_stage_tiles<float, 5>() * 64 * 4));
  const unsigned grid = (unsigned)((B + 63) / 64);
  auto launch = [&]() {
    if (mf)
      hipLaunchKernelGGL((lq::lmi_quad_kernel<float, 5, true>), dim3(grid), dim3(256), lds_mf, 0, wm, ids, r, n, k, 0, 1, 0,
                         elems_mf, v, B, (int64_t)n, y, (int64_t)k, (float*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr, ks);
    else
      hipLaunchKernelGGL((lq::lmi_quad_kernel<float, 5, false>), dim3(grid), dim3(256), lds, 0, image, ids, r, n, k, 0, 1, 0, elems, v, B,
                         (int64_t)n, y, (int64_t)k, (float*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr, 0);
  };
  std::vector<float> hy((size_t)B * k);
  for (int i = 0; i < 2000; ++i) launch();
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  const int reps = 2000;
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(hy.data(), y, hy.size() * 4, hipMemcpyDeviceToHost);
  double chk = 0;
  for (float x : hy) chk += x;
  printf("B=%lld  %s  %.2f us per launch (grid %u)  checksum %.9g\n", (long long)B, mf ? "S on MFMA" : "S from LDS ", ms / reps * 1e3, grid, chk);
#ifdef RAYEN_LQ_STAMPS
  std::vector<unsigned long long> st(4096 * 8);
  hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(lq::lq_stamp_buf), st.size() * 8);
  const int waves = (int)std::min<int64_t>(grid, 1024) * 4;
  double sum[8] = {0};
  unsigned long long t_first = ~0ull, t_last = 0;
  for (int w = 0; w < waves; ++w) {
    for (int s = 1; s <= 5; ++s) sum[s] += (double)(st[w * 8 + s] - st[w * 8 + s - 1]);
    t_first = std::min(t_first, st[w * 8 + 0]);
    t_last = std::max(t_last, st[w * 8 + 5]);
  }
  const char* names[6] = {"", "image + v -> LDS, barrier", "S = sum v_a G_a", "Householder sweep", "Gershgorin + Sturm multisection", "write-out"};
  double total = 0;
  for (int s = 1; s <= 5; ++s) total += sum[s] / waves;
  for (int s = 1; s <= 5; ++s) printf("  %-34s %9.0f ticks (%4.1f %%)\n", names[s], sum[s] / waves, 100.0 * sum[s] / waves / total);
  printf("  per wave %0.f ticks of s_memtime\n", total);
  (void)t_first; (void)t_last;
#endif
  return 0;
}
