// Microbenchmark: what does a filler cost between two asm MFMAs at ONE wave per SIMD (A operand in the accumulator file,
// D named literally)?   hipcc --offload-arch=gfx950 -O3 scripts/ubench/ws_slot.hip -o scripts/ubench/ws_slot && scripts/ubench/ws_slot
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define MFMA(R, a, b) asm volatile("v_mfma_f32_32x32x16_f16 v[%c2:%c3], %0, %1, v[%c2:%c3]" : : "a"(a), "v"(b), "i"(192 + 16 * (R)), "i"(207 + 16 * (R)) : "v255")
#define MFMAV(R, a, b) asm volatile("v_mfma_f32_32x32x16_f16 v[%c2:%c3], %0, %1, v[%c2:%c3]" : : "v"(a), "v"(b), "i"(192 + 16 * (R)), "i"(207 + 16 * (R)) : "v255")

template <int V>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(192))) void slot_kernel(const f16x8* __restrict__ src, float* __restrict__ out, unsigned* __restrict__ cycles, int iters) {
  const int lane = threadIdx.x & 63;
  f16x8 A[8], Av[8], B[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=a"(A[i]) : "v"(lane * 16 + i * 1024), "s"(src));
    B[i] = src[64 * (8 + i) + lane];
    Av[i] = src[64 * (16 + i) + lane];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float k0 = 0.f, k1 = 0.f, k2 = 0.f, k3 = 0.f, k4 = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 24; ++s) {
      // accumulators alternate like the kernel's: set 0 (R = 0, 1) is written, set 1 (R = 2, 3) is read by the filler
      if constexpr (V == 4) MFMAV(s & 1, Av[s & 7], B[(s >> 1) & 7]);
      else MFMA(s & 1, A[s & 7], B[(s >> 1) & 7]);
      if constexpr (V == 1 || V == 3) asm volatile("s_nop 0\n\ts_nop 0\n\ts_nop 0");
      if constexpr (V == 2 || V == 3) asm volatile("v_max3_f32 %0, %0, v[%c1], v[%c2]\n\tv_max3_f32 %0, %0, v[%c3], v[%c4]" : "+v"(k0) : "i"(224 + (s & 7) * 4), "i"(225 + (s & 7) * 4), "i"(226 + (s & 7) * 4), "i"(227 + (s & 7) * 4));
      if constexpr (V == 5) {   // five independent single-issue fillers
        asm volatile("v_max_f32 %0, %0, v[%c5]\n\tv_max_f32 %1, %1, v[%c5]\n\tv_max_f32 %2, %2, v[%c5]\n\tv_max_f32 %3, %3, v[%c5]\n\tv_max_f32 %4, %4, v[%c5]"
                     : "+v"(k0), "+v"(k1), "+v"(k2), "+v"(k3), "+v"(k4) : "i"(224 + s));
      }
      if constexpr (V == 6) {   // two independent v_max3
        asm volatile("v_max3_f32 %0, %0, v[%c2], v[%c3]\n\tv_max3_f32 %1, %1, v[%c4], v[%c5]" : "+v"(k0), "+v"(k1) : "i"(224 + (s & 7) * 4), "i"(225 + (s & 7) * 4), "i"(226 + (s & 7) * 4), "i"(227 + (s & 7) * 4));
      }
      if constexpr (V == 7) {   // four v_fma on two chains (the sums of squares)
        asm volatile("v_fma_f32 %0, v[%c2], v[%c2], %0\n\tv_fma_f32 %1, v[%c3], v[%c3], %1\n\tv_fma_f32 %0, v[%c4], v[%c4], %0\n\tv_fma_f32 %1, v[%c5], v[%c5], %1"
                     : "+v"(k0), "+v"(k1) : "i"(224 + (s & 7) * 4), "i"(225 + (s & 7) * 4), "i"(226 + (s & 7) * 4), "i"(227 + (s & 7) * 4));
      }
      if constexpr (V == 8) asm volatile("s_nop 0");
      if constexpr (V == 9) {   // two v_max3 on VGPRs the MFMAs do not touch
        asm volatile("v_max3_f32 %0, %0, %2, %3\n\tv_max3_f32 %1, %1, %3, %2" : "+v"(k0), "+v"(k1) : "v"(k2), "v"(k3));
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_nop 15\n\ts_nop 15");
  float r;
  asm volatile("v_mov_b32 %0, v[192]" : "=v"(r));
  out[blockIdx.x * 256 + threadIdx.x] = r + k0 + k1 + k2 + k3 + k4;
  if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = (unsigned)(t1 - t0);
}

template <int V>
void run(const char* what, const f16x8* src, float* out, unsigned* cyc, int grid) {
  const int iters = 200;
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(slot_kernel<V>, dim3(grid), dim3(256), 0, 0, src, out, cyc, iters);
  hipDeviceSynchronize();
  unsigned c = 0;
  hipMemcpy(&c, cyc, 4, hipMemcpyDeviceToHost);
  printf("%-58s %6.1f cycles per MFMA slot (grid %d)\n", what, (double)c / (iters * 24.0), grid);
}

int main() {
  f16x8* src; float* out; unsigned* cyc;
  hipMalloc(&src, 64 * 32 * sizeof(f16x8)); hipMemset(src, 0, 64 * 32 * sizeof(f16x8));
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 4);
  for (int grid : {1, 256}) {
    run<0>("MFMA only (A in AGPR)", src, out, cyc, grid);
    run<4>("MFMA only (A in VGPR)", src, out, cyc, grid);
    run<8>("MFMA + s_nop 0", src, out, cyc, grid);
    run<1>("MFMA + 3 x s_nop 0", src, out, cyc, grid);
    run<2>("MFMA + 2 dependent v_max3 on accumulators", src, out, cyc, grid);
    run<6>("MFMA + 2 independent v_max3 on accumulators", src, out, cyc, grid);
    run<9>("MFMA + 2 independent v_max3 on other VGPRs", src, out, cyc, grid);
    run<3>("MFMA + 3 s_nop + 2 dependent v_max3", src, out, cyc, grid);
    run<5>("MFMA + 5 independent v_max", src, out, cyc, grid);
    run<7>("MFMA + 4 v_fma (two chains) on accumulators", src, out, cyc, grid);
  }
  return 0;
}
