// Microbenchmark (round 6): does the second wave of a SIMD get vector / scalar / LDS instructions issued WHILE the first
// wave's v_mfma_f32_32x32x16_f16 stream occupies the matrix pipe -- i.e. is `MFMA cycles + other cycles` (serial) or
// `max(...)` (concurrent) the budget of a SIMD on gfx950?  Decides what the headline walk (rayen_mfma_pair_io.hip) can
// gain from dephasing its two waves per SIMD / from a third wave.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_coissue.hip -o scripts/ubench/mfma_coissue
// Workgroups of 8 waves, one per CU; waves 0-3 take role A, waves 4-7 role B (wave w and w + 4 share SIMD w & 3: checked
// through HW_ID and printed).  Roles: 0 nothing, 1 MFMA stream (2 accumulator chains), 2 v_fma_f32 (8 chains),
// 3 s_add_u32 chain, 4 ds_read_b128 (8 in flight), 5 MFMA with K independent v_fma_f32 behind every MFMA (same wave),
// 6 v_pk_fma_f32 (8 chains), 7 global_load_dwordx4 of a cached 1 KiB (8 in flight)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int ROLE, int K>
__device__ __forceinline__ float run_role(const float* in, const int iters, float* lds) {
  const int lane = threadIdx.x & 63;
  float r = 0.f;
  if constexpr (ROLE == 1 || ROLE == 5) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)in[(lane * 8 + i) & 1023]; b[i] = (_Float16)in[(lane * 8 + i + 512) & 1023]; }
    f32x16 c0, c1;
    for (int g = 0; g < 16; ++g) { c0[g] = 0.f; c1[g] = 0.f; }
    float s[8];
    for (int i = 0; i < 8; ++i) s[i] = in[lane + i];
    const float x = in[lane + 9] * 1e-3f, y = in[lane + 10];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 12; ++u) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
        if constexpr (ROLE == 5) {
#pragma unroll
          for (int i = 0; i < K; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s[i & 7]) : "v"(x), "v"(y));
        }
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b));
        if constexpr (ROLE == 5) {
#pragma unroll
          for (int i = 0; i < K; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s[i & 7]) : "v"(x), "v"(y));
        }
      }
    }
    for (int g = 0; g < 16; ++g) r += c0[g] + c1[g];
    for (int i = 0; i < 8; ++i) r += s[i];
  } else if constexpr (ROLE == 2) {
    float s[8];
    for (int i = 0; i < 8; ++i) s[i] = in[lane + i];
    const float x = in[lane + 9] * 1e-3f, y = in[lane + 10];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s[i]) : "v"(x), "v"(y));
    }
    for (int i = 0; i < 8; ++i) r += s[i];
  } else if constexpr (ROLE == 6) {
    f32x2 s[8];
    for (int i = 0; i < 8; ++i) s[i] = f32x2{in[lane + i], in[lane + i + 8]};
    const f32x2 x = f32x2{in[lane + 9], in[lane + 3]} * 1e-3f, y = f32x2{in[lane + 10], in[lane + 11]};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(s[i]) : "v"(x), "v"(y));
    }
    for (int i = 0; i < 8; ++i) r += s[i][0] + s[i][1];
  } else if constexpr (ROLE == 3) {
    unsigned a = (unsigned)iters, b = 3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 64; ++u) asm volatile("s_add_u32 %0, %0, %1" : "+s"(a) : "s"(b) : "scc");
    }
    r = (float)a;
  } else if constexpr (ROLE == 4) {
    const f32x4* p = reinterpret_cast<const f32x4*>(lds) + lane;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
      f32x4 t[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t[i]) : "v"((unsigned)(uintptr_t)p), "n"(i * 1024));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 8; ++i) acc += t[i];
    }
    r = acc[0] + acc[1] + acc[2] + acc[3];
  } else if constexpr (ROLE == 7) {
    const f32x4* p = reinterpret_cast<const f32x4*>(in) + lane + 128;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
      f32x4 t[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(t[i]) : "v"(p), "n"(i * 512 - 2048));
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 8; ++i) acc += t[i];
    }
    r = acc[0] + acc[1] + acc[2] + acc[3];
  } else if constexpr (ROLE >= 10 && ROLE < 30) {
    // one vector instruction form, 8 independent chains of it (64 per iteration)
    float s[8];
    for (int i = 0; i < 8; ++i) s[i] = in[lane + i];
    float x = in[lane + 9] * 1e-3f, y = in[lane + 10];
    unsigned addr = (unsigned)(uintptr_t)lds + lane * 16;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if constexpr (ROLE == 10) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(x), "v"(y));
          else if constexpr (ROLE == 11) asm volatile("v_max_f32 %0, %0, %1" : "+v"(s[i]) : "v"(x));
          else if constexpr (ROLE == 12) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(s[i]) : "v"(y));
          else if constexpr (ROLE == 13) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(s[i]) : "v"(x), "v"(y) : "vcc");
          else if constexpr (ROLE == 14) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(s[i]) : "v"(x), "v"(y));
          else if constexpr (ROLE == 15) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(s[i]) : "v"(x), "v"(y));
          else if constexpr (ROLE == 16) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(s[i]) : "v"(x), "v"(y));
          else if constexpr (ROLE == 17) asm volatile("v_sqrt_f32 %0, %0" : "+v"(s[i]));
          else if constexpr (ROLE == 18) asm volatile("v_add_u32 %0, %0, %1" : "+v"(s[i]) : "v"(x));
          else if constexpr (ROLE == 19) asm volatile("v_mov_b32 %0, %1" : "=v"(s[i]) : "v"(x));
          else if constexpr (ROLE == 20) asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(s[i]) : "v"(addr));
          else if constexpr (ROLE == 21) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*reinterpret_cast<f32x2*>(&s[i & 6])) : "v"(f32x2{y, y}));
          else if constexpr (ROLE == 22) asm volatile("v_accvgpr_write_b32 a0, %0" : : "v"(s[i]) : "a0");
          else if constexpr (ROLE == 23) asm volatile("v_readfirstlane_b32 s20, %0" : : "v"(s[i]) : "s20");
        }
    }
    for (int i = 0; i < 8; ++i) r += s[i];
  } else if constexpr (ROLE == 30) {
    // ds_read_b128 x 8 in flight, no arithmetic on the results
    const unsigned addr = (unsigned)(uintptr_t)lds + lane * 16;
    f32x4 t[8];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t[i]) : "v"(addr), "n"(i * 1024));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    for (int i = 0; i < 8; ++i) r += t[i][0];
  } else if constexpr (ROLE == 31) {
    const f32x4* p = reinterpret_cast<const f32x4*>(in) + lane + 128;
    f32x4 t[8];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(t[i]) : "v"(p), "n"(i * 512 - 2048));
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    for (int i = 0; i < 8; ++i) r += t[i][0];
  } else if constexpr (ROLE == 33) {
    // LDS-DMA: global_load_lds_dwordx4 x 8 (cached lines) + wait; nothing returns through the vector registers
    const unsigned off = lane * 16;
    const unsigned ldsb = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds + (threadIdx.x >> 6) * 1024);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2 offset:%3" : : "v"(off), "s"(ldsb), "s"(in), "n"(i * 512) : "memory");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    r = lds[lane];
  } else if constexpr (ROLE == 34) {
    // global_store_dwordx4 x 8 (the same 8 lines of this wave's own 4 KiB of `out`) + wait
    f32x4 t = {in[lane], in[lane + 1], in[lane + 2], in[lane + 3]};
    float* dst = in == nullptr ? nullptr : const_cast<float*>(in) + 16384 + (blockIdx.x * 8 + (threadIdx.x >> 6)) * 1024 + lane * 4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("global_store_dwordx4 %0, %1, off offset:%2" : : "v"(dst), "v"(t), "n"(i * 512 - 2048 + 2048) : "memory");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    r = t[0];
  } else if constexpr (ROLE == 35) {
    // global_load_dword x 8 (256 B per instruction) + wait
    const float* p = in + lane + 512;
    float t[8];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("global_load_dword %0, %1, off offset:%2" : "=v"(t[i]) : "v"(p), "n"(i * 256 - 1024));
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    for (int i = 0; i < 8; ++i) r += t[i];
  } else if constexpr (ROLE == 36) {
    // LDS-DMA of single words: global_load_lds_dword x 8 + wait
    const unsigned off = lane * 4;
    const unsigned ldsb = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds + (threadIdx.x >> 6) * 1024);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, %2 offset:%3" : : "v"(off), "s"(ldsb), "s"(in), "n"(i * 256) : "memory");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    r = lds[lane];
  } else if constexpr (ROLE == 32) {
    // one dependent chain of MFMAs (the W-in-LDS kernel's burst: one accumulator)
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)in[(lane * 8 + i) & 1023]; b[i] = (_Float16)in[(lane * 8 + i + 512) & 1023]; }
    f32x16 c0;
    for (int g = 0; g < 16; ++g) c0[g] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 24; ++u) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
    }
    for (int g = 0; g < 16; ++g) r += c0[g];
  }
  return r;
}

template <int RA, int RB, int K>
__global__ __launch_bounds__(512, 1) void kern(const float* __restrict__ in, float* __restrict__ out, int iters_a, int iters_b,
                                               unsigned long long* clk, unsigned* hwid) {
  __shared__ float lds[8 * 1024 * 4 / 4 + 64 * 4];
  for (int i = threadIdx.x; i < 8 * 1024 + 256; i += 512) lds[i] = in[i & 4095];
  __syncthreads();
  const int wave = threadIdx.x >> 6;
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    hwid[wave] = id;
  }
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  float r;
  if (wave < 4) r = run_role<RA, K>(in, iters_a, lds);
  else r = run_role<RB, K>(in, iters_b, lds);
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 512 + threadIdx.x] = r;
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) clk[wave] = c1 - c0;
}

static int g_blocks = 256;
static float* g_in; static float* g_out; static unsigned long long* g_clk; static unsigned* g_hw;

template <int RA, int RB, int K = 0>
void run(const char* name, int iters_a, int iters_b) {
  kern<RA, RB, K><<<g_blocks, 512>>>(g_in, g_out, 10, 10, g_clk, g_hw);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    kern<RA, RB, K><<<g_blocks, 512>>>(g_in, g_out, iters_a, iters_b, g_clk, g_hw);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  unsigned long long c[8]; hipMemcpy(c, g_clk, 64, hipMemcpyDeviceToHost);
  printf("%-74s %8.3f ms | wave 0 %10llu ticks, wave 4 %10llu ticks\n", name, best, c[0], c[4]);
}

int main(int argc, char** argv) {
  if (argc > 1) g_blocks = atoi(argv[1]);
  const bool quick = argc > 2;
  printf("workgroups: %d\n", g_blocks);
  std::vector<float> h(1 << 16);
  for (auto& x : h) x = (float)rand() / RAND_MAX * 2.f - 1.f;
  hipMalloc(&g_in, (h.size() + 16384 + 256 * 8 * 1024 + 4096) * 4); hipMalloc(&g_out, 256 * 512 * 4); hipMalloc(&g_clk, 64); hipMalloc(&g_hw, 32);
  hipMemcpy(g_in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  const int IA = 4000;            // x 24 MFMAs x 32 cycles = 3.07 M cycles
  run<1, 0>("A: MFMA stream alone (waves 0-3; 24 x 4000 MFMAs each)", IA, 0);
  unsigned hw[8]; hipMemcpy(hw, g_hw, 32, hipMemcpyDeviceToHost);
  printf("   HW_ID simd of waves 0..7:");
  for (int i = 0; i < 8; ++i) printf(" %u", (hw[i] >> 4) & 3);
  printf("  (cu:");
  for (int i = 0; i < 8; ++i) printf(" %u", (hw[i] >> 8) & 15);
  printf(")\n");
  run<1, 1>("A+A: MFMA stream on both waves of a SIMD", IA, IA);
  if (quick) return 0;
  const bool vmem_only = getenv("COISSUE_VMEM_ONLY") != nullptr;
  if (!vmem_only) {
  run<0, 2>("B: v_fma_f32 alone (waves 4-7; 64 x 12000)", 0, 12000);
  run<1, 2>("A+B: MFMA stream | v_fma_f32 on the partner wave", IA, 12000);
  run<0, 6>("B': v_pk_fma_f32 alone (64 x 8000)", 0, 8000);
  run<1, 6>("A+B': MFMA stream | v_pk_fma_f32 on the partner wave", IA, 8000);
  run<0, 3>("C: s_add_u32 chain alone (64 x 12000)", 0, 12000);
  run<1, 3>("A+C: MFMA stream | s_add_u32 chain on the partner wave", IA, 12000);
  run<0, 4>("D: ds_read_b128 x 8 + wait, alone (x 20000)", 0, 20000);
  run<1, 4>("A+D: MFMA stream | ds_read_b128 on the partner wave", IA, 20000);
  run<0, 7>("E: global_load_dwordx4 x 8 (cached) + wait, alone (x 6000)", 0, 6000);
  run<1, 7>("A+E: MFMA stream | cached global loads on the partner wave", IA, 6000);
  run<5, 0, 1>("F1: one wave, 1 v_fma_f32 behind every MFMA", IA, 0);
  run<5, 0, 2>("F2: one wave, 2 v_fma_f32 behind every MFMA", IA, 0);
  run<5, 0, 4>("F4: one wave, 4 v_fma_f32 behind every MFMA", IA, 0);
  run<5, 0, 6>("F6: one wave, 6 v_fma_f32 behind every MFMA", IA, 0);
  run<5, 0, 8>("F8: one wave, 8 v_fma_f32 behind every MFMA", IA, 0);
  run<5, 5, 4>("F4+F4: both waves, 4 v_fma_f32 behind every MFMA", IA, IA);
  run<5, 5, 8>("F8+F8: both waves, 8 v_fma_f32 behind every MFMA", IA, IA);
  run<32, 0>("G: ONE dependent MFMA chain alone (24 x 4000)", IA, 0);
  run<32, 32>("G+G: one dependent chain on both waves", IA, IA);
  }
#define CLASS(R, NAME, N) run<0, R>(NAME " alone", 0, N); run<1, R>("   A | " NAME, IA, N);
  if (!vmem_only) {
  CLASS(10, "v_max3_f32", 12000)
  CLASS(11, "v_max_f32", 12000)
  CLASS(12, "v_mul_f32", 12000)
  CLASS(13, "v_cmp_gt_f32 + v_cndmask_b32", 6000)
  CLASS(14, "v_fma_mix_f32", 12000)
  CLASS(15, "v_fma_mixlo_f16", 12000)
  CLASS(16, "v_pk_fma_f16", 12000)
  CLASS(17, "v_sqrt_f32", 6000)
  CLASS(18, "v_add_u32", 12000)
  CLASS(19, "v_mov_b32", 12000)
  CLASS(20, "ds_bpermute_b32 + wait", 1500)
  CLASS(21, "v_pk_mul_f32", 8000)
  CLASS(22, "v_accvgpr_write_b32", 12000)
  CLASS(23, "v_readfirstlane_b32", 12000)
  }
  CLASS(30, "ds_read_b128 x 8 + wait (no arithmetic)", 20000)
  CLASS(31, "global_load_dwordx4 x 8 cached + wait (no arithmetic)", 6000)
  CLASS(33, "global_load_lds_dwordx4 x 8 cached + wait (LDS-DMA)", 6000)
  CLASS(34, "global_store_dwordx4 x 8 + wait", 6000)
  CLASS(35, "global_load_dword x 8 cached + wait", 6000)
  CLASS(36, "global_load_lds_dword x 8 cached + wait (LDS-DMA)", 6000)
  return 0;
}
