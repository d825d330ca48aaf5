// Microbenchmark (round 6): what does ONE wave per SIMD get out of the vector ALU on gfx950, and what do TWO get?
// Decides the layout question of the small-LMI kernel (config 4: 16 384 samples x 4 lanes = exactly one wave per SIMD):
// is a stream of v_pk_fma_f32 / v_fma_f32 / DPP moves from a single wave issue-limited below what the SIMD retires?
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/valu_issue.hip -o /tmp/valu_issue && /tmp/valu_issue
// Every kernel runs `iters` x 64 instructions of one form per wave; dependent = one chain, independent = 8 chains.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));

enum Form { PK_IND, PK_DEP, FMA_IND, FMA_DEP, DPP_IND, FMAC_DPP_IND, MIX_QUAD, PK_DEP2, FMA_DEP2, NFORMS };
static const char* kNames[NFORMS] = {"v_pk_fma_f32, 8 independent chains", "v_pk_fma_f32, ONE dependent chain",
                                     "v_fma_f32, 8 independent chains", "v_fma_f32, ONE dependent chain",
                                     "v_mov_b32_dpp quad_perm, independent", "v_fmac_f32_dpp quad_perm, 8 independent chains",
                                     "5 v_pk_fma_f32 + 2 v_mov_b32_dpp (the Householder update's mix)",
                                     "v_pk_fma_f32, TWO interleaved dependent chains", "v_fma_f32, TWO interleaved dependent chains"};

template <int FORM>
__global__ __launch_bounds__(256) void k(const float* __restrict__ in, float* __restrict__ out, int iters) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  f2 acc[8], a, b;
  float s[8], x, y;
  a = f2{in[tid & 0xffff], in[(tid + 1) & 0xffff]} * 1e-3f;
  b = f2{in[(tid + 2) & 0xffff], in[(tid + 3) & 0xffff]};
  x = a[0]; y = b[0];
#pragma unroll
  for (int i = 0; i < 8; ++i) { acc[i] = f2{in[(tid + 4 + i) & 0xffff], 0.f}; s[i] = acc[i][0]; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if constexpr (FORM == PK_IND) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
      } else if constexpr (FORM == PK_DEP) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[0]) : "v"(a), "v"(b));
      } else if constexpr (FORM == PK_DEP2) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i & 1]) : "v"(a), "v"(b));
      } else if constexpr (FORM == FMA_IND) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s[i]) : "v"(x), "v"(y));
      } else if constexpr (FORM == FMA_DEP) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s[0]) : "v"(x), "v"(y));
      } else if constexpr (FORM == FMA_DEP2) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s[i & 1]) : "v"(x), "v"(y));
      } else if constexpr (FORM == DPP_IND) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
          asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf" : "=v"(s[i]) : "v"(s[(i + 1) & 7]));
      } else if constexpr (FORM == FMAC_DPP_IND) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
          asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf" : "+v"(s[i]) : "v"(x), "v"(y));
      } else {  // MIX_QUAD: 8 instructions = 6 pk_fma + 2 dpp (close to the sweep's 5 : 2)
#pragma unroll
        for (int i = 0; i < 6; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
        asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf" : "=v"(s[6]) : "v"(s[0]));
        asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf" : "=v"(s[7]) : "v"(s[1]));
      }
    }
  }
  float r = x + y;
#pragma unroll
  for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][1] + s[i];
  out[tid] = r;
}

template <int FORM>
void run_form(const float* in, float* out, hipEvent_t e0, hipEvent_t e1) {
  for (int wps = 1; wps <= 4; wps *= 2) {
    const int blocks = 256 * wps, iters = 40000;
    k<FORM><<<blocks, 256>>>(in, out, 200);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<FORM><<<blocks, 256>>>(in, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double per_wave = (double)iters * 64, ns_per_instr_per_simd = ms * 1e6 / (per_wave * wps);
    printf("%-66s %d wave(s)/SIMD: %7.3f ms   %.3f ns per instruction per SIMD = %.2f cycles at 2.4 GHz\n", kNames[FORM], wps, ms,
           ns_per_instr_per_simd, ns_per_instr_per_simd * 2.4);
  }
}

int main() {
  const int n = 1 << 16;
  std::vector<float> h(n);
  for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  float *in, *out;
  hipMalloc(&in, n * 4);
  hipMalloc(&out, 1 << 24);
  hipMemcpy(in, h.data(), n * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  run_form<PK_IND>(in, out, e0, e1);
  run_form<PK_DEP>(in, out, e0, e1);
  run_form<PK_DEP2>(in, out, e0, e1);
  run_form<FMA_IND>(in, out, e0, e1);
  run_form<FMA_DEP>(in, out, e0, e1);
  run_form<FMA_DEP2>(in, out, e0, e1);
  run_form<DPP_IND>(in, out, e0, e1);
  run_form<FMAC_DPP_IND>(in, out, e0, e1);
  run_form<MIX_QUAD>(in, out, e0, e1);
  return 0;
}
