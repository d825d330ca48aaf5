// Round 5 reproducer of the intermittent fault of the flat-row kernel's NA_E write-out ("Repetition"):
// on gfx950 a packed-fp32 VOP3P instruction whose op_sel makes the LOW result read the HIGH half of a source (hipcc's SLP
// vectoriser emits it to broadcast one scalar -- `scale[t]` -- to both halves: `v_pk_fma_f32 d, x, s, d op_sel:[0,1,0]`)
// now and then computes lanes 48-63 of the low result with the wrong half, while MFMAs are in flight on the SIMD.
//   hipcc --offload-arch=gfx950 -O2 scripts/ubench/pkfma_hazard.hip -o scripts/ubench/pkfma_hazard && scripts/ubench/pkfma_hazard
// modes: 0 v_pk_fma_f32 op_sel:[0,1,0]   1 the same behind s_nop 3   2 no op_sel (natural halves)   3 op_sel:[1,0,0] (src0)
//        4 v_pk_mul_f32 op_sel:[0,1]      5 op_sel_hi:[1,0,1] (the HIGH result reads the LOW half of src1)
// Each wave: a burst of MFMAs (or none), then 16 packed operations on fresh values, every lane checks both results and the
// first wrong one per mode is reported with what it equals.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct Report { unsigned long long bad; unsigned hist[128]; float first[8]; };
__global__ __launch_bounds__(512, 2) void k(int mode, int mfmas, int iters, Report* rep, float* sink) {
  const int lane = threadIdx.x & 63;
  f32x16 acc = {};
  f16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {1, 1, 1, 1, 1, 1, 1, 1};
  unsigned long long nbad = 0;
  for (int it = 0; it < iters; ++it) {
    for (int m = 0; m < mfmas; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const f32x2 x = {(float)(lane + 1 + r + (it & 255)), (float)(2 * lane + 3 + r)};
      const f32x2 s = {0.5f, 0.25f};
      const f32x2 c = {100.f + lane, 200.f + lane};
      f32x2 d = c, e;
#define RUN(text) asm volatile(text : [d] "+v"(d) : [x] "v"(x), [s] "v"(s))
      if (mode == 0) { RUN("v_pk_fma_f32 %[d], %[x], %[s], %[d] op_sel:[0,1,0]"); e = {fmaf(x[0], s[1], c[0]), fmaf(x[1], s[1], c[1])}; }
      else if (mode == 1) { RUN("s_nop 3\n\tv_pk_fma_f32 %[d], %[x], %[s], %[d] op_sel:[0,1,0]"); e = {fmaf(x[0], s[1], c[0]), fmaf(x[1], s[1], c[1])}; }
      else if (mode == 2) { RUN("v_pk_fma_f32 %[d], %[x], %[s], %[d]"); e = {fmaf(x[0], s[0], c[0]), fmaf(x[1], s[1], c[1])}; }
      else if (mode == 3) { RUN("v_pk_fma_f32 %[d], %[x], %[s], %[d] op_sel:[1,0,0]"); e = {fmaf(x[1], s[0], c[0]), fmaf(x[1], s[1], c[1])}; }
      else if (mode == 4) { RUN("v_pk_mul_f32 %[d], %[x], %[s] op_sel:[0,1]"); e = {x[0] * s[1], x[1] * s[1]}; }
      else { RUN("v_pk_fma_f32 %[d], %[x], %[s], %[d] op_sel_hi:[1,0,1]"); e = {fmaf(x[0], s[0], c[0]), fmaf(x[1], s[0], c[1])}; }
      const bool w0 = d[0] != e[0], w1 = d[1] != e[1];
      if (w0 || w1) {
        if (nbad == 0 && atomicAdd(&rep->hist[(w1 ? 64 : 0) + lane], 1u) == 0 && atomicCAS((unsigned*)&rep->first[7], 0u, 1u) == 0) {
          rep->first[0] = w0 ? d[0] : d[1]; rep->first[1] = w0 ? e[0] : e[1]; rep->first[2] = x[0]; rep->first[3] = x[1];
          rep->first[4] = c[0]; rep->first[5] = c[1]; rep->first[6] = (float)lane;
        } else {
          atomicAdd(&rep->hist[(w1 ? 64 : 0) + lane], 1u);
        }
        ++nbad;
      }
    }
  }
  if (nbad) atomicAdd(&rep->bad, nbad);
  sink[blockIdx.x * 512 + threadIdx.x] = acc[0];
}
int main() {
  Report* rep; float* sink;
  (void)hipMalloc(&rep, sizeof(Report)); (void)hipMalloc(&sink, 256 * 512 * 4);
  for (int mfmas = 6; mfmas >= 0; mfmas -= 6)
    for (int mode = 0; mode < 6; ++mode) {
      (void)hipMemset(rep, 0, sizeof(Report));
      hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, mfmas, 20000, rep, sink);
      (void)hipDeviceSynchronize();
      Report h;
      (void)hipMemcpy(&h, rep, sizeof(Report), hipMemcpyDeviceToHost);
      unsigned q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int i = 0; i < 128; ++i) q[(i >> 6) * 4 + ((i & 63) >> 4)] += h.hist[i];
      printf("mfmas/iter %d mode %d: %llu wrong of %.3g | low result, lanes 0-15 16-31 32-47 48-63: %u %u %u %u | high result: %u %u %u %u", mfmas, mode, h.bad,
             256.0 * 512 * 20000 * 16, q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7]);
      if (h.bad) printf(" | first: lane %g got %g want %g (x = %g, %g; c = %g, %g)", h.first[6], h.first[0], h.first[1], h.first[2], h.first[3], h.first[4], h.first[5]);
      printf("\n");
    }
  return 0;
}
