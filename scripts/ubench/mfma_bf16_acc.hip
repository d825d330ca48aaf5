// How v_mfma_f32_32x32x16_bf16 adds its 16 products to C: rounding or truncation, and at which width?
//   hipcc --offload-arch=gfx950 -O2 scripts/ubench/mfma_bf16_acc.hip -o /tmp/mfma_bf16_acc && /tmp/mfma_bf16_acc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k(const float* av, const float* bv, float c0, float* out) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)av[i + 8 * (threadIdx.x >> 5)]; b[i] = (__bf16)bv[i + 8 * (threadIdx.x >> 5)]; }
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = c0;
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = c[0];
}

static float run(const float (&a)[16], const float (&b)[16], float c0) {
  float *da, *db, *dout, h;
  hipMalloc(&da, 64); hipMalloc(&db, 64); hipMalloc(&dout, 4);
  hipMemcpy(da, a, 64, hipMemcpyHostToDevice); hipMemcpy(db, b, 64, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, c0, dout);
  hipMemcpy(&h, dout, 4, hipMemcpyDeviceToHost);
  hipFree(da); hipFree(db); hipFree(dout);
  return h;
}

int main() {
  for (int e = 20; e <= 30; ++e) {
    float a[16], b[16];
    for (int i = 0; i < 16; ++i) { a[i] = ldexpf(1.f, -e); b[i] = 1.f; }
    const float got = run(a, b, 1.0f);
    const double want = 1.0 + 16.0 * ldexp(1.0, -e);
    printf("C=1, sixteen products of 2^-%d: got 1+%.3e (exact 1+%.3e, fp32-rounded %.9g) -> %.9g\n", e, (double)got - 1.0,
           want - 1.0, (double)(float)want, (double)got);
  }
  for (int e = 20; e <= 28; ++e) {   // one product only
    float a[16] = {0}, b[16] = {0};
    a[0] = ldexpf(1.5f, -e); b[0] = 1.f;
    const float got = run(a, b, 1.0f);
    printf("C=1, one product 1.5*2^-%d: got 1+%.3e (exact 1+%.3e)\n", e, (double)got - 1.0, 1.5 * ldexp(1.0, -e));
  }
  {  // cancellation inside the instruction: +1 and -1 and small terms
    float a[16] = {0}, b[16] = {0};
    a[0] = 1.f; b[0] = 1.f; a[1] = -1.f; b[1] = 1.f;
    for (int i = 2; i < 16; ++i) { a[i] = ldexpf(1.f, -26); b[i] = 1.f; }
    printf("C=0: +1 -1 + fourteen 2^-26: got %.6e (exact %.6e)\n", (double)run(a, b, 0.f), 14.0 * ldexp(1.0, -26));
  }
  return 0;
}
