// One dot product of 32 terms through the f16-pair sequence of the kernels (cross products of both K-steps first, leading
// products last) on v_mfma_f32_32x32x16_f16, against the exact sum -- for a case file written by bwdp_diag.py:
// 32 floats (scaled row of the image) + 32 floats (scaled direction).
//   hipcc --offload-arch=gfx950 -O2 scripts/ubench/mfma_f16_case.hip -o /tmp/mfma_f16_case && /tmp/mfma_f16_case scripts/ubench/_case/row.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// a1, a2, b1, b2: [2 K-steps][16] as floats (already exactly representable in f16)
__global__ void k(const float* a1, const float* a2, const float* b1, const float* b2, int order, float* out) {
  const int hi = threadIdx.x >> 5, col = threadIdx.x & 31;
  f16x8 A1[2], A2[2], B1[2], B2[2];
  for (int sp = 0; sp < 2; ++sp)
    for (int i = 0; i < 8; ++i) {
      const int kidx = 16 * sp + 8 * (i >> 2) + 4 * hi + (i & 3);
      A1[sp][i] = col == 0 ? (_Float16)a1[kidx] : (_Float16)0.f;   // row 0 of the tile
      A2[sp][i] = col == 0 ? (_Float16)a2[kidx] : (_Float16)0.f;
      B1[sp][i] = col == 0 ? (_Float16)b1[kidx] : (_Float16)0.f;   // sample 0
      B2[sp][i] = col == 0 ? (_Float16)b2[kidx] : (_Float16)0.f;
    }
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  if (order == 0) {
    for (int sp = 0; sp < 2; ++sp) {
      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(A2[sp], B1[sp], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1[sp], B2[sp], c, 0, 0, 0);
    }
    for (int sp = 0; sp < 2; ++sp) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1[sp], B1[sp], c, 0, 0, 0);
  } else if (order == 1) {   // leading products only
    for (int sp = 0; sp < 2; ++sp) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1[sp], B1[sp], c, 0, 0, 0);
  } else {                   // cross products only
    for (int sp = 0; sp < 2; ++sp) {
      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(A2[sp], B1[sp], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1[sp], B2[sp], c, 0, 0, 0);
    }
  }
  if (threadIdx.x == 0) out[0] = c[0];
}

int main(int argc, char** argv) {
  if (argc < 2) return 1;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  float w[32], x[32];
  if (fread(w, 4, 32, f) != 32 || fread(x, 4, 32, f) != 32) return 3;
  fclose(f);
  float a1[32], a2[32], b1[32], b2[32];
  double lead = 0, cross = 0, drop = 0, exact = 0;
  for (int i = 0; i < 32; ++i) {
    a1[i] = (float)(_Float16)w[i]; a2[i] = (float)(_Float16)(w[i] - a1[i]);
    b1[i] = (float)(_Float16)x[i]; b2[i] = (float)(_Float16)(x[i] - b1[i]);
    lead += (double)a1[i] * b1[i]; cross += (double)a2[i] * b1[i] + (double)a1[i] * b2[i]; drop += (double)a2[i] * b2[i];
    exact += (double)w[i] * x[i];
  }
  float *d[4], *dout;
  const float* h[4] = {a1, a2, b1, b2};
  for (int j = 0; j < 4; ++j) { hipMalloc(&d[j], 128); hipMemcpy(d[j], h[j], 128, hipMemcpyHostToDevice); }
  hipMalloc(&dout, 4);
  for (int order = 0; order < 3; ++order) {
    float got;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d[0], d[1], d[2], d[3], order, dout);
    hipMemcpy(&got, dout, 4, hipMemcpyDeviceToHost);
    const double want = order == 0 ? lead + cross : (order == 1 ? lead : cross);
    printf("order %d: got %.10g  sum of its products %.10g  diff %.4g  (exact dot %.10g, dropped w2v2 %.4g)\n", order, (double)got,
           want, (double)got - want, exact, drop);
  }
  double big = 0;
  for (int i = 0; i < 32; ++i) big = fmax(big, fabs((double)a1[i] * b1[i]));
  printf("largest leading product %.6g = 2^%.2f; 2^-25 of its binade = %.4g\n", big, log2(big), ldexp(1.0, (int)floor(log2(big)) - 25));
  return 0;
}
