// Vector types and the device image of the split-operand forward kernel (rayen_mfma_split.hip): three bf16 pieces of
// every entry of W in MFMA fragment order + the item list.
#pragma once

#include "rayen_mfma_kernel.h"

namespace rayen {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Every hand-placed (inline asm) vector-memory instruction of the split-operand kernels takes its scalar base through an
// SALU copy made INSIDE the statement.  Why: hipcc may restore the base from an SGPR spill lane (v_readlane_b32) with the
// instruction right in front of the statement; a VALU write of an SGPR needs five wait states before a VMEM instruction
// reads it, and the compiler pads such hazards only for instructions it emits itself.  The instances with ~200 spilled
// SGPRs (fused mapper x staged NA_E write-out) read a stale base that way and faulted on wild addresses -- the device
// fault that kept the fused mapper from sets with equality constraints in rounds 1 and 2.  An SALU read of a
// VALU-written SGPR is interlocked by the hardware, and an SALU-written SGPR needs no wait state before VMEM.
// scripts/check_split_asm.py scans every instance for the hazard (a VALU write of an SGPR within five instructions in
// front of an asm VMEM instruction that reads it).
#ifndef RAYEN_ASM_NO_BASE_COPY
#define RAYEN_ASM_BASE_COPY "s_mov_b64 %[b], %[base]\n\t"
#define RAYEN_ASM_BASE "%[b]"
#else   // developer A/B builds only: the hazard is back
#define RAYEN_ASM_BASE_COPY ""
#define RAYEN_ASM_BASE "%[base]"
#endif
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// 2^13 / 2^floor(log2 m) as a float and its inverse, from the biased exponent of m (clamped to [14, 254])
__device__ __forceinline__ void pow2_scale(const float m, float& scale, float& inv, int& exp_scale) {
  unsigned e = __builtin_bit_cast(unsigned, m) >> 23;
  e = e < 14u ? 14u : (e > 254u ? 254u : e);
  scale = __builtin_bit_cast(float, (267u - e) << 23);
  inv = __builtin_bit_cast(float, (e - 13u) << 23);
  exp_scale = 140 - (int)e;
}

// kappa candidate of a second-order cone from the walk's scaled quantities (f16-pair kernels): a' x^2 + b' x + c' = 0
// (rayen/constraint_module.py:392-396, 339-348), a' < 0.  The coefficients mix in the set's constants f0 = tau,
// f1 = a', so they are formed in natural units (wi = 1 / (gW f_s), vi = 1 / sv) and the root goes back to the scaled
// domain.  ONE definition for every schedule of the pair forward, with floating-point contraction off: the schedules
// must agree bit for bit, and which of `rt*rt - cr*cr`'s products hipcc fuses depends on the code around it (the
// W-stationary kernel differed from the plain one by one ulp on 1 row in 30 000 until this was shared).
__device__ __forceinline__ float pair_soc_candidate(const float a0, const float a1, const float total, const float wi,
                                                    const float vi, const float f0, const float f1, const float v_scl,
                                                    const float w_scale) {
#pragma clang fp contract(off)
  const float cr = (a0 * wi) * vi;
  const float br = (a1 * wi) * vi;
  const float rt = (__builtin_amdgcn_sqrtf(total) * wi) * vi;
  const float cp = rt * rt - cr * cr;
  const float bp = 2.f * br - 2.f * cr * f0;
  const float disc = bp * bp - 4.f * f1 * cp;
  float kc = 0.f;
  if (disc >= 0.f) {
    const float root = __builtin_amdgcn_sqrtf(disc);
    const float inv2a = 0.5f * __builtin_amdgcn_rcpf(f1);
    kc = (fmaxf((-bp - root) * inv2a, (-bp + root) * inv2a) * v_scl) * w_scale;
  }
  return kc;
}

struct SplitImage {
  void* Wb = nullptr;      // [n_tiles][NS][3][64] x 8 bf16
  MItem* items = nullptr;
  MPack* packs = nullptr;
  float* y0 = nullptr;
  int n_items = 0;
  int nkk = 0;
  int identity = 0;
  int n_simd = 1024;
  int64_t bytes = 0;
};

// rayen_mfma_pair.hip: two f16 pieces of every entry of gW W (gW a power of two), same fragment order
struct PairImage {
  void* Wh = nullptr;      // [n_tiles][NS][2][64] x 8 f16
  MItem* items = nullptr;
  MPack* packs = nullptr;
  float* y0 = nullptr;
  int n_items = 0;
  int nkk = 0;
  int identity = 0;
  int n_simd = 1024;
  float w_scale = 1.f, w_inv = 1.f;
  int aux_rows = 0;        // aux rows (phi | c, M'beta) of the whole set
  int first_out = 0;       // index of the first NA_E tile in the item list (n_items when NA_E = I)
  int64_t bytes = 0;
  std::vector<MItem> host_items;   // the item list as uploaded (rayen_mfma_pair_ws8.hip deals it out to eight waves)
};

}  // namespace rayen
