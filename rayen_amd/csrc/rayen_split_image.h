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

// One K-step of the f16-pair walks' tile burst (n_pad = 64 instances of rayen_mfma_pair.hip / rayen_mfma_pair_io.hip) as
// ONE statement with its control flow INSIDE: items that use only half of a tile (rayen_tiles.h) skip the other half's
// K-steps and wait with other counts.  To the compiler each statement is a single definition of the chunk registers and
// of the accumulators whatever path runs inside.  Written as C++ branches around plain statements and MFMA builtins,
// hipcc (a) gave a chunk different registers on different paths and copied chunks that were still in flight
// (scripts/check_split_asm.py), and (b) gave the accumulators a second register set, 32 v_mov and an exposed s_nop 10
// behind every guarded pair of MFMAs (0.082 ms on config 3 where the unguarded stream takes 0.058).
// `ctrl` (wave-uniform, one SGPR per item, pair_item_ctrl below): bit sp = K-step sp belongs to the item; bit 4 + sp = its
// chunks have only the tight count of younger operations behind them; bit 8 = the item's first K-step is 2 (the accumulators
// start afresh there).  A full tile behind a full tile takes no branch at all:
//   pair_kstep1<NT, SP, T, R>: [not the item's: nothing] vmcnt(R) [tight: vmcnt(T)]
//                              [first K-step: acc = a2 b1 | else acc += a2 b1] acc += a1 b2;  a2 <- 1 KiB at base
//   pair_kstep2<NT, SP, LAST>: [not the item's: nothing] acc += a1 b1; a1 <- 1 KiB at base
//                              [LAST: the 12 wait states between an 8-pass MFMA's result and its first reader that is not
//                              an accumulating MFMA -- hipcc pads nothing for instructions inside a statement]
// (Two sample tiles: the chains of acc[0] and acc[1] alternate, as hipcc schedules the builtins.)
#define RAYEN_MFMA16 "v_mfma_f32_32x32x16_f16 "
#define RAYEN_PAIR_RELOAD_TEXT(reg) RAYEN_ASM_BASE_COPY "global_load_dwordx4 " reg ", %[off], " RAYEN_ASM_BASE "\n"
__device__ __forceinline__ int pair_item_ctrl(const int shape, const bool after_half_b) {
  // MS_FULL: K-steps 0..3, tight unless it follows a second half (then K-steps 0,1 are relaxed: their chunks were re-loaded
  // by the FIRST half, a whole item earlier); MS_HALF_A: K-steps 0,1, tight; MS_HALF_B: K-steps 2,3, relaxed, fresh start
  int ctrl = shape == 1 ? 0x33 : (shape == 2 ? 0x10C : (after_half_b ? 0xCF : 0xFF));
  return __builtin_amdgcn_readfirstlane(ctrl);
}
// (the plain re-load of a chunk: n_pad = 32 instances, where every item is a full tile)
__device__ __forceinline__ void pair_reload(u32x4& chunk, const char* base, const unsigned off) {
  uint64_t asm_base;
  asm volatile(RAYEN_ASM_BASE_COPY "global_load_dwordx4 %[d], %[off], " RAYEN_ASM_BASE "" : [d] "+v"(chunk), [b] "=&s"(asm_base) : [off] "v"(off), [base] "s"(base));
}
template <int NT, int SP, int T, int R>
__device__ __forceinline__ void pair_kstep1(u32x4& a1, u32x4& a2, f32x16 (&acc)[NT], const f16x8 (&b1)[NT], const f16x8 (&b2)[NT],
                                            const int ctrl, const char* base, const unsigned off) {
  uint64_t asm_base;
  // SP = 0: always a first K-step (C = 0); SP = 2: first iff bit 8; SP = 1, 3: never
#define RAYEN_K1_HEAD "s_bitcmp1_b32 %[ctrl], %[sp]\n\ts_cbranch_scc0 9f\n\ts_waitcnt vmcnt(%[R])\n\ts_bitcmp1_b32 %[ctrl], %[tb]\n\ts_cbranch_scc0 1f\n\ts_waitcnt vmcnt(%[T])\n1:\n\t"
  if constexpr (NT == 2) {
    if constexpr (SP == 0) {
      // (the accumulators are OUTPUTS here: an item's first statement.  As read-write operands they would be live from one
      // item into the next and across the group boundary -- 32 registers hipcc then spills, and the reload's vmcnt(0),
      // which it places inside the item loop, drains the A stream and the trickled rows on every item.  A second half
      // skips this statement and starts them at K-step 2.)
      asm volatile(RAYEN_K1_HEAD
                   RAYEN_MFMA16 "%[c0], %[a2], %[b10], 0\n\t" RAYEN_MFMA16 "%[c1], %[a2], %[b11], 0\n\t"
                   RAYEN_MFMA16 "%[c0], %[a1], %[b20], %[c0]\n\t" RAYEN_MFMA16 "%[c1], %[a1], %[b21], %[c1]\n\t"
                   RAYEN_PAIR_RELOAD_TEXT("%[a2]") "9:"
                   : [a1] "+v"(a1), [a2] "+v"(a2), [c0] "=&v"(acc[0]), [c1] "=&v"(acc[1]), [b] "=&s"(asm_base)
                   : [b10] "v"(b1[0]), [b11] "v"(b1[1]), [b20] "v"(b2[0]), [b21] "v"(b2[1]), [ctrl] "s"(ctrl), [base] "s"(base),
                     [off] "v"(off), [T] "n"(T), [R] "n"(R), [sp] "n"(SP), [tb] "n"(4 + SP)
                   : "scc");
    } else if constexpr (SP == 2) {
      asm volatile(RAYEN_K1_HEAD
                   "s_bitcmp1_b32 %[ctrl], 8\n\ts_cbranch_scc1 3f\n\t"
                   RAYEN_MFMA16 "%[c0], %[a2], %[b10], %[c0]\n\t" RAYEN_MFMA16 "%[c1], %[a2], %[b11], %[c1]\n4:\n\t"
                   RAYEN_MFMA16 "%[c0], %[a1], %[b20], %[c0]\n\t" RAYEN_MFMA16 "%[c1], %[a1], %[b21], %[c1]\n\t"
                   RAYEN_PAIR_RELOAD_TEXT("%[a2]") "\ts_branch 9f\n3:\n\t"
                   RAYEN_MFMA16 "%[c0], %[a2], %[b10], 0\n\t" RAYEN_MFMA16 "%[c1], %[a2], %[b11], 0\n\ts_branch 4b\n9:"
                   : [a1] "+v"(a1), [a2] "+v"(a2), [c0] "+v"(acc[0]), [c1] "+v"(acc[1]), [b] "=&s"(asm_base)
                   : [b10] "v"(b1[0]), [b11] "v"(b1[1]), [b20] "v"(b2[0]), [b21] "v"(b2[1]), [ctrl] "s"(ctrl), [base] "s"(base),
                     [off] "v"(off), [T] "n"(T), [R] "n"(R), [sp] "n"(SP), [tb] "n"(4 + SP)
                   : "scc");
    } else {
      asm volatile(RAYEN_K1_HEAD
                   RAYEN_MFMA16 "%[c0], %[a2], %[b10], %[c0]\n\t" RAYEN_MFMA16 "%[c1], %[a2], %[b11], %[c1]\n\t"
                   RAYEN_MFMA16 "%[c0], %[a1], %[b20], %[c0]\n\t" RAYEN_MFMA16 "%[c1], %[a1], %[b21], %[c1]\n\t"
                   RAYEN_PAIR_RELOAD_TEXT("%[a2]") "9:"
                   : [a1] "+v"(a1), [a2] "+v"(a2), [c0] "+v"(acc[0]), [c1] "+v"(acc[1]), [b] "=&s"(asm_base)
                   : [b10] "v"(b1[0]), [b11] "v"(b1[1]), [b20] "v"(b2[0]), [b21] "v"(b2[1]), [ctrl] "s"(ctrl), [base] "s"(base),
                     [off] "v"(off), [T] "n"(T), [R] "n"(R), [sp] "n"(SP), [tb] "n"(4 + SP)
                   : "scc");
    }
  } else {
    if constexpr (SP == 0) {
      asm volatile(RAYEN_K1_HEAD
                   RAYEN_MFMA16 "%[c0], %[a2], %[b10], 0\n\t" RAYEN_MFMA16 "%[c0], %[a1], %[b20], %[c0]\n\t"
                   RAYEN_PAIR_RELOAD_TEXT("%[a2]") "9:"
                   : [a1] "+v"(a1), [a2] "+v"(a2), [c0] "=&v"(acc[0]), [b] "=&s"(asm_base)
                   : [b10] "v"(b1[0]), [b20] "v"(b2[0]), [ctrl] "s"(ctrl), [base] "s"(base), [off] "v"(off), [T] "n"(T), [R] "n"(R),
                     [sp] "n"(SP), [tb] "n"(4 + SP)
                   : "scc");
    } else if constexpr (SP == 2) {
      asm volatile(RAYEN_K1_HEAD
                   "s_bitcmp1_b32 %[ctrl], 8\n\ts_cbranch_scc1 3f\n\t"
                   RAYEN_MFMA16 "%[c0], %[a2], %[b10], %[c0]\n4:\n\t" RAYEN_MFMA16 "%[c0], %[a1], %[b20], %[c0]\n\t"
                   RAYEN_PAIR_RELOAD_TEXT("%[a2]") "\ts_branch 9f\n3:\n\t"
                   RAYEN_MFMA16 "%[c0], %[a2], %[b10], 0\n\ts_branch 4b\n9:"
                   : [a1] "+v"(a1), [a2] "+v"(a2), [c0] "+v"(acc[0]), [b] "=&s"(asm_base)
                   : [b10] "v"(b1[0]), [b20] "v"(b2[0]), [ctrl] "s"(ctrl), [base] "s"(base), [off] "v"(off), [T] "n"(T), [R] "n"(R),
                     [sp] "n"(SP), [tb] "n"(4 + SP)
                   : "scc");
    } else {
      asm volatile(RAYEN_K1_HEAD
                   RAYEN_MFMA16 "%[c0], %[a2], %[b10], %[c0]\n\t" RAYEN_MFMA16 "%[c0], %[a1], %[b20], %[c0]\n\t"
                   RAYEN_PAIR_RELOAD_TEXT("%[a2]") "9:"
                   : [a1] "+v"(a1), [a2] "+v"(a2), [c0] "+v"(acc[0]), [b] "=&s"(asm_base)
                   : [b10] "v"(b1[0]), [b20] "v"(b2[0]), [ctrl] "s"(ctrl), [base] "s"(base), [off] "v"(off), [T] "n"(T), [R] "n"(R),
                     [sp] "n"(SP), [tb] "n"(4 + SP)
                   : "scc");
    }
  }
#undef RAYEN_K1_HEAD
}
template <int NT, int SP, bool LAST>
__device__ __forceinline__ void pair_kstep2(u32x4& a1, f32x16 (&acc)[NT], const f16x8 (&b1)[NT], const int ctrl, const char* base,
                                            const unsigned off) {
  uint64_t asm_base;
  if constexpr (NT == 2) {
    asm volatile("s_bitcmp1_b32 %[ctrl], %[sp]\n\ts_cbranch_scc0 9f\n\t"
                 RAYEN_MFMA16 "%[c0], %[a1], %[b10], %[c0]\n\t" RAYEN_MFMA16 "%[c1], %[a1], %[b11], %[c1]\n\t"
                 RAYEN_PAIR_RELOAD_TEXT("%[a1]") "9:\n\ts_nop %[pad]"
                 : [a1] "+v"(a1), [c0] "+v"(acc[0]), [c1] "+v"(acc[1]), [b] "=&s"(asm_base)
                 : [b10] "v"(b1[0]), [b11] "v"(b1[1]), [ctrl] "s"(ctrl), [base] "s"(base), [off] "v"(off), [sp] "n"(SP),
                   [pad] "n"(LAST ? 11 : 0)
                 : "scc");
  } else {
    asm volatile("s_bitcmp1_b32 %[ctrl], %[sp]\n\ts_cbranch_scc0 9f\n\t"
                 RAYEN_MFMA16 "%[c0], %[a1], %[b10], %[c0]\n\t"
                 RAYEN_PAIR_RELOAD_TEXT("%[a1]") "9:\n\ts_nop %[pad]"
                 : [a1] "+v"(a1), [c0] "+v"(acc[0]), [b] "=&s"(asm_base)
                 : [b10] "v"(b1[0]), [ctrl] "s"(ctrl), [base] "s"(base), [off] "v"(off), [sp] "n"(SP), [pad] "n"(LAST ? 11 : 0)
                 : "scc");
  }
}

// 2^13 / 2^floor(log2 m) as a float and its inverse, from the biased exponent of m (clamped to [14, 254])
__device__ __forceinline__ void pow2_scale(const float m, float& scale, float& inv, int& exp_scale) {
  unsigned e = __builtin_bit_cast(unsigned, m) >> 23;
  e = e < 14u ? 14u : (e > 254u ? 254u : e);
  scale = __builtin_bit_cast(float, (267u - e) << 23);
  inv = __builtin_bit_cast(float, (e - 13u) << 23);
  exp_scale = 140 - (int)e;
}

// ---- the group boundary of the f16-pair walks in the instructions it needs (round 6).  By the cycle accounting of DESIGN.md
// 4.0 a SIMD's vector work and its MFMAs are serial, and 1 300 of the 2 000 vector instructions a wave spends per 64-sample
// group were boundary code: hipcc turned `x = v sv; p1 = (f16) x; p2 = (f16)(x - p1)` into 3.5 instructions per value
// (v_mul, half a v_cvt_pk_f16_f32, two v_fma_mix*) and `(float) p1 + (float) p2` into three (two conversions and an add).
// The mixed-precision FMA does each in ONE: f16(v sv + 0), f16(v sv - p1) (the product by a power of two is exact, so the
// fused forms round the same values: same bits), and fl32(p1 * 1 + p2).
// pair_split_lo / _hi: p1, p2 of v * sv into the LOW / HIGH half of the registers w1, w2 (low first: it defines the register).
__device__ __forceinline__ void pair_split_lo(unsigned& w1, unsigned& w2, const float v, const float sv) {
  asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(w1) : "v"(v), "v"(sv));
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(w2) : "v"(v), "v"(sv), "v"(w1));
}
__device__ __forceinline__ void pair_split_hi(unsigned& w1, unsigned& w2, const float v, const float sv) {
  asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(w1) : "v"(v), "v"(sv));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(w2) : "v"(v), "v"(sv), "v"(w1));
}
// fl32(p1 + p2) of the LOW / HIGH halves of w1, w2 (`one` = 1.0f in a register)
__device__ __forceinline__ float pair_rebuild_lo(const unsigned w1, const unsigned w2, const float one) {
  float r;
  asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(w1), "v"(one), "v"(w2));
  return r;
}
__device__ __forceinline__ float pair_rebuild_hi(const unsigned w1, const unsigned w2, const float one) {
  float r;
  asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(w1), "v"(one), "v"(w2));
  return r;
}
// NaN / Inf anywhere in a row, from its p1 pieces (eight f16 per u32x4): z <- p1 * 0 + z per register pair -- a finite piece
// leaves z alone, an infinite or NaN piece makes it NaN and it stays NaN.  y = y0 + v / max(1, kappa) is NaN exactly when a
// component of v is (an infinite one: inf * 0 from the step), so ONE test of z per row replaces a compare per output value.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void pair_nan_fold(f16x2& z, const unsigned w1) {
  asm("v_pk_fma_f16 %0, %1, 0, %0" : "+v"(z) : "v"(w1));
}
__device__ __forceinline__ bool pair_nan_seen(const f16x2 z) { return (z[0] != z[0]) || (z[1] != z[1]); }

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
  bool has_halves = false; // some items read half of a shared tile (rayen_tiles.h): not for the mapped instances
  int n_tiles = 0;         // tiles of the image (rayen_mfma_pair_wl.hip copies all of them into LDS)
  bool wl_ready = false;          // the W-in-LDS kernels were promised their dynamic LDS at pack creation
  bool wl_mapped_ready = false;   // ... and their mapped instances (room for the widest mapper next to the image)
  int64_t bytes = 0;
  std::vector<MItem> host_items;   // the item list as uploaded (rayen_mfma_pair_ws8.hip deals it out to eight waves)
};

}  // namespace rayen
