// Paired-half forward: the tile walk of rayen_mfma_split.hip on v_mfma_f32_32x32x16_f16 with TWO f16 pieces per
// operand -- three piece products per fp32 product instead of six.
//
// An f16 significand has 11 bits, so x = x1 + x2 carries 22 of the 24 bits of an fp32 number: the operands are
// REPRESENTED to 2^-23 relative (half an ulp of the second piece), and a product is rebuilt from x1y1, x1y2, x2y1
// (the dropped x2y2 is <= 2^-22 of it).  Unlike rounding in an fp32 FMA chain these errors do not accumulate along K:
// emulated on configs 2, 3 and 5 the row results T = W v are CLOSER to fp64 than those of an fp32 FMA chain, and on the
// GPU the worst error of y against the fp64 kernel is below the exact-fp32 kernel's.  f16 has a narrow exponent range, so both operands are scaled by powers of two (exact):
//   * W by one global factor gW that puts its largest entry into [2^13, 2^14) (entries down to 2^-17 of the largest
//     keep the full 22 bits; smaller ones are good to 2^-39 of the largest),
//   * every sample's direction by its own factor sv (largest component into [2^13, 2^14)).
// Every candidate of kappa is homogeneous of degree 1 in v and in the rows of W (rayen/constraint_module.py:351-458),
// so the whole walk runs on the scaled numbers and kappa is unscaled once per sample; only the closed form of a
// second-order cone, which mixes in constants of the set, is evaluated in natural units.
// Whether a pack is served by this kernel is measured at pack creation (rayen_abi.hip::fp32_selfcheck), like the
// bf16 triple kernel: against the fp64 lane kernel, next to the exact-fp32 MFMA kernel.
//
// Everything else -- persistent barrier-free waves, two per SIMD, 64 samples per wave, the rolling A buffer with
// hand-placed loads and counted waits, the epilogues -- is the design of rayen_mfma_split.hip; see there.
#include "rayen_split_image.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

namespace rayen {

// The module's mapper v = Wm x + b (rayen/constraint_module.py:259-263, 525) in front of the walk (NKX > 0 instances),
// the f16-pair form of rayen_mfma_split.hip's: Wm as an image of f16 pairs of gM Wm (gM a power of two chosen by
// pair_mapper_image_kernel from the weights' largest entry), x scaled per sample like v, the accumulators start at
// gM sx b, and the fp32 results -- which ARE in B-operand order -- are re-scaled and re-split in registers into the
// walk's B operands.  v reaches memory only when v_out != null (training).
struct PairMapper {
  const f16x8* img = nullptr;   // [NKK][NSX][2][64] x 8 f16, then n_pad floats of bias, then gM, 1 / gM
  int in_dim = 0;
  float* v_out = nullptr;
  int64_t ldvo = 0;
};

template <int NKK, bool TRACK, bool STAGED, int NKX, int NT = 2>
__device__ __forceinline__ void mfma_pair_fwd_body(
    const f16x8* __restrict__ Wh, const MItem* __restrict__ items, int n_items,
    const MPack* __restrict__ packs, const float* __restrict__ y0, int identity, int k, int n,
    const float* __restrict__ v, int64_t B, int64_t ldv, int vec_in, float* __restrict__ y, int64_t ldy,
    int vec_out, float* __restrict__ kappa_out, int32_t* __restrict__ active_out,
    int32_t* __restrict__ nan_flag, const float w_scale, const float w_inv, const PairMapper mp) {
  constexpr int NS = NKK * 2, NCH = NS * 2, KK = NKK * 16;   // NT = sample tiles per wave (2; 1 for small batches)
  __shared__ float aux_lds[kMfmaWaves][NT][32][32];
  __shared__ __attribute__((aligned(16))) float y0_lds[NKK * 32];
  __shared__ __attribute__((aligned(16))) float bias_lds[NKX > 0 ? NKK * 32 : 4];   // gM b, zero-padded
  constexpr int LSTR = NKK * 32 + 4;
  __shared__ __attribute__((aligned(16))) float line_lds[kMfmaWaves][32][LSTR];

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int col = lane & 31;
  const int hi = lane >> 5;
  const int64_t n_groups = (B + NT * 32 - 1) / (NT * 32);
  const int64_t wave_id = (int64_t)blockIdx.x * kMfmaWaves + wave;
  const int64_t wave_stride = (int64_t)gridDim.x * kMfmaWaves;
  bool bad = false;
  for (int i = threadIdx.x; i < NKK * 32; i += kMfmaWaves * 64) y0_lds[i] = y0[i];
  float gm = 1.f, gm_inv = 1.f;
  int gm_exp = 0;
  if constexpr (NKX > 0) {
    const float* tail = reinterpret_cast<const float*>(mp.img + (size_t)NKK * (NKX * 2) * 2 * 64);
    gm = tail[NKK * 32];
    gm_inv = tail[NKK * 32 + 1];
    gm_exp = (int)((__builtin_bit_cast(unsigned, gm) >> 23) & 255u) - 127;
    for (int i = threadIdx.x; i < NKK * 32; i += kMfmaWaves * 64) bias_lds[i] = tail[i] * gm;
  }
  __syncthreads();  // the only workgroup barrier
  float (*patch)[LSTR] = line_lds[wave];

  // ---- A operands: the rolling register buffer of rayen_mfma_split.hip, two chunks (a1, a2) per K-step
  u32x4 abuf[NCH];
  const unsigned lane_off = lane * 16;
  auto load_step_fresh = [&](const int sp) {
    const char* sb = reinterpret_cast<const char*>(Wh) + sp * 2048;
    uint64_t asm_base;
    asm volatile(RAYEN_ASM_BASE_COPY "global_load_dwordx4 %[d], %[off], " RAYEN_ASM_BASE "" : [d] "=v"(abuf[2 * sp + 0]), [b] "=&s"(asm_base) : [off] "v"(lane_off), [base] "s"(sb));
    asm volatile(RAYEN_ASM_BASE_COPY "global_load_dwordx4 %[d], %[off], " RAYEN_ASM_BASE " offset:1024" : [d] "=v"(abuf[2 * sp + 1]), [b] "=&s"(asm_base) : [off] "v"(lane_off), [base] "s"(sb));
  };
  if constexpr (NKX == 0) {
    // tile 0 for the first group; no wait: the group's row loads queue behind these and loads return in order
#pragma unroll
    for (int sp = 0; sp < NS; ++sp) load_step_fresh(sp);
  }

  const int64_t n_rounds = (n_groups + wave_stride - 1) / wave_stride;
  for (int64_t round = 0; round < n_rounds; ++round) {
  const int64_t grp = wave_id + round * wave_stride;
  if (grp >= n_groups) continue;
  const int64_t s_base = grp * (NT * 32);

  bool live[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) live[t] = (s_base + t * 32 + col) < B;
  // vb[t][piece][k-step] = 8 f16 = the B operand of one MFMA; element i = column 16 sp + 8 (i >> 2) + 4 hi + (i & 3)
  f16x8 vb[NT][2][NS];
  float v_scl[NT], v_inv[NT];  // sv and 1 / sv (powers of two; applied one after the other with gW, never multiplied
                               // together: sv spans 2^-114 .. 2^126)
  if constexpr (NKX > 0) {
    constexpr int NSX = NKX * 2;
    float xr[NT][NKX * 16];
    load_rows<NT, NKX, LSTR, true>(xr, v, ldv, mp.in_dim, vec_in & 1, s_base, B, live, patch, lane);
    float sx[NT], sx_inv[NT];
    int sx_exp[NT];
    f32x16 macc[NKK][NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float m = 0.f;
#pragma unroll
      for (int i = 0; i < NKX * 16; ++i) m = fmaxf(m, __builtin_fabsf(xr[t][i]));
      m = fmaxf(m, xhalf(m));
      pow2_scale(m, sx[t], sx_inv[t], sx_exp[t]);
    }
#pragma unroll
    for (int tp = 0; tp < NKK; ++tp)
#pragma unroll
      for (int a4 = 0; a4 < 4; ++a4) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(&bias_lds[32 * tp + 8 * a4 + 4 * hi]);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int c = 0; c < 4; ++c) macc[tp][t][4 * a4 + c] = b4[c] * sx[t];
      }
    const f16x8* mimg = mp.img + lane;
#pragma unroll
    for (int sxs = 0; sxs < NSX; ++sxs) {
      f16x8 xb[NT][2];   // the K-step's pieces of sx x: element i = column 16 sxs + 8 (i >> 2) + 4 hi + (i & 3)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float x = xr[t][8 * sxs + i] * sx[t];
          const _Float16 p1 = (_Float16)x;
          xb[t][0][i] = p1;
          xb[t][1][i] = (_Float16)(x - (float)p1);
        }
#pragma unroll
      for (int tp = 0; tp < NKK; ++tp) {
        const f16x8* ch = mimg + (size_t)((tp * NSX + sxs) * 2) * 64;
        const f16x8 a1 = ch[0], a2 = ch[64];
        // (cross products first; the accumulator starts at the bias)
#pragma unroll
        for (int t = 0; t < NT; ++t) macc[tp][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, xb[t][0], macc[tp][t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) macc[tp][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, xb[t][1], macc[tp][t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) macc[tp][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, xb[t][0], macc[tp][t], 0, 0, 0);
      }
    }
    // the rolling buffer is dead while the mapper runs (its registers hold x and the mapper's accumulators): tile 0
    // is fetched afresh once the mapper's MFMAs are issued -- the walk's counted waits cover it
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int sp = 0; sp < NS; ++sp) load_step_fresh(sp);
    __builtin_amdgcn_sched_barrier(0);
    // macc = gM sx v
    if (mp.v_out != nullptr) {
      float vr[NT][KK];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int tp = 0; tp < NKK; ++tp)
#pragma unroll
          for (int g = 0; g < 16; ++g) vr[t][16 * tp + g] = macc[tp][t][g] * gm_inv;
      (void)store_rows<NT, NKK, LSTR, true>(vr, sx_inv, nullptr, mp.v_out, mp.ldvo, n,
                                            (mp.ldvo % 4 == 0) && ((reinterpret_cast<uintptr_t>(mp.v_out) & 15) == 0),
                                            s_base, B, live, patch, lane);
    }
    // result register g = 4 a + c of row tile tp is direction element 32 tp + 8 a + 4 hi + c = element 4 (a & 1) + c
    // of K-step 2 tp + (a >> 1): re-scaled (largest component into [2^13, 2^14)) and split in place
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float m = 0.f;
#pragma unroll
      for (int tp = 0; tp < NKK; ++tp)
#pragma unroll
        for (int g = 0; g < 16; ++g) m = fmaxf(m, __builtin_fabsf(macc[tp][t][g]));
      m = fmaxf(m, xhalf(m));
      float f, f_inv;
      int f_exp;
      pow2_scale(m, f, f_inv, f_exp);
      // sv = f gM sx as a power of two (the exponents are added, the product of the floats could overflow on the way)
      int sv_exp = f_exp + gm_exp + sx_exp[t];
      sv_exp = sv_exp > 126 ? 126 : (sv_exp < -126 ? -126 : sv_exp);
      v_scl[t] = __builtin_bit_cast(float, (unsigned)(127 + sv_exp) << 23);
      v_inv[t] = __builtin_bit_cast(float, (unsigned)(127 - sv_exp) << 23);
#pragma unroll
      for (int tp = 0; tp < NKK; ++tp)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float x = macc[tp][t][4 * a + c] * f;
            const _Float16 p1 = (_Float16)x;
            vb[t][0][2 * tp + (a >> 1)][4 * (a & 1) + c] = p1;
            vb[t][1][2 * tp + (a >> 1)][4 * (a & 1) + c] = (_Float16)(x - (float)p1);
          }
    }
  } else {
    float vr[NT][KK];
    if (NKK == 1 && (vec_in & 2) && n != NKK * 32) {
      // ragged rows stored back to back (config-5-like shapes): whole-line float4 loads of the tile's contiguous
      // block, through the patch as a flat array (rayen_mfma_split.hip)
      float* flat = &patch[0][0];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int64_t row0 = s_base + 32 * t;
        const int64_t left = B - row0;
        const int nfl = (int)(left >= 32 ? 32 : (left > 0 ? left : 0)) * n;
        const float* src = v + row0 * (int64_t)n;
#pragma unroll
        for (int jj = 0; jj < NKK * 4; ++jj) {
          const int i4 = lane + 64 * jj;
          if (i4 < 8 * n) {
            f32x4 x = {0.f, 0.f, 0.f, 0.f};
            if (4 * i4 + 3 < nfl) {
              x = *reinterpret_cast<const f32x4*>(src + 4 * i4);
            } else {
              if (4 * i4 + 0 < nfl) x[0] = src[4 * i4 + 0];
              if (4 * i4 + 1 < nfl) x[1] = src[4 * i4 + 1];
              if (4 * i4 + 2 < nfl) x[2] = src[4 * i4 + 2];
            }
            *reinterpret_cast<f32x4*>(flat + 4 * i4) = x;
          }
        }
        __builtin_amdgcn_wave_barrier();
        const float* myrow = flat + col * n + 4 * hi;
#pragma unroll
        for (int q = 0; q < NKK * 4; ++q)
#pragma unroll
          for (int c = 0; c < 4; ++c) vr[t][4 * q + c] = (8 * q + 4 * hi + c < n) ? myrow[8 * q + c] : 0.f;
        __builtin_amdgcn_wave_barrier();
      }
    } else {
      load_rows<NT, NKK, LSTR, true>(vr, v, ldv, n, vec_in & 1, s_base, B, live, patch, lane);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      // sv = 2^(13 - floor(log2 max|v|)): exponent arithmetic only.  (Biased exponent clamped to [14, 254]: a row
      // below 2^-113 keeps sv finite and loses relative precision where kappa << 1 decides nothing; inf / NaN rows
      // stay inf / NaN.)
      float m = 0.f;
#pragma unroll
      for (int i = 0; i < KK; ++i) m = fmaxf(m, __builtin_fabsf(vr[t][i]));
      m = fmaxf(m, xhalf(m));
      float sv;
      int sv_exp;
      pow2_scale(m, sv, v_inv[t], sv_exp);
      v_scl[t] = sv;
#pragma unroll
      for (int q = 0; q < NKK * 4; ++q)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int i = (q & 1) * 4 + c;
          const float x = vr[t][4 * q + c] * sv;
          const _Float16 p1 = (_Float16)x;
          const float r1 = x - (float)p1;
          vb[t][0][q >> 1][i] = p1;
          vb[t][1][q >> 1][i] = (_Float16)r1;
        }
    }
  }

  // kap, part, the aux patch and the accumulators live in the SCALED domain (gW sv times the natural value)
  float kap[NT], part[NT], scale[NT], knat[NT];
  int acode[NT];  // arg-max bookkeeping in one register: (segment << 20) | row, -1 = none
#pragma unroll
  for (int t = 0; t < NT; ++t) { kap[t] = 0.f; part[t] = 0.f; scale[t] = 1.f; knat[t] = 0.f; acode[t] = -1; }

  auto finish_kappa = [&]() {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float other = xhalf(kap[t]);
      if (TRACK) {
        const int ocode = __shfl_xor(acode[t], 32);
        if (other > kap[t] || (other == kap[t] && hi == 1)) acode[t] = ocode;
      }
      kap[t] = fmaxf(kap[t], other);
      knat[t] = (kap[t] * w_inv) * v_inv[t];
      // what multiplies the scaled numbers on the way out: the accumulators of NA_E rows carry gW sv, the rebuilt
      // direction of the NA_E = I write-out only sv
      const float out = v_inv[t] * (1.0f / fmaxf(1.0f, knat[t]));
      scale[t] = identity ? out : out * w_inv;
    }
  };

  f32x16 acc[NT];
  // (which tile an item reads and which K-steps of it come with the PREVIOUS item's record -- MItem::qbegin, mfma_pair_build:
  // nothing in front of a burst waits for a scalar load)
  int ts_next = items[0].tile_shape;
  bool after_half_b = false;   // the previous item was the second half of a shared tile (its loads sit later in the queue)
  for (int it = 0; it < n_items; ++it) {
    const MItem item = items[it];
    const int ts = ts_next;
    ts_next = item.qbegin;
    if (item.type == MI_OUT && (item.flags & MF_FIRST)) finish_kappa();
    // the tile after this one; the last tile of a group fetches tile 0 for the next group
    const char* next_tile = reinterpret_cast<const char*>(Wh) + (size_t)(((ts >> 30) & 1) ? 0 : (ts & 0xFFFFFF) + 1) * (NCH * 1024);
    {
      const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      __builtin_amdgcn_s_setprio(0);
      // Two passes over the K-steps, by product size: the 2^-11 cross products of ALL steps first, the leading
      // products last (the instruction aligns its products and C to the largest and keeps ~26 bits: small products go
      // in while the accumulator is still small, scripts/ubench/mfma_bf16_acc.hip)
      // Which K-steps of the tile an item multiplies (`k_*`) and streams (`s_*`: waits for, re-loads for the next tile):
      // a full tile all of them; the halves of a shared tile (NS = 4, rayen_tiles.h) K-steps 0,1 or 2,3; a block alone in
      // its tile multiplies K-steps 2,3 and streams all.  ONE instruction stream: a K-step -- wait, MFMAs, re-load -- is a
      // single statement with its branches inside (rayen_split_image.h::pair_kstep1 / pair_kstep2).
      // Counted waits: a chunk's re-load for the NEXT tile follows its last use, so K-step sp of a full tile behind a full
      // tile has NS - 1 younger loads behind its chunks; the second half of a shared tile has the first half's four
      // re-loads behind them on top, and so have K-steps 0,1 of the tile after it.
      if constexpr (NS == 4 && NKX == 0) {
        const int shape = (ts >> 24) & 3;
        const int ctrl = pair_item_ctrl(shape, after_half_b);
        auto step1 = [&](auto SP) {
          constexpr int sp = decltype(SP)::value;
          f16x8 b1[NT], b2[NT];
#pragma unroll
          for (int t = 0; t < NT; ++t) { b1[t] = vb[t][0][sp]; b2[t] = vb[t][1][sp]; }
          __builtin_amdgcn_sched_barrier(0);
          pair_kstep1<NT, sp, 3, 5>(abuf[2 * sp + 0], abuf[2 * sp + 1], acc, b1, b2, ctrl, next_tile + (2 * sp + 1) * 1024, lane_off);
          __builtin_amdgcn_sched_barrier(0);
        };
        auto step2 = [&](auto SP) {
          constexpr int sp = decltype(SP)::value;
          f16x8 b1[NT];
#pragma unroll
          for (int t = 0; t < NT; ++t) b1[t] = vb[t][0][sp];
          __builtin_amdgcn_sched_barrier(0);
          pair_kstep2<NT, sp, sp == NS - 1>(abuf[2 * sp + 0], acc, b1, ctrl, next_tile + (2 * sp + 0) * 1024, lane_off);
          __builtin_amdgcn_sched_barrier(0);
        };
        step1(std::integral_constant<int, 0>{}); step1(std::integral_constant<int, 1>{});
        step1(std::integral_constant<int, 2>{}); step1(std::integral_constant<int, 3>{});
        step2(std::integral_constant<int, 0>{}); step2(std::integral_constant<int, 1>{});
        step2(std::integral_constant<int, 2>{}); step2(std::integral_constant<int, 3>{});
        after_half_b = shape == MS_HALF_B;
      } else {
        // n_pad = 32, and the instances behind the fused mapper (their image is laid out without shared tiles,
        // mfma_pair_build_dense: as statements the burst cost them 20 more spilled registers, 0.101 against 0.088 ms on
        // config 3 behind a 64-column mapper): every item is a full tile; the builtins, scheduled by hipcc (rounds 3-4)
#pragma unroll
        for (int sp = 0; sp < NS; ++sp) {
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (NS == 4) asm volatile("s_waitcnt vmcnt(3)" : "+v"(abuf[2 * sp + 0]), "+v"(abuf[2 * sp + 1]));
          else asm volatile("s_waitcnt vmcnt(1)" : "+v"(abuf[2 * sp + 0]), "+v"(abuf[2 * sp + 1]));
          const f16x8 a1 = __builtin_bit_cast(f16x8, abuf[2 * sp + 0]), a2 = __builtin_bit_cast(f16x8, abuf[2 * sp + 1]);
#pragma unroll
          for (int t = 0; t < NT; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, vb[t][0][sp], sp == 0 ? zero : acc[t], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, vb[t][1][sp], acc[t], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          pair_reload(abuf[2 * sp + 1], next_tile + (2 * sp + 1) * 1024, lane_off);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int sp = 0; sp < NS; ++sp) {
          const f16x8 a1 = __builtin_bit_cast(f16x8, abuf[2 * sp + 0]);
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, vb[t][0][sp], acc[t], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          pair_reload(abuf[2 * sp + 0], next_tile + (2 * sp + 0) * 1024, lane_off);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __builtin_amdgcn_s_setprio(1);
    }
    if (item.type == MI_LIN) {
      const int lin_code = (item.seg << 20) + item.row0 + 4 * hi;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (TRACK) {
#pragma unroll
          for (int g = 0; g < 16; ++g)
            if (acc[t][g] > kap[t]) {
              kap[t] = acc[t][g];
              acode[t] = lin_code + ((g & 3) + 8 * (g >> 2));
            }
        } else {
#pragma unroll
          for (int g = 0; g < 16; ++g) kap[t] = fmaxf(kap[t], acc[t][g]);
        }
      }
    } else if (item.type == MI_QFAC || item.type == MI_SOC) {
      // a running sum of squares over the segment's tiles, closed on its last tile
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        f32x2 s2 = {(item.flags & MF_FIRST) ? 0.f : part[t], 0.f};
#pragma unroll
        for (int g = 0; g < 16; g += 2) {
          const f32x2 a2 = {acc[t][g], acc[t][g + 1]};
          s2 = __builtin_elementwise_fma(a2, a2, s2);
        }
        part[t] = s2[0] + s2[1];
      }
      if (item.flags & MF_LAST) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const float total = part[t] + xhalf(part[t]);
          const float a0 = aux_lds[wave][t][item.aux][col];
          float kc;
          if (item.type != MI_SOC) {
            kc = (a0 + __builtin_amdgcn_sqrtf(fmaxf(total, 0.f))) * item.seg_inv;   // (the segment's own power of two undone)
          } else {
            kc = pair_soc_candidate(a0, aux_lds[wave][t][item.aux + 1][col], total, w_inv * item.seg_inv, v_inv[t], item.f0,
                                    item.f1, v_scl[t], w_scale);
          }
          if (kc > kap[t]) { kap[t] = kc; acode[t] = item.seg << 20; }
        }
      }
    } else if (item.type == MI_AUX) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 16; ++g)
          aux_lds[wave][t][(g & 3) + 8 * (g >> 2) + 4 * hi][col] = acc[t][g];
      __builtin_amdgcn_wave_barrier();
    } else if (STAGED && NKK == 2 && item.type == MI_OUT) {
      // n > 32: rows of NA_E leave as 16-byte pieces straight from the accumulators (rayen_mfma_split.hip)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (!live[t]) continue;
        float* yrow = y + (s_base + t * 32 + col) * ldy;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const int r0 = item.row0 + 8 * a + 4 * hi;
          if (r0 >= k) continue;
          f32x4 o;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            o[c] = fmaf(acc[t][4 * a + c], scale[t], y0[r0 + c]);  // y0 is padded to a tile multiple
            bad |= (o[c] != o[c]) && (r0 + c < k);
          }
          if (vec_out && r0 + 3 < k) {
            *reinterpret_cast<f32x4*>(yrow + r0) = o;
          } else {
#pragma unroll
            for (int c = 0; c < 4; ++c)
              if (r0 + c < k) yrow[r0 + c] = o[c];
          }
        }
      }
    } else if (STAGED && item.type == MI_OUT) {
      // rows of NA_E: through this wave's aux patch (XOR-swizzled), out as row-coalesced stores
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        float* stage = &aux_lds[wave][t][0][0];
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          const int r = (g & 3) + 8 * (g >> 2) + 4 * hi;
          const float o = fmaf(acc[t][g], scale[t], y0[item.row0 + r]);  // y0 is padded to a tile multiple
          bad |= live[t] && (item.row0 + r < k) && (o != o);
          stage[col * 32 + (r ^ col)] = o;
        }
        __builtin_amdgcn_wave_barrier();
        const int orow = item.row0 + col;
        float* ybase = y + (s_base + t * 32 + hi) * ldy + orow;
#pragma unroll 4
        for (int j = 0; j < 16; ++j) {
          const int sm = 2 * j + hi;
          const float o = stage[sm * 32 + (col ^ sm)];
          if (s_base + t * 32 + sm < B && orow < k) ybase[(int64_t)(2 * j) * ldy] = o;
        }
        __builtin_amdgcn_wave_barrier();
      }
    } else if (item.type == MI_PACK) {
      const MPack pk = packs[item.aux];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int slot = hi ? pk.aux[a][1] : pk.aux[a][0];
        const int sid = hi ? pk.seg[a][1] : pk.seg[a][0];
        const bool pair = (item.row0 >> a) & 1;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          float qs = acc[t][4 * a] * acc[t][4 * a];
#pragma unroll
          for (int c = 1; c < 4; ++c) qs = fmaf(acc[t][4 * a + c], acc[t][4 * a + c], qs);
          if (pair) qs += xhalf(qs);
          const float kc = (aux_lds[wave][t][slot & 31][col] + __builtin_amdgcn_sqrtf(qs)) * (hi ? pk.inv[a][1] : pk.inv[a][0]);
          if (sid >= 0 && kc > kap[t]) { kap[t] = kc; acode[t] = sid << 20; }
        }
      }
    }
  }

  if constexpr (NKX > 0) {
    // (mapped instances discard the prefetched tile -- the mapper needs its registers -- but the loads are still in
    // flight here: the chunks stay live up to this wait so that the compiler cannot hand them out before)
    if constexpr (NCH == 8)
      asm volatile("s_waitcnt vmcnt(0)"
                   : "+v"(abuf[0]), "+v"(abuf[1]), "+v"(abuf[2]), "+v"(abuf[3]), "+v"(abuf[4]), "+v"(abuf[5]), "+v"(abuf[6]), "+v"(abuf[7])
                   :
                   : "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(abuf[0]), "+v"(abuf[1]), "+v"(abuf[2]), "+v"(abuf[3]) : : "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next group's first tile has landed
  }

  if (identity) {
    finish_kappa();
    // y = y0 + v / max(1, kappa): v rebuilt from its pieces (22 bits of it; scaled by sv, undone by `scale`)
    float vr[NT][KK];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int sp = 0; sp < NS; ++sp)
#pragma unroll
        for (int i = 0; i < 8; ++i)
          // element i of K-step sp = register 4 (2 sp + (i >> 2)) + (i & 3) of the fp32 layout
          vr[t][4 * (2 * sp + (i >> 2)) + (i & 3)] = (float)vb[t][0][sp][i] + (float)vb[t][1][sp][i];
    // (full-line rows leave as non-temporal stores: -3 % on config 3; the 4-byte stores of the NA_E write-out above
    // lose 15 % with the same hint)
    bad |= store_rows<NT, NKK, LSTR, true, true>(vr, scale, y0_lds, y, ldy, k, vec_out, s_base, B, live, patch, lane);
  }

  if (hi == 0) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (!live[t]) continue;
      const int64_t s = s_base + t * 32 + col;
      if (kappa_out) kappa_out[s] = knat[t];
      if (TRACK) { active_out[2 * s] = acode[t] >> 20; active_out[2 * s + 1] = acode[t] < 0 ? 0 : (acode[t] & 0xFFFFF); }
    }
  }
  }  // persistent loop over sample groups
  if (nan_flag && bad) atomicOr(nan_flag, 1);
}

// NT = 1: 32 samples per wave for batches that cannot fill the chip with 64-sample groups (config 2 at B = 4096: 128
// waves of half the work instead of 64; selected when B / 32 <= resident waves)
template <int NKK, bool TRACK, bool STAGED, int NT = 2>
__global__ __launch_bounds__(kMfmaWaves * 64, kMfmaWaves / 4) void mfma_pair_fwd_kernel(
    const f16x8* __restrict__ Wh, const MItem* __restrict__ items, int n_items,
    const MPack* __restrict__ packs, const float* __restrict__ y0, int identity, int k, int n,
    const float* __restrict__ v, int64_t B, int64_t ldv, int vec_in, float* __restrict__ y, int64_t ldy,
    int vec_out, float* __restrict__ kappa_out, int32_t* __restrict__ active_out,
    int32_t* __restrict__ nan_flag, const float w_scale, const float w_inv) {
  mfma_pair_fwd_body<NKK, TRACK, STAGED, 0, NT>(Wh, items, n_items, packs, y0, identity, k, n, v, B, ldv, vec_in, y, ldy,
                                                vec_out, kappa_out, active_out, nan_flag, w_scale, w_inv, PairMapper());
}

// the same walk behind the fused mapper (x in place of v; NKX 32-column blocks of x)
template <int NKK, bool TRACK, int NKX, bool STAGED = false>
__global__ __launch_bounds__(kMfmaWaves * 64, kMfmaWaves / 4) void mfma_pair_map_kernel(
    const f16x8* __restrict__ Wh, const MItem* __restrict__ items, int n_items,
    const MPack* __restrict__ packs, const float* __restrict__ y0, int identity, int k, int n,
    const float* __restrict__ x, int64_t B, int64_t ldx, int vec_in, float* __restrict__ y, int64_t ldy,
    int vec_out, float* __restrict__ kappa_out, int32_t* __restrict__ active_out,
    int32_t* __restrict__ nan_flag, const float w_scale, const float w_inv, const PairMapper mp) {
  mfma_pair_fwd_body<NKK, TRACK, STAGED, NKX>(Wh, items, n_items, packs, y0, identity, k, n, x, B, ldx, vec_in, y, ldy,
                                             vec_out, kappa_out, active_out, nan_flag, w_scale, w_inv, mp);
}

// Wm [n, ldw] (row-major fp32, torch.nn.Linear.weight) -> the image the mapped kernel reads.  ONE workgroup: the
// largest |entry| (LDS reduction) fixes gM = the power of two that puts it into [2^13, 2^14); then chunk (row tile
// tp, K-step s, piece) = 64 lanes x 8 f16, element i of lane l = piece of gM Wm[32 tp + (l & 31)][16 s + 8 (i >> 2) +
// 4 (l >> 5) + (i & 3)], zero beyond (n, in_dim); then the bias (zero-padded to n_pad floats), gM and 1 / gM.
__global__ __launch_bounds__(256) void pair_mapper_image_kernel(const float* __restrict__ w, int64_t ldw,
                                                                const float* __restrict__ bias, int n, int in_dim,
                                                                int nkk, int nsx, f16x8* __restrict__ img) {
  __shared__ float red[256];
  float m = 0.f;
  for (int i = threadIdx.x; i < n * in_dim; i += 256) {
    const float x = __builtin_fabsf(w[(int64_t)(i / in_dim) * ldw + (i % in_dim)]);
    if (x < __builtin_inff()) m = fmaxf(m, x);     // (NaN / inf entries do not set the scale; they propagate)
  }
  red[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  unsigned e = __builtin_bit_cast(unsigned, red[0]) >> 23;
  e = red[0] > 0.f ? (e < 67u ? 67u : (e > 200u ? 200u : e)) : 140u;     // |shift| <= 73
  const float gm = __builtin_bit_cast(float, (267u - e) << 23);
  const float gm_inv = __builtin_bit_cast(float, (e - 13u) << 23);
  const int l = threadIdx.x & 63;
  for (int chunk = threadIdx.x >> 6; chunk < nkk * nsx; chunk += 4) {
    const int tp = chunk / nsx, sx = chunk - tp * nsx;
    const int row = 32 * tp + (l & 31);
    f16x8 o1, o2;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int col = 16 * sx + 8 * (i >> 2) + 4 * (l >> 5) + (i & 3);
      const float x = ((row < n && col < in_dim) ? w[(int64_t)row * ldw + col] : 0.f) * gm;
      const _Float16 p1 = (_Float16)x;
      o1[i] = p1;
      o2[i] = (_Float16)(x - (float)p1);
    }
    f16x8* dst = img + (size_t)chunk * 2 * 64 + l;
    dst[0] = o1;
    dst[64] = o2;
  }
  float* tail = reinterpret_cast<float*>(img + (size_t)nkk * nsx * 2 * 64);
  for (int i = threadIdx.x; i < nkk * 32; i += 256) tail[i] = (bias != nullptr && i < n) ? bias[i] : 0.f;
  if (threadIdx.x == 0) { tail[nkk * 32] = gm; tail[nkk * 32 + 1] = gm_inv; }
}

// ---------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------
void mfma_pair_free(PairImage* img) {
  if (img == nullptr) return;
  if (img->Wh) (void)hipFree(img->Wh);
  if (img->items) (void)hipFree(img->items);
  if (img->packs) (void)hipFree(img->packs);
  if (img->y0) (void)hipFree(img->y0);
  delete img;
}

static int pair_build(const RayenPack* p, PairImage** out, int64_t* bytes, bool tri);
int mfma_pair_build(const RayenPack* p, PairImage** out, int64_t* bytes) {
  // (RAYEN_PAIR_TRI=0: dense factors, two tiles each -- the image of rounds 3 and 4, for A/B measurements)
  const char* tri_env = std::getenv("RAYEN_PAIR_TRI");
  return pair_build(p, out, bytes, !(tri_env != nullptr && tri_env[0] == '0'));
}
bool mfma_pair_has_halves(const PairImage* img) { return img != nullptr && img->has_halves; }
// the image of the instances behind the fused mapper: every item a full tile
int mfma_pair_build_dense(const RayenPack* p, PairImage** out, int64_t* bytes) { return pair_build(p, out, bytes, false); }

static int pair_build(const RayenPack* p, PairImage** out, int64_t* bytes, const bool tri) {
  TileLayout b(p->n);
  const int rc = layout_tiles(p, b, /*allow_pack=*/true, /*allow_sym=*/false, tri);
  if (rc != RAYEN_OK) return rc;
  if (b.packs.empty()) { MPack none; std::memset(&none, 0, sizeof(none)); b.packs.push_back(none); }
  // an item also carries the NEXT item's tile and shape (in `qbegin`, which only symmetric-form layouts use): the walks
  // need them in front of an item's burst, and a scalar load issued there would be waited for there
  for (size_t i = 0; i < b.items.size(); ++i) b.items[i].qbegin = b.items[i + 1 < b.items.size() ? i + 1 : i].tile_shape;
  // ---- one power of two per quadratic / cone on top of the image's gW (round 3).  f16 has five exponent bits: with ONE
  // scale for the whole image, a constraint whose rows are 2^-15 of the image's largest entry keeps only its leading
  // pieces (config 5's jerk limits next to its corridor rows: 3e-5 -- the creation-time measurement sent the set to the
  // bf16 triples).  A candidate phi.v + ||U v|| is homogeneous in ITS OWN rows (aux rows and factor rows together), so
  // every such segment's rows are boosted by f_s = 2^e_s into the band the image's largest entry sits in, and its
  // candidate is multiplied by 1 / f_s (exact) before it meets the running maximum (MItem::seg_inv, MPack::inv).
  // Linear rows keep the image's scale (their maximum runs over rows of different segments' worth of scale).
  std::vector<float> seg_inv(p->segs.size(), 1.f);
  {
    // who owns an entry of the image: [tile row][column half] (a shared tile's rows belong to one segment in columns
    // 0..31 and to another in columns 32..63, rayen_tiles.h)
    const int n_tiles0 = b.n_tiles();
    const int half_w = b.n_pad >= 64 ? 32 : b.n_pad;
    std::vector<int> cell_seg((size_t)n_tiles0 * 32 * 2, -1);
    auto own_row = [&](int tile, int r, int shape, int seg) {
      if (shape != MS_HALF_B) cell_seg[((size_t)tile * 32 + r) * 2 + 0] = seg;
      if (shape != MS_HALF_A) cell_seg[((size_t)tile * 32 + r) * 2 + 1] = seg;
    };
    int cur_aux = -1;
    for (size_t idx = 0; idx < b.items.size(); ++idx) {
      const MItem& it = b.items[idx];
      if (it.type == MI_AUX) cur_aux = it.tile();
      if (it.type == MI_QFAC || it.type == MI_SOC) {
        for (int r = 0; r < 32; ++r) own_row(it.tile(), r, it.shape(), it.seg);
        if (cur_aux >= 0) {
          own_row(cur_aux, it.aux, MS_FULL, it.seg);
          if (it.type == MI_SOC) own_row(cur_aux, it.aux + 1, MS_FULL, it.seg);
        }
      }
      if (it.type == MI_PACK) {
        const MPack& pk = b.packs[it.aux];
        for (int a = 0; a < 4; ++a)
          for (int h = 0; h < 2; ++h) {
            if (pk.seg[a][h] < 0) continue;
            for (int c = 0; c < 4; ++c) own_row(it.tile(), 8 * a + 4 * h + c, MS_FULL, pk.seg[a][h]);
            if (cur_aux >= 0) own_row(cur_aux, pk.aux[a][h], MS_FULL, pk.seg[a][h]);
          }
      }
    }
    auto cell_of = [&](size_t r, int c) { return cell_seg[r * 2 + (c >= half_w ? 1 : 0)]; };
    double image_big = 0.0;
    std::vector<double> seg_big(p->segs.size(), 0.0);
    for (size_t r = 0; r < (size_t)n_tiles0 * 32; ++r)
      for (int c = 0; c < b.n_pad; ++c) {
        const double x = std::fabs(b.raw[r * b.n_pad + c]);
        if (!std::isfinite(x)) continue;
        image_big = x > image_big ? x : image_big;
        const int sg = cell_of(r, c);
        if (sg >= 0 && x > seg_big[sg]) seg_big[sg] = x;
      }
    std::vector<double> boost(p->segs.size(), 1.0);
    for (size_t s = 0; s < p->segs.size(); ++s) {
      if (!(seg_big[s] > 0.0) || !(image_big > 0.0)) continue;
      int ex_seg = 0, ex_img = 0;
      (void)std::frexp(seg_big[s], &ex_seg);
      (void)std::frexp(image_big, &ex_img);
      int e = ex_img - ex_seg;                    // the segment's largest entry into the binade of the image's
      e = e < 0 ? 0 : (e > 60 ? 60 : e);
      boost[s] = std::ldexp(1.0, e);
      seg_inv[s] = (float)std::ldexp(1.0, -e);
    }
    for (size_t r = 0; r < (size_t)n_tiles0 * 32; ++r)
      for (int c = 0; c < b.n_pad; ++c) {
        const int sg = cell_of(r, c);
        if (sg >= 0 && boost[sg] != 1.0) b.raw[r * b.n_pad + c] *= boost[sg];
      }
    for (MItem& it : b.items)
      if (it.type == MI_QFAC || it.type == MI_SOC) it.seg_inv = seg_inv[it.seg];
    for (MPack& pk : b.packs)
      for (int a = 0; a < 4; ++a)
        for (int h = 0; h < 2; ++h) pk.inv[a][h] = pk.seg[a][h] >= 0 ? seg_inv[pk.seg[a][h]] : 1.f;
  }
  const std::vector<float> frag = b.fragments_f32();

  PairImage* img = new PairImage();
  img->nkk = b.n_pad / 32;
  img->identity = p->out_identity;
  img->n_items = (int)b.items.size();
  img->host_items = b.items;
  for (const MItem& it : b.items) img->has_halves = img->has_halves || it.shape() != MS_FULL;
  for (const RayenSegment& g : p->segs) img->aux_rows += aux_rows_of(g);
  img->first_out = img->n_items;
  for (int i = img->n_items - 1; i >= 0; --i)
    if (b.items[i].type == MI_OUT) img->first_out = i;
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, p->device) == hipSuccess && prop.multiProcessorCount > 0)
      img->n_simd = prop.multiProcessorCount * 4;
  }
  // gW: the largest entry of the image into [2^13, 2^14)
  float big = 0.f;
  for (const float x : frag)
    if (std::isfinite(x)) big = std::fmax(big, std::fabs(x));
  int ex = 0;
  if (big > 0.f) (void)std::frexp(big, &ex);   // big = f 2^ex, f in [0.5, 1)
  int shift = big > 0.f ? 14 - ex : 0;
  shift = shift > 100 ? 100 : (shift < -100 ? -100 : shift);   // (beyond: f16 overflow -> the self-check rejects the pack)
  img->w_scale = std::ldexp(1.0f, shift);
  img->w_inv = std::ldexp(1.0f, -shift);
  // two f16 pieces of every scaled entry, in the fragment order of v_mfma_f32_32x32x16_f16 (the bf16 instruction's):
  // chunk (tile, k-step s, piece) = 64 lanes x 8 elements, element i of lane l = column
  // 16 s + 8 (i >> 2) + 4 (l >> 5) + (i & 3) of row l & 31 = entry [2 s + (i >> 2)][l][i & 3] of the fp32 image
  const int n_tiles = b.n_tiles(), ns = b.nq() / 2;
  img->n_tiles = n_tiles;
  std::vector<_Float16> wh((size_t)n_tiles * ns * 2 * 64 * 8);
  for (int t = 0; t < n_tiles; ++t)
    for (int sp = 0; sp < ns; ++sp)
      for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 8; ++i) {
          const float x = frag[(((size_t)t * b.nq() + 2 * sp + (i >> 2)) * 64 + l) * 4 + (i & 3)] * img->w_scale;
          const _Float16 h1 = (_Float16)x;                    // round to nearest even
          const _Float16 h2 = (_Float16)(x - (float)h1);      // (exact difference)
          const size_t base = (((size_t)t * ns + sp) * 2) * 64 * 8 + (size_t)l * 8 + i;
          wh[base] = h1;
          wh[base + 64 * 8] = h2;
        }
  const int k_tiles = (p->k + 31) / 32;
  std::vector<float> y0((size_t)k_tiles * 32 + 32, 0.f);
  for (int i = 0; i < p->k; ++i) y0[i] = (float)p->y0[i];
  const bool ok =
      hipMalloc(&img->Wh, wh.size() * 2) == hipSuccess &&
      hipMemcpy(img->Wh, wh.data(), wh.size() * 2, hipMemcpyHostToDevice) == hipSuccess &&
      hipMalloc(&img->y0, y0.size() * sizeof(float)) == hipSuccess &&
      hipMemcpy(img->y0, y0.data(), y0.size() * sizeof(float), hipMemcpyHostToDevice) == hipSuccess &&
      hipMalloc(&img->items, b.items.size() * sizeof(MItem)) == hipSuccess &&
      hipMemcpy(img->items, b.items.data(), b.items.size() * sizeof(MItem), hipMemcpyHostToDevice) == hipSuccess &&
      hipMalloc(&img->packs, b.packs.size() * sizeof(MPack)) == hipSuccess &&
      hipMemcpy(img->packs, b.packs.data(), b.packs.size() * sizeof(MPack), hipMemcpyHostToDevice) == hipSuccess;
  if (!ok) { mfma_pair_free(img); return RAYEN_E_ALLOC; }
  img->bytes = (int64_t)(wh.size() * 2 + y0.size() * sizeof(float) + b.items.size() * sizeof(MItem) +
                         b.packs.size() * sizeof(MPack));
  *bytes = img->bytes;
  *out = img;
  return RAYEN_OK;
}

template <int NKK, int NT>
static int launch_pair(const RayenPack* p, const PairImage* img, const float* v, int64_t B, int64_t ldv,
                       float* y, int64_t ldy, float* kappa, int32_t* active, int32_t* nan_flag,
                       hipStream_t stream) {
  constexpr int per_wave = NT * 32;
  const int64_t n_groups = (B + per_wave - 1) / per_wave;
  const int64_t slots = (int64_t)launch_simds(img->n_simd) * kMfmaWavesPerSimd;
  const int64_t rounds = (n_groups + slots - 1) / slots;
  const int64_t waves = (n_groups + rounds - 1) / rounds;
  const int64_t grid = (waves + kMfmaWaves - 1) / kMfmaWaves;
  // bit 0: rows are 16-byte aligned | bit 1: rows are stored back to back and the base is 16-byte aligned
  const int vec_in = (((ldv % 4 == 0) && ((reinterpret_cast<uintptr_t>(v) & 15) == 0)) ? 1 : 0) |
                     ((ldv == p->n && (reinterpret_cast<uintptr_t>(v) & 15) == 0) ? 2 : 0);
  const int vec_out = (ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0);
  auto go = [&](auto kern) {
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kMfmaWaves * 64), 0, stream,
                       static_cast<const f16x8*>(img->Wh), img->items, img->n_items, img->packs, img->y0,
                       img->identity, p->k, p->n, v, B, ldv, vec_in, y, ldy, vec_out, kappa, active, nan_flag,
                       img->w_scale, img->w_inv);
  };
  if (img->identity) {
    if (active != nullptr) go(mfma_pair_fwd_kernel<NKK, true, false, NT>);
    else go(mfma_pair_fwd_kernel<NKK, false, false, NT>);
  } else {
    if (active != nullptr) go(mfma_pair_fwd_kernel<NKK, true, true, NT>);
    else go(mfma_pair_fwd_kernel<NKK, false, true, NT>);
  }
  return hipGetLastError() == hipSuccess ? RAYEN_OK : RAYEN_E_LAUNCH;
}

// ---- fused mapper: in_dim <= n_pad columns of x (sets with equality constraints: the STAGED instances)
int64_t mfma_pair_mapper_image_bytes(const RayenPack* p, const PairImage* img, int in_dim) {
  (void)p;
  if (img == nullptr || in_dim < 1 || in_dim > img->nkk * 32) return 0;
  const int nsx = (in_dim + 31) / 32 * 2;
  return (int64_t)img->nkk * nsx * 2 * 1024 + (int64_t)(img->nkk * 32 + 4) * sizeof(float);
}

int mfma_pair_mapper_prepare(const RayenPack* p, const PairImage* img, const float* w, int64_t ldw, int in_dim,
                             const float* bias, void* image, hipStream_t stream) {
  if (mfma_pair_mapper_image_bytes(p, img, in_dim) == 0) return RAYEN_E_UNSUPPORTED;
  const int nsx = (in_dim + 31) / 32 * 2;
  hipLaunchKernelGGL(pair_mapper_image_kernel, dim3(1), dim3(256), 0, stream, w, ldw, bias, p->n, in_dim, img->nkk, nsx,
                     static_cast<f16x8*>(image));
  return hipGetLastError() == hipSuccess ? RAYEN_OK : RAYEN_E_LAUNCH;
}

template <int NKK, int NKX>
static int launch_pair_map(const RayenPack* p, const PairImage* img, const float* x, int64_t B, int64_t ldx,
                           const PairMapper& mp, float* y, int64_t ldy, float* kappa, int32_t* active,
                           int32_t* nan_flag, hipStream_t stream) {
  constexpr int per_wave = 64;
  const int64_t n_groups = (B + per_wave - 1) / per_wave;
  const int64_t slots = (int64_t)launch_simds(img->n_simd) * kMfmaWavesPerSimd;
  const int64_t rounds = (n_groups + slots - 1) / slots;
  const int64_t waves = (n_groups + rounds - 1) / rounds;
  const int64_t grid = (waves + kMfmaWaves - 1) / kMfmaWaves;
  const int vec_in = ((ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0)) ? 1 : 0;
  const int vec_out = (ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0);
  auto go = [&](auto kern) {
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kMfmaWaves * 64), 0, stream,
                       static_cast<const f16x8*>(img->Wh), img->items, img->n_items, img->packs, img->y0,
                       img->identity, p->k, p->n, x, B, ldx, vec_in, y, ldy, vec_out, kappa, active, nan_flag,
                       img->w_scale, img->w_inv, mp);
  };
  if (img->identity) {
    if (active != nullptr) go(mfma_pair_map_kernel<NKK, true, NKX>);
    else go(mfma_pair_map_kernel<NKK, false, NKX>);
  } else {
    if (active != nullptr) go(mfma_pair_map_kernel<NKK, true, NKX, true>);
    else go(mfma_pair_map_kernel<NKK, false, NKX, true>);
  }
  return hipGetLastError() == hipSuccess ? RAYEN_OK : RAYEN_E_LAUNCH;
}

int mfma_pair_forward_mapped(const RayenPack* p, const PairImage* img, const float* x, int64_t B, int64_t ldx,
                             int in_dim, const void* image, float* v_out, int64_t ldvo, float* y, int64_t ldy,
                             float* kappa, int32_t* active, int32_t* nan_flag, hipStream_t stream) {
  if (mfma_pair_mapper_image_bytes(p, img, in_dim) == 0 || image == nullptr || img->has_halves) return RAYEN_E_UNSUPPORTED;
  if (B == 0) return RAYEN_OK;
  PairMapper mp;
  mp.img = static_cast<const f16x8*>(image);
  mp.in_dim = in_dim;
  mp.v_out = v_out;
  mp.ldvo = ldvo;
  const int nkx = (in_dim + 31) / 32;
  if (img->nkk == 1 && nkx == 1) return launch_pair_map<1, 1>(p, img, x, B, ldx, mp, y, ldy, kappa, active, nan_flag, stream);
  if (img->nkk == 2 && nkx == 1) return launch_pair_map<2, 1>(p, img, x, B, ldx, mp, y, ldy, kappa, active, nan_flag, stream);
  if (img->nkk == 2 && nkx == 2) return launch_pair_map<2, 2>(p, img, x, B, ldx, mp, y, ldy, kappa, active, nan_flag, stream);
  return RAYEN_E_UNSUPPORTED;
}

int mfma_pair_forward(const RayenPack* p, const PairImage* img, const float* v, int64_t B, int64_t ldv,
                      float* y, int64_t ldy, float* kappa, int32_t* active, int32_t* nan_flag,
                      hipStream_t stream) {
  if (B == 0) return RAYEN_OK;
  // 32 samples per wave while that still is one round (config 3, B = 65536: 23.9 against 25.4 us with 64-sample groups;
  // B = 98304, two rounds of 32-sample groups: 36.3 against ~30)
  const bool small = (B + 31) / 32 <= (int64_t)img->n_simd * kMfmaWavesPerSimd;
  if (img->nkk == 1)
    return small ? launch_pair<1, 1>(p, img, v, B, ldv, y, ldy, kappa, active, nan_flag, stream)
                 : launch_pair<1, 2>(p, img, v, B, ldv, y, ldy, kappa, active, nan_flag, stream);
  if (img->nkk == 2)
    return small ? launch_pair<2, 1>(p, img, v, B, ldv, y, ldy, kappa, active, nan_flag, stream)
                 : launch_pair<2, 2>(p, img, v, B, ldv, y, ldy, kappa, active, nan_flag, stream);
  return RAYEN_E_UNSUPPORTED;
}

}  // namespace rayen
