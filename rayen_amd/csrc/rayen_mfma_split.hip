// Split-operand forward: the fp32 tile walk of rayen_mfma_kernel.h on v_mfma_f32_32x32x16_bf16.
//
// Every fp32 operand is split EXACTLY into three bf16 pieces x = x1 + x2 + x3 (8 significant bits each;
// bf16 has fp32's exponent range, so no scaling is involved) and a product is rebuilt from the six piece
// products of order <= 2^-16 (x1y1, x1y2, x2y1, x1y3, x2y2, x3y1), each exact in the fp32 accumulator.  The
// dropped terms are <= 2^-24 relative, below the rounding error of an fp32 FMA chain: fp32-grade results at
// 6/16 of the fp32 MFMA time.
//
// Shape of the kernel: the fp32 kernel's (persistent, barrier-free, two waves per SIMD, 64 samples per wave).
// What differs:
//   * A operands: three bf16 images of W in the fragment order of the bf16 instruction, streamed from L2 into a
//     ROLLING register buffer -- the three chunks of a K-step are re-loaded for the next tile right after the
//     step's MFMAs were issued, so every load has most of a tile (> 1000 cycles) to land;
//   * the direction lives in registers only as bf16 pieces (B operands).  No epilogue may need it in fp32,
//     so symmetric forms are laid out through their factors (||U v||^2, rayen_tiles.h: allow_sym = false) -- which
//     is also the better conditioned evaluation --, and the NA_E = I write-out rebuilds v = v1 + v2 + v3 (exact).
#include "rayen_split_image.h"

#include <cstdlib>
#include <cstring>
#include <vector>

namespace rayen {

// Wave priority: everything OUTSIDE a tile's MFMA burst (epilogues, group boundary) runs at s_setprio 1, so that a wave
// gets through its VALU / memory sections ahead of its SIMD partner's MFMA burst and returns to the matrix pipe sooner
// (+2 % on configs 3 and 5; the opposite assignment costs 1 %).
// The module's mapper v = Wm x + b (rayen/constraint_module.py:259-263, 525) in front of the walk (NKX > 0
// instances): Wm as a split-operand fragment image built by mapper_image_kernel below (caller-owned memory, rebuilt
// when the weights change), x split into bf16 pieces like v, the fp32 result accumulators -- which ARE in B-operand
// order (rayen_mfma_kernel.h) -- re-split in registers into the walk's B operands.  v reaches memory only when
// v_out != null (training: the backward's input).
struct SplitMapper {
  const bf16x8* img = nullptr;   // [NKK][NSX][3][64] x 8 bf16, then n_pad floats of bias
  int in_dim = 0;
  float* v_out = nullptr;
  int64_t ldvo = 0;
};

__device__ __forceinline__ void split3(const float x, __bf16& p1, __bf16& p2, __bf16& p3) {
  p1 = (__bf16)x;
  const float r1 = x - (float)p1;
  p2 = (__bf16)r1;
  const float r2 = r1 - (float)p2;
  p3 = (__bf16)r2;
}

template <int NKK, bool TRACK, bool STAGED, int NKX>
__device__ __forceinline__ void mfma_split_fwd_body(
    const bf16x8* __restrict__ Wb, const MItem* __restrict__ items, int n_items,
    const MPack* __restrict__ packs, const float* __restrict__ y0, int identity, int k, int n,
    const float* __restrict__ v, int64_t B, int64_t ldv, int vec_in, float* __restrict__ y, int64_t ldy,
    int vec_out, float* __restrict__ kappa_out, int32_t* __restrict__ active_out,
    int32_t* __restrict__ nan_flag, const SplitMapper mp) {
  constexpr int NT = 2, NS = NKK * 2, NCH = NS * 3, KK = NKK * 16;
  __shared__ float aux_lds[kMfmaWaves][NT][32][32];
  __shared__ __attribute__((aligned(16))) float y0_lds[NKK * 32];
  __shared__ __attribute__((aligned(16))) float bias_lds[NKX > 0 ? NKK * 32 : 4];
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
  if constexpr (NKX > 0) {
    const float* bias = reinterpret_cast<const float*>(mp.img + (size_t)NKK * (NKX * 2) * 3 * 64);
    for (int i = threadIdx.x; i < NKK * 32; i += kMfmaWaves * 64) bias_lds[i] = bias[i];
  }
  __syncthreads();  // the only workgroup barrier
  float (*patch)[LSTR] = line_lds[wave];

  // ---- A operands: a rolling register buffer filled by hand-placed loads.  hipcc treats loads from the
  // (read-only) image as freely movable and gathers them at the end of a tile, right in front of their uses;
  // as asm statements they stay where the latency is covered.  The compiler does not see the data arrive, so:
  //   * `wait_step` (s_waitcnt) takes the three chunks as read-write operands -- the MFMAs depend on it;
  //   * asm loads are in flight only inside the tile loop (vmcnt(0) before any code that may spill or copy).
  u32x4 abuf[NCH];
  const unsigned lane_off = lane * 16;
  auto load_step = [&](const char* tile_base, const int sp) {
    const char* sb = tile_base + sp * 3072;
    uint64_t asm_base;
    asm volatile(RAYEN_ASM_BASE_COPY "global_load_dwordx4 %[d], %[off], " RAYEN_ASM_BASE "" : [d] "+v"(abuf[3 * sp + 0]), [b] "=&s"(asm_base) : [off] "v"(lane_off), [base] "s"(sb));
    asm volatile(RAYEN_ASM_BASE_COPY "global_load_dwordx4 %[d], %[off], " RAYEN_ASM_BASE " offset:1024" : [d] "+v"(abuf[3 * sp + 1]), [b] "=&s"(asm_base) : [off] "v"(lane_off), [base] "s"(sb));
    asm volatile(RAYEN_ASM_BASE_COPY "global_load_dwordx4 %[d], %[off], " RAYEN_ASM_BASE " offset:2048" : [d] "+v"(abuf[3 * sp + 2]), [b] "=&s"(asm_base) : [off] "v"(lane_off), [base] "s"(sb));
  };
  // mapped instances: the rolling buffer is dead while the mapper runs (its registers hold x and the mapper's
  // accumulators); tile 0 is fetched afresh once the mapper's MFMAs are issued -- the walk's counted waits cover it
  auto load_step_fresh = [&](const int sp) {
    const char* sb = reinterpret_cast<const char*>(Wb) + sp * 3072;
    uint64_t asm_base;
    asm volatile(RAYEN_ASM_BASE_COPY "global_load_dwordx4 %[d], %[off], " RAYEN_ASM_BASE "" : [d] "=v"(abuf[3 * sp + 0]), [b] "=&s"(asm_base) : [off] "v"(lane_off), [base] "s"(sb));
    asm volatile(RAYEN_ASM_BASE_COPY "global_load_dwordx4 %[d], %[off], " RAYEN_ASM_BASE " offset:1024" : [d] "=v"(abuf[3 * sp + 1]), [b] "=&s"(asm_base) : [off] "v"(lane_off), [base] "s"(sb));
    asm volatile(RAYEN_ASM_BASE_COPY "global_load_dwordx4 %[d], %[off], " RAYEN_ASM_BASE " offset:2048" : [d] "=v"(abuf[3 * sp + 2]), [b] "=&s"(asm_base) : [off] "v"(lane_off), [base] "s"(sb));
  };
  if constexpr (NKX == 0) {
    // tile 0 for the first group.  No wait here: the group's row loads are issued behind these and waited for by
    // the compiler's own counts -- loads return in order, so the tile has landed by then -- and the two round
    // trips overlap.  (Output-only operands: no initialisation of the buffer that could race with the loads.)
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
  // vb[t][piece][k-step] = 8 bf16 = the B operand of one MFMA; element i = column 16 sp + 8 (i >> 2) + 4 hi + (i & 3)
  bf16x8 vb[NT][3][NS];
  if constexpr (NKX > 0) {
    constexpr int NSX = NKX * 2;
    float xr[NT][NKX * 16];
    load_rows<NT, NKX, LSTR, true>(xr, v, ldv, mp.in_dim, vec_in & 1, s_base, B, live, patch, lane);
    f32x16 macc[NKK][NT];
#pragma unroll
    for (int tp = 0; tp < NKK; ++tp)
#pragma unroll
      for (int a4 = 0; a4 < 4; ++a4) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(&bias_lds[32 * tp + 8 * a4 + 4 * hi]);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int c = 0; c < 4; ++c) macc[tp][t][4 * a4 + c] = b4[c];
      }
    const bf16x8* mimg = mp.img + lane;
#pragma unroll
    for (int sx = 0; sx < NSX; ++sx) {
      bf16x8 xb[NT][3];   // the K-step's pieces of x: element i = column 16 sx + 8 (i >> 2) + 4 hi + (i & 3)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          __bf16 p1, p2, p3;
          split3(xr[t][8 * sx + i], p1, p2, p3);
          xb[t][0][i] = p1;
          xb[t][1][i] = p2;
          xb[t][2][i] = p3;
        }
#pragma unroll
      for (int tp = 0; tp < NKK; ++tp) {
        const bf16x8* ch = mimg + (size_t)((tp * NSX + sx) * 3) * 64;
        const bf16x8 a1 = ch[0], a2 = ch[64], a3 = ch[128];
        // smallest products first (the accumulator starts at the bias)
#pragma unroll
        for (int t = 0; t < NT; ++t) macc[tp][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, xb[t][0], macc[tp][t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) macc[tp][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, xb[t][1], macc[tp][t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) macc[tp][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, xb[t][2], macc[tp][t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) macc[tp][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, xb[t][0], macc[tp][t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) macc[tp][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, xb[t][1], macc[tp][t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) macc[tp][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, xb[t][0], macc[tp][t], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int sp = 0; sp < NS; ++sp) load_step_fresh(sp);
    __builtin_amdgcn_sched_barrier(0);
    if (mp.v_out != nullptr) {
      float vr[NT][KK], one[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        one[t] = 1.f;
#pragma unroll
        for (int tp = 0; tp < NKK; ++tp)
#pragma unroll
          for (int g = 0; g < 16; ++g) vr[t][16 * tp + g] = macc[tp][t][g];
      }
      (void)store_rows<NT, NKK, LSTR, true>(vr, one, nullptr, mp.v_out, mp.ldvo, n,
                                            (mp.ldvo % 4 == 0) && ((reinterpret_cast<uintptr_t>(mp.v_out) & 15) == 0),
                                            s_base, B, live, patch, lane);
    }
    // result register g = 4 a + c of row tile tp is direction element 32 tp + 8 a + 4 hi + c = element 4 (a & 1) + c
    // of K-step 2 tp + (a >> 1): split in place into the walk's B operands
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int tp = 0; tp < NKK; ++tp)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            __bf16 p1, p2, p3;
            split3(macc[tp][t][4 * a + c], p1, p2, p3);
            vb[t][0][2 * tp + (a >> 1)][4 * (a & 1) + c] = p1;
            vb[t][1][2 * tp + (a >> 1)][4 * (a & 1) + c] = p2;
            vb[t][2][2 * tp + (a >> 1)][4 * (a & 1) + c] = p3;
          }
  } else {
    float vr[NT][KK];
    if (NKK == 1 && (vec_in & 2) && n != NKK * 32) {  // (n > 32: this path would cost the kernel its last registers)
      // Ragged rows stored back to back (ldv == n, 16-byte aligned base: config-5-like shapes).  A tile's 32 rows
      // are one contiguous, 16-byte aligned block of 32 n floats: it comes in as whole-line float4 loads, goes
      // through the patch as a flat array and is read back row-wise (a lane's own row, 4-byte pieces).  The
      // per-element path below costs a third of the kernel at n = 30.
      float* flat = &patch[0][0];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int64_t row0 = s_base + 32 * t;
        const int64_t left = B - row0;
        const int nfl = (int)(left >= 32 ? 32 : (left > 0 ? left : 0)) * n;  // floats of this tile that exist
        const float* src = v + row0 * (int64_t)n;
#pragma unroll
        for (int jj = 0; jj < NKK * 4; ++jj) {
          const int i4 = lane + 64 * jj;  // float4 index inside the block
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
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int q = 0; q < NKK * 4; ++q)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int i = (q & 1) * 4 + c;
          const float x = vr[t][4 * q + c];
          const __bf16 p1 = (__bf16)x;
          const float r1 = x - (float)p1;
          const __bf16 p2 = (__bf16)r1;
          const float r2 = r1 - (float)p2;
          vb[t][0][q >> 1][i] = p1;
          vb[t][1][q >> 1][i] = p2;
          vb[t][2][q >> 1][i] = (__bf16)r2;
        }
  }

  float kap[NT], part[NT], scale[NT];
  int acode[NT];  // arg-max bookkeeping in one register: (segment << 20) | row, -1 = none (host: < 2048 segments, < 2^20 rows)
#pragma unroll
  for (int t = 0; t < NT; ++t) { kap[t] = 0.f; part[t] = 0.f; scale[t] = 1.f; acode[t] = -1; }

  auto finish_kappa = [&]() {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float other = xhalf(kap[t]);
      if (TRACK) {
        const int ocode = __shfl_xor(acode[t], 32);
        if (other > kap[t] || (other == kap[t] && hi == 1)) acode[t] = ocode;
      }
      kap[t] = fmaxf(kap[t], other);
      scale[t] = 1.0f / fmaxf(1.0f, kap[t]);
    }
  };

  f32x16 acc[NT];
  for (int it = 0; it < n_items; ++it) {
    const MItem item = items[it];
    if (item.type == MI_OUT && (item.flags & MF_FIRST)) finish_kappa();
    // the tile after this one; the last tile of a group fetches tile 0 for the next group
    const char* next_tile = reinterpret_cast<const char*>(Wb) + (size_t)(it + 1 == n_items ? 0 : it + 1) * (NCH * 1024);
    // ---- the tile's MFMAs; each K-step's chunks are re-loaded for the next tile as soon as they were used
    {
      const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      __builtin_amdgcn_s_setprio(0);
      // Three passes over the K-steps, by product size: the 2^-16 products of ALL steps first, then the 2^-8 ones,
      // the leading products last.  The instruction aligns its 16 products and C to the largest of them and keeps
      // ~26 bits (scripts/ubench/mfma_bf16_acc.hip): a small product added to an accumulator that already holds
      // leading products loses its low bits, so the small ones go in while the accumulator is still small.
      auto load_chunk = [&](const int idx) {
        const char* sb = next_tile + idx * 1024;
        uint64_t asm_base;
        asm volatile(RAYEN_ASM_BASE_COPY "global_load_dwordx4 %[d], %[off], " RAYEN_ASM_BASE "" : [d] "+v"(abuf[idx]), [b] "=&s"(asm_base) : [off] "v"(lane_off), [base] "s"(sb));
      };
#pragma unroll
      for (int sp = 0; sp < NS; ++sp) {
        __builtin_amdgcn_sched_barrier(0);
        // the step's three chunks: the youngest (a1, re-loaded in the last pass of the previous tile) has NS - 1 loads behind it
        if constexpr (NS == 4)
          asm volatile("s_waitcnt vmcnt(3)" : "+v"(abuf[3 * sp + 0]), "+v"(abuf[3 * sp + 1]), "+v"(abuf[3 * sp + 2]));
        else
          asm volatile("s_waitcnt vmcnt(1)" : "+v"(abuf[3 * sp + 0]), "+v"(abuf[3 * sp + 1]), "+v"(abuf[3 * sp + 2]));
        const bf16x8 a1 = __builtin_bit_cast(bf16x8, abuf[3 * sp + 0]), a2 = __builtin_bit_cast(bf16x8, abuf[3 * sp + 1]),
                     a3 = __builtin_bit_cast(bf16x8, abuf[3 * sp + 2]);
#pragma unroll
        for (int t = 0; t < NT; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, vb[t][0][sp], sp == 0 ? zero : acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, vb[t][1][sp], acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, vb[t][2][sp], acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        load_chunk(3 * sp + 2);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int sp = 0; sp < NS; ++sp) {
        const bf16x8 a1 = __builtin_bit_cast(bf16x8, abuf[3 * sp + 0]), a2 = __builtin_bit_cast(bf16x8, abuf[3 * sp + 1]);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, vb[t][0][sp], acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, vb[t][1][sp], acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        load_chunk(3 * sp + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int sp = 0; sp < NS; ++sp) {
        const bf16x8 a1 = __builtin_bit_cast(bf16x8, abuf[3 * sp + 0]);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, vb[t][0][sp], acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        load_chunk(3 * sp + 0);
        __builtin_amdgcn_sched_barrier(0);
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
    } else if (item.type == MI_AUX) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 16; ++g)
          aux_lds[wave][t][(g & 3) + 8 * (g >> 2) + 4 * hi][col] = acc[t][g];
      __builtin_amdgcn_wave_barrier();
    } else if (STAGED && NKK == 2 && item.type == MI_OUT) {
      // (n > 32: the staged write-out below does not fit the registers next to 96 B-operand registers -- hipcc
      // spills ~100 of them at the group boundary --, so the rows leave as 16-byte pieces straight from the
      // accumulators)
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
          const float kc = aux_lds[wave][t][slot & 31][col] + __builtin_amdgcn_sqrtf(qs);
          if (sid >= 0 && kc > kap[t]) { kap[t] = kc; acode[t] = sid << 20; }
        }
      }
    } else if (item.type == MI_QFAC || item.type == MI_SOC) {
      // a running sum of squares over the segment's tiles, closed on its last tile
      // (packed fp32 FMAs on the accumulators' own register pairs: 8 + 1 instructions per sample tile)
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
          // (v_sqrt_f32 / v_rcp_f32, 1 ulp: the IEEE-exact forms cost ~10 VALU instructions each, and VALU work
          // is serial with the MFMA stream)
          if (item.type != MI_SOC) {
            kc = a0 + __builtin_amdgcn_sqrtf(fmaxf(total, 0.f));
          } else {
            // a' x^2 + b' x + c' = 0  (rayen/constraint_module.py:392-396, 339-348), a' < 0
            const float br = aux_lds[wave][t][item.aux + 1][col];
            const float cp = total - a0 * a0;
            const float bp = 2.f * br - 2.f * a0 * item.f0;
            const float disc = bp * bp - 4.f * item.f1 * cp;
            kc = 0.f;
            if (disc >= 0.f) {
              const float root = __builtin_amdgcn_sqrtf(disc);
              const float inv2a = 0.5f * __builtin_amdgcn_rcpf(item.f1);
              kc = fmaxf((-bp - root) * inv2a, (-bp + root) * inv2a);
            }
          }
          if (kc > kap[t]) { kap[t] = kc; acode[t] = item.seg << 20; }
        }
      }
    }
  }

  if constexpr (NKX > 0) {
    // (mapped instances discard the prefetched tile -- the mapper needs its registers -- but the loads are still
    // in flight here: the chunks stay live up to this wait so that the compiler cannot hand them out before)
    if constexpr (NCH == 12)
      asm volatile("s_waitcnt vmcnt(0)"
                   : "+v"(abuf[0]), "+v"(abuf[1]), "+v"(abuf[2]), "+v"(abuf[3]), "+v"(abuf[4]), "+v"(abuf[5]), "+v"(abuf[6]),
                     "+v"(abuf[7]), "+v"(abuf[8]), "+v"(abuf[9]), "+v"(abuf[10]), "+v"(abuf[11])
                   :
                   : "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)"
                   : "+v"(abuf[0]), "+v"(abuf[1]), "+v"(abuf[2]), "+v"(abuf[3]), "+v"(abuf[4]), "+v"(abuf[5])
                   :
                   : "memory");
  } else {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next group's first tile has landed
  }

  if (identity) {
    finish_kappa();
    // y = y0 + v / max(1, kappa): v rebuilt from its pieces, v1 + v2 + v3 (exact)
    float vr[NT][KK];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int sp = 0; sp < NS; ++sp) {
        const u32x4 w1 = __builtin_bit_cast(u32x4, vb[t][0][sp]);
        const u32x4 w2 = __builtin_bit_cast(u32x4, vb[t][1][sp]);
        const u32x4 w3 = __builtin_bit_cast(u32x4, vb[t][2][sp]);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int sh = (i & 1) ? 0 : 16;
          const unsigned m = (i & 1) ? 0xFFFF0000u : 0xFFFFFFFFu;
          const float x1 = __builtin_bit_cast(float, (w1[i >> 1] << sh) & m);
          const float x2 = __builtin_bit_cast(float, (w2[i >> 1] << sh) & m);
          const float x3 = __builtin_bit_cast(float, (w3[i >> 1] << sh) & m);
          // element i of K-step sp = register 4 (2 sp + (i >> 2)) + (i & 3) of the fp32 layout
          vr[t][4 * (2 * sp + (i >> 2)) + (i & 3)] = (x1 + x2) + x3;
        }
      }
    bad |= store_rows<NT, NKK, LSTR, true, true>(vr, scale, y0_lds, y, ldy, k, vec_out, s_base, B, live, patch, lane);  // (non-temporal stores: the rows are not read again)
  }

  if (hi == 0) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (!live[t]) continue;
      const int64_t s = s_base + t * 32 + col;
      if (kappa_out) kappa_out[s] = kap[t];
      if (TRACK) { active_out[2 * s] = acode[t] >> 20; active_out[2 * s + 1] = acode[t] < 0 ? 0 : (acode[t] & 0xFFFFF); }
    }
  }
  }  // persistent loop over sample groups
  if (nan_flag && bad) atomicOr(nan_flag, 1);
}

template <int NKK, bool TRACK, bool STAGED>
__global__ __launch_bounds__(kMfmaWaves * 64, kMfmaWaves / 4) void mfma_split_fwd_kernel(
    const bf16x8* __restrict__ Wb, const MItem* __restrict__ items, int n_items,
    const MPack* __restrict__ packs, const float* __restrict__ y0, int identity, int k, int n,
    const float* __restrict__ v, int64_t B, int64_t ldv, int vec_in, float* __restrict__ y, int64_t ldy,
    int vec_out, float* __restrict__ kappa_out, int32_t* __restrict__ active_out,
    int32_t* __restrict__ nan_flag) {
  mfma_split_fwd_body<NKK, TRACK, STAGED, 0>(Wb, items, n_items, packs, y0, identity, k, n, v, B, ldv, vec_in, y, ldy,
                                             vec_out, kappa_out, active_out, nan_flag, SplitMapper());
}

// the same walk behind the fused mapper (x in place of v; NKX 32-column blocks of x)
template <int NKK, bool TRACK, bool STAGED, int NKX>
__global__ __launch_bounds__(kMfmaWaves * 64, kMfmaWaves / 4) void mfma_split_map_kernel(
    const bf16x8* __restrict__ Wb, const MItem* __restrict__ items, int n_items,
    const MPack* __restrict__ packs, const float* __restrict__ y0, int identity, int k, int n,
    const float* __restrict__ x, int64_t B, int64_t ldx, int vec_in, float* __restrict__ y, int64_t ldy,
    int vec_out, float* __restrict__ kappa_out, int32_t* __restrict__ active_out,
    int32_t* __restrict__ nan_flag, const SplitMapper mp) {
  mfma_split_fwd_body<NKK, TRACK, STAGED, NKX>(Wb, items, n_items, packs, y0, identity, k, n, x, B, ldx, vec_in, y, ldy,
                                               vec_out, kappa_out, active_out, nan_flag, mp);
}

// Wm [n, ldw] (row-major fp32, torch.nn.Linear.weight) -> the split-operand fragment image the mapped kernel reads:
// chunk (row tile tp, K-step s, piece) = 64 lanes x 8 bf16, element i of lane l = Wm[32 tp + (l & 31)][16 s + 8 (i >> 2)
// + 4 (l >> 5) + (i & 3)], zero beyond (n, in_dim); then the bias, zero-padded to n_pad floats.
__global__ void mapper_image_kernel(const float* __restrict__ w, int64_t ldw, const float* __restrict__ bias, int n,
                                    int in_dim, int nkk, int nsx, bf16x8* __restrict__ img) {
  const int chunk = blockIdx.x;            // (tp, s)
  const int tp = chunk / nsx, sx = chunk - tp * nsx;
  const int l = threadIdx.x;
  if (l < 64) {
    const int row = 32 * tp + (l & 31);
    bf16x8 o1, o2, o3;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int col = 16 * sx + 8 * (i >> 2) + 4 * (l >> 5) + (i & 3);
      const float x = (row < n && col < in_dim) ? w[(int64_t)row * ldw + col] : 0.f;
      __bf16 p1, p2, p3;
      split3(x, p1, p2, p3);
      o1[i] = p1;
      o2[i] = p2;
      o3[i] = p3;
    }
    bf16x8* dst = img + (size_t)chunk * 3 * 64 + l;
    dst[0] = o1;
    dst[64] = o2;
    dst[128] = o3;
  }
  if (chunk == 0) {
    float* b = reinterpret_cast<float*>(img + (size_t)nkk * nsx * 3 * 64);
    for (int i = l; i < nkk * 32; i += blockDim.x) b[i] = (bias != nullptr && i < n) ? bias[i] : 0.f;
  }
}

// ---------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------
bool mfma_split_eligible(const RayenPack* p) {
  if (p->n > 64) return false;
  if (p->segs.size() >= 2048 || p->n_rows >= (1 << 20)) return false;  // (segment, row) share one register
  for (const RayenSegment& g : p->segs)
    if (g.type == RAYEN_SEG_LMI) return false;
  TileLayout b(p->n);
  if (layout_tiles(p, b, /*allow_pack=*/true, /*allow_sym=*/false) != RAYEN_OK || b.items.empty()) return false;
  const int64_t padded = (int64_t)b.items.size() * 32;
  return b.useful_rows * 2 >= padded && p->n * 2 >= b.n_pad;
}

void mfma_split_free(SplitImage* img) {
  if (img == nullptr) return;
  if (img->Wb) (void)hipFree(img->Wb);
  if (img->items) (void)hipFree(img->items);
  if (img->packs) (void)hipFree(img->packs);
  if (img->y0) (void)hipFree(img->y0);
  delete img;
}

int mfma_split_build(const RayenPack* p, SplitImage** out, int64_t* bytes) {
  TileLayout b(p->n);
  const int rc = layout_tiles(p, b, /*allow_pack=*/true, /*allow_sym=*/false);
  if (rc != RAYEN_OK) return rc;
  const int n_items = (int)b.items.size();
  const std::vector<float> frag = b.fragments_f32();
  if (b.packs.empty()) b.packs.push_back(MPack());

  SplitImage* img = new SplitImage();
  img->nkk = b.n_pad / 32;
  img->identity = p->out_identity;
  img->n_items = n_items;

  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, p->device) == hipSuccess && prop.multiProcessorCount > 0)
      img->n_simd = prop.multiProcessorCount * 4;
  }
  // three bf16 pieces of every entry, in the fragment order of v_mfma_f32_32x32x16_bf16:
  // chunk (tile, k-step s, piece) = 64 lanes x 8 elements, element i of lane l = column
  // 16 s + 8 (i >> 2) + 4 (l >> 5) + (i & 3) of row l & 31 = entry [2 s + (i >> 2)][l][i & 3] of the fp32 image
  auto rne = [](float x) -> uint16_t {
    uint32_t u;
    std::memcpy(&u, &x, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
  };
  auto widen = [](uint16_t h) -> float {
    const uint32_t u = (uint32_t)h << 16;
    float x;
    std::memcpy(&x, &u, 4);
    return x;
  };
  const int n_tiles = b.n_tiles(), ns = b.nq() / 2;
  std::vector<uint16_t> wb((size_t)n_tiles * ns * 3 * 64 * 8);
  for (int t = 0; t < n_tiles; ++t)
    for (int sp = 0; sp < ns; ++sp)
      for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 8; ++i) {
          const float x = frag[(((size_t)t * b.nq() + 2 * sp + (i >> 2)) * 64 + l) * 4 + (i & 3)];
          const uint16_t h1 = rne(x);
          const float r1 = x - widen(h1);
          const uint16_t h2 = rne(r1);
          const float r2 = r1 - widen(h2);
          const uint16_t h3 = rne(r2);
          const size_t base = (((size_t)t * ns + sp) * 3) * 64 * 8 + (size_t)l * 8 + i;
          wb[base] = h1;
          wb[base + 64 * 8] = h2;
          wb[base + 2 * 64 * 8] = h3;
        }
  const int k_tiles = (p->k + 31) / 32;
  std::vector<float> y0((size_t)k_tiles * 32 + 32, 0.f);
  for (int i = 0; i < p->k; ++i) y0[i] = (float)p->y0[i];
  const bool ok =
      hipMalloc(&img->Wb, wb.size() * 2) == hipSuccess &&
      hipMemcpy(img->Wb, wb.data(), wb.size() * 2, hipMemcpyHostToDevice) == hipSuccess &&
      hipMalloc(&img->y0, y0.size() * sizeof(float)) == hipSuccess &&
      hipMemcpy(img->y0, y0.data(), y0.size() * sizeof(float), hipMemcpyHostToDevice) == hipSuccess &&
      hipMalloc(&img->items, b.items.size() * sizeof(MItem)) == hipSuccess &&
      hipMemcpy(img->items, b.items.data(), b.items.size() * sizeof(MItem), hipMemcpyHostToDevice) == hipSuccess &&
      hipMalloc(&img->packs, b.packs.size() * sizeof(MPack)) == hipSuccess &&
      hipMemcpy(img->packs, b.packs.data(), b.packs.size() * sizeof(MPack), hipMemcpyHostToDevice) == hipSuccess;
  if (!ok) { mfma_split_free(img); return RAYEN_E_ALLOC; }
  img->bytes = (int64_t)(wb.size() * 2 + y0.size() * sizeof(float) + b.items.size() * sizeof(MItem) +
                         b.packs.size() * sizeof(MPack));
  *bytes = img->bytes;
  *out = img;
  return RAYEN_OK;
}

template <int NKK>
static int launch_split(const RayenPack* p, const SplitImage* img, const float* v, int64_t B, int64_t ldv,
                        float* y, int64_t ldy, float* kappa, int32_t* active, int32_t* nan_flag,
                        hipStream_t stream) {
  constexpr int per_wave = 64;
  const int64_t n_groups = (B + per_wave - 1) / per_wave;
  const int64_t slots = (int64_t)launch_simds(img->n_simd) * kMfmaWavesPerSimd;
  const int64_t rounds = (n_groups + slots - 1) / slots;
  const int64_t waves = (n_groups + rounds - 1) / rounds;
  const int64_t grid = (waves + kMfmaWaves - 1) / kMfmaWaves;
  // bit 0: rows are 16-byte aligned (float4 pieces of a row) | bit 1: rows are stored back to back and the base is 16-byte aligned
  const int vec_in = (((ldv % 4 == 0) && ((reinterpret_cast<uintptr_t>(v) & 15) == 0)) ? 1 : 0) |
                     ((ldv == p->n && (reinterpret_cast<uintptr_t>(v) & 15) == 0) ? 2 : 0);
  const int vec_out = (ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0);
  auto go = [&](auto kern) {
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kMfmaWaves * 64), 0, stream,
                       static_cast<const bf16x8*>(img->Wb), img->items, img->n_items, img->packs, img->y0,
                       img->identity, p->k, p->n, v, B, ldv, vec_in, y, ldy, vec_out, kappa, active, nan_flag);
  };
  if (img->identity) {
    if (active != nullptr) go(mfma_split_fwd_kernel<NKK, true, false>);
    else go(mfma_split_fwd_kernel<NKK, false, false>);
  } else {
    if (active != nullptr) go(mfma_split_fwd_kernel<NKK, true, true>);
    else go(mfma_split_fwd_kernel<NKK, false, true>);
  }
  return hipGetLastError() == hipSuccess ? RAYEN_OK : RAYEN_E_LAUNCH;
}

// ---- fused mapper: in_dim <= n_pad columns of x (the transposition patch and the register budget are the walk's)
int64_t mfma_split_mapper_image_bytes(const RayenPack* p, const SplitImage* img, int in_dim) {
  (void)p;
  // (sets with equality constraints, NA_E != I, are served too since round 3: the device fault of their mapped
  // instances was an SGPR restored by v_readlane right in front of an inline-asm VMEM instruction -- five wait states
  // hipcc cannot insert into an asm string; RAYEN_ASM_BASE_COPY)
  if (img == nullptr || in_dim < 1 || in_dim > img->nkk * 32) return 0;
  const int nsx = (in_dim + 31) / 32 * 2;
  return (int64_t)img->nkk * nsx * 3 * 1024 + (int64_t)img->nkk * 32 * sizeof(float);
}

int mfma_split_mapper_prepare(const RayenPack* p, const SplitImage* img, const float* w, int64_t ldw, int in_dim,
                              const float* bias, void* image, hipStream_t stream) {
  if (mfma_split_mapper_image_bytes(p, img, in_dim) == 0) return RAYEN_E_UNSUPPORTED;
  const int nsx = (in_dim + 31) / 32 * 2;
  hipLaunchKernelGGL(mapper_image_kernel, dim3((unsigned)(img->nkk * nsx)), dim3(64), 0, stream, w, ldw, bias, p->n,
                     in_dim, img->nkk, nsx, static_cast<bf16x8*>(image));
  return hipGetLastError() == hipSuccess ? RAYEN_OK : RAYEN_E_LAUNCH;
}

template <int NKK, int NKX>
static int launch_split_map(const RayenPack* p, const SplitImage* img, const float* x, int64_t B, int64_t ldx,
                            const SplitMapper& mp, float* y, int64_t ldy, float* kappa, int32_t* active,
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
                       static_cast<const bf16x8*>(img->Wb), img->items, img->n_items, img->packs, img->y0,
                       img->identity, p->k, p->n, x, B, ldx, vec_in, y, ldy, vec_out, kappa, active, nan_flag, mp);
  };
  if (img->identity) {
    if (active != nullptr) go(mfma_split_map_kernel<NKK, true, false, NKX>);
    else go(mfma_split_map_kernel<NKK, false, false, NKX>);
  } else {
    if (active != nullptr) go(mfma_split_map_kernel<NKK, true, true, NKX>);
    else go(mfma_split_map_kernel<NKK, false, true, NKX>);
  }
  return hipGetLastError() == hipSuccess ? RAYEN_OK : RAYEN_E_LAUNCH;
}

int mfma_split_forward_mapped(const RayenPack* p, const SplitImage* img, const float* x, int64_t B, int64_t ldx,
                              int in_dim, const void* image, float* v_out, int64_t ldvo, float* y, int64_t ldy,
                              float* kappa, int32_t* active, int32_t* nan_flag, hipStream_t stream) {
  if (mfma_split_mapper_image_bytes(p, img, in_dim) == 0 || image == nullptr) return RAYEN_E_UNSUPPORTED;
  if (B == 0) return RAYEN_OK;
  SplitMapper mp;
  mp.img = static_cast<const bf16x8*>(image);
  mp.in_dim = in_dim;
  mp.v_out = v_out;
  mp.ldvo = ldvo;
  const int nkx = (in_dim + 31) / 32;
  if (img->nkk == 1 && nkx == 1) return launch_split_map<1, 1>(p, img, x, B, ldx, mp, y, ldy, kappa, active, nan_flag, stream);
  if (img->nkk == 2 && nkx == 1) return launch_split_map<2, 1>(p, img, x, B, ldx, mp, y, ldy, kappa, active, nan_flag, stream);
  if (img->nkk == 2 && nkx == 2) return launch_split_map<2, 2>(p, img, x, B, ldx, mp, y, ldy, kappa, active, nan_flag, stream);
  return RAYEN_E_UNSUPPORTED;
}

int mfma_split_forward(const RayenPack* p, const SplitImage* img, const float* v, int64_t B, int64_t ldv,
                       float* y, int64_t ldy, float* kappa, int32_t* active, int32_t* nan_flag,
                       hipStream_t stream) {
  if (B == 0) return RAYEN_OK;
  if (img->nkk == 1) return launch_split<1>(p, img, v, B, ldv, y, ldy, kappa, active, nan_flag, stream);
  if (img->nkk == 2) return launch_split<2>(p, img, v, B, ldv, y, ldy, kappa, active, nan_flag, stream);
  return RAYEN_E_UNSUPPORTED;
}

}  // namespace rayen
