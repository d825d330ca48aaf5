// Split-operand forward, ONE wave per SIMD: the walk of rayen_mfma_split.hip with the epilogue of every tile issued
// INSIDE the MFMA stream of the next one.
//
// What bounds rayen_mfma_split.hip (two waves per SIMD, 256 registers each): its matrix pipe is busy half the time --
// an in-order wave cannot issue its epilogue next to its own MFMAs (the results are not there yet), and the two
// co-resident waves of a SIMD share one issue port: VALU work of either is serial with the MFMA stream of the other
// (rocprofv3: 26 k quad-cycles of MFMA + 11 k of VALU + 14 k idle per SIMD and launch).  A gfx950 SIMD hides up to
// five single-issue instructions behind every 32-cycle v_mfma_f32_32x32x16_bf16 of the SAME wave -- if they are
// independent of it.  So here a wave owns the whole SIMD (512 registers), FOUR sample tiles (every A chunk feeds four
// MFMA chains: half the L2 traffic per sample) and TWO accumulator sets: while the MFMAs of tile i fill one set, the
// epilogue of tile i-1 reduces the other, cut into slices that sit between the MFMA groups of tile i in program
// order (sched_group_barrier pins the interleave).  Same image, same item list, same piece order of the products and
// the same epilogue arithmetic as rayen_mfma_split.hip: the two kernels return identical bits.
//
// Serves packs with NA_E = I (no output tiles) and n <= 64; everything else stays on rayen_mfma_split.hip.
#include "rayen_split_image.h"

namespace rayen {

namespace s4 {

#ifndef RAYEN_S4_NT
#define RAYEN_S4_NT 2
#endif
constexpr int kNT = RAYEN_S4_NT;   // sample tiles per wave
constexpr int kWaves = 4;   // one per SIMD; the workgroup is only a launch unit (no barrier after the prologue)

enum : int { EPI_NONE = 0, EPI_LIN = 1, EPI_AUX = 2, EPI_PACK = 3, EPI_SUMSQ = 4 };

__device__ __forceinline__ void split3(const float x, __bf16& p1, __bf16& p2, __bf16& p3) {
  p1 = (__bf16)x;
  const float r1 = x - (float)p1;
  p2 = (__bf16)r1;
  const float r2 = r1 - (float)p2;
  p3 = (__bf16)r2;
}

// one MFMA followed by `fill` independent VALU / LDS instructions, in this order
#define RAYEN_S4_GROUP(fill)                          \
  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  \
  __builtin_amdgcn_sched_group_barrier(0x002 | 0x100 | 0x200, fill, 0)

template <int NKK, bool TRACK, int NT>
__global__ __launch_bounds__(kWaves * 64, 1) void mfma_split4_fwd_kernel(
    const bf16x8* __restrict__ Wb, const MItem* __restrict__ items, int n_items,
    const MPack* __restrict__ packs, const float* __restrict__ y0, int k, int n,
    const float* __restrict__ v, int64_t B, int64_t ldv, int vec_in, float* __restrict__ y, int64_t ldy,
    int vec_out, float* __restrict__ kappa_out, int32_t* __restrict__ active_out,
    int32_t* __restrict__ nan_flag) {
  constexpr int NS = NKK * 2, NCH = NS * 3, KK = NKK * 16;
  __shared__ float aux_lds[kWaves][NT][32][32];
  __shared__ __attribute__((aligned(16))) float y0_lds[NKK * 32];
  constexpr int LSTR = NKK * 32 + 4;
  __shared__ __attribute__((aligned(16))) float line_lds[kWaves][32][LSTR];

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int col = lane & 31;
  const int hi = lane >> 5;
  const int64_t n_groups = (B + NT * 32 - 1) / (NT * 32);
  const int64_t wave_id = (int64_t)blockIdx.x * kWaves + wave;
  const int64_t wave_stride = (int64_t)gridDim.x * kWaves;
  bool bad = false;
  for (int i = threadIdx.x; i < NKK * 32; i += kWaves * 64) y0_lds[i] = y0[i];
  __syncthreads();  // the only workgroup barrier
  float (*patch)[LSTR] = line_lds[wave];

  // ---- A operands: the rolling register buffer with hand-placed loads of rayen_mfma_split.hip
  u32x4 abuf[NCH];
  unsigned lane_off = lane * 16;   // (not const: captured by the nested lambdas as a register operand)
  auto load_step = [&](const char* tile_base, const int sp) {
    const char* sb = tile_base + sp * 3072;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(abuf[3 * sp + 0]) : "v"(lane_off), "s"(sb));
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "+v"(abuf[3 * sp + 1]) : "v"(lane_off), "s"(sb));
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048" : "+v"(abuf[3 * sp + 2]) : "v"(lane_off), "s"(sb));
  };
#pragma unroll
  for (int c = 0; c < NCH; ++c) abuf[c] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
  for (int sp = 0; sp < NS; ++sp) load_step(reinterpret_cast<const char*>(Wb), sp);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

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
  {
    // two sample tiles at a time through the patch (the fp32 rows are dead once they are split)
#pragma unroll
    for (int h = 0; h < NT; h += 2) {
      float vr[2][KK];
      const bool lv[2] = {live[h], live[h + 1]};
      load_rows<2, NKK, LSTR, true>(vr, v, ldv, n, vec_in & 1, s_base + 32 * h, B, lv, patch, lane);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < NKK * 4; ++q)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            __bf16 p1, p2, p3;
            split3(vr[t][4 * q + c], p1, p2, p3);
            vb[h + t][0][q >> 1][(q & 1) * 4 + c] = p1;
            vb[h + t][1][q >> 1][(q & 1) * 4 + c] = p2;
            vb[h + t][2][q >> 1][(q & 1) * 4 + c] = p3;
          }
    }
  }

  float kap[NT], part[NT], scale[NT];
  int acode[NT];  // (segment << 20) | row, -1 = none
#pragma unroll
  for (int t = 0; t < NT; ++t) { kap[t] = 0.f; part[t] = 0.f; scale[t] = 1.f; acode[t] = -1; }

  // ---- the epilogue of one tile, in NSL slices (slice j = sample tile j & 3, part j >> 2); identical arithmetic,
  // in identical order per sample tile, to rayen_mfma_split.hip.  Every slice is branch-free for its EPI kind.
  constexpr int NSL = 12, NPARTS = NSL / NT;   // slice j = (sample tile j % NT, part j / NT)
  auto epi_slice = [&](auto kind_tag, const MItem& item, const MPack& pk, f32x16 (&acc)[NT], const int j) {
    constexpr int EPI = decltype(kind_tag)::value;
    const int t = j % NT, part_i = j / NT;
    const int g0 = (16 * part_i) / NPARTS, g1 = (16 * (part_i + 1)) / NPARTS;   // this part's share of the 16 registers
    if constexpr (EPI == EPI_LIN) {
      const int lin_code = (item.seg << 20) + item.row0 + 4 * hi;
#pragma unroll
      for (int tt = 0; tt < NT; ++tt)
        if (tt == t) {
#pragma unroll
          for (int g = 0; g < 16; ++g)
            if (g >= g0 && g < g1) {
              if (TRACK) {
                const bool up = acc[tt][g] > kap[tt];
                kap[tt] = up ? acc[tt][g] : kap[tt];
                acode[tt] = up ? lin_code + ((g & 3) + 8 * (g >> 2)) : acode[tt];
              } else {
                kap[tt] = fmaxf(kap[tt], acc[tt][g]);
              }
            }
        }
    } else if constexpr (EPI == EPI_AUX) {
#pragma unroll
      for (int tt = 0; tt < NT; ++tt)
        if (tt == t) {
#pragma unroll
          for (int g = 0; g < 16; ++g)
            if (g >= g0 && g < g1) aux_lds[wave][tt][(g & 3) + 8 * (g >> 2) + 4 * hi][col] = acc[tt][g];
        }
    } else if constexpr (EPI == EPI_PACK) {
      // the four quads of rows, in order: quad a belongs to part (a * NPARTS) / 4 ... spread as evenly as the parts allow
#pragma unroll
      for (int tt = 0; tt < NT; ++tt)
        if (tt == t) {
#pragma unroll
          for (int a = 0; a < 4; ++a) {
            const bool mine = NPARTS >= 4 ? (part_i == a) : ((part_i == 0 && a < 2) || (part_i == 1 && a == 2) || (part_i == 2 && a == 3));
            if (mine) {
              const int slot = hi ? pk.aux[a][1] : pk.aux[a][0];
              const int sid = hi ? pk.seg[a][1] : pk.seg[a][0];
              const bool pair = (item.row0 >> a) & 1;
              float qs = acc[tt][4 * a] * acc[tt][4 * a];
#pragma unroll
              for (int c = 1; c < 4; ++c) qs = fmaf(acc[tt][4 * a + c], acc[tt][4 * a + c], qs);
              const float other = xhalf(qs);
              qs = pair ? qs + other : qs;
              const float kc = aux_lds[wave][tt][slot & 31][col] + __builtin_amdgcn_sqrtf(qs);
              const bool up = sid >= 0 && kc > kap[tt];
              kap[tt] = up ? kc : kap[tt];
              acode[tt] = up ? (sid << 20) : acode[tt];
            }
          }
        }
    } else if constexpr (EPI == EPI_SUMSQ) {
      // QFAC / SOC: running sum of squares in parts 0 and 1 (the two halves of the 16 registers), closed in part 2.
      // (rayen_mfma_split.hip sums the register pairs into (s.x, s.y) and adds the halves at the end of the tile: the
      // y half rides in scale[] between the parts, which nothing reads before the group's end)
#pragma unroll
      for (int tt = 0; tt < NT; ++tt)
        if (tt == t) {
          if (part_i < 2) {
            f32x2 carry = {(part_i == 0) ? ((item.flags & MF_FIRST) ? 0.f : part[tt]) : part[tt],
                           (part_i == 0) ? 0.f : scale[tt]};
#pragma unroll
            for (int g = 0; g < 8; g += 2) {
#pragma unroll
              for (int hh = 0; hh < 2; ++hh)
                if (hh == part_i) {
                  const f32x2 a2 = {acc[tt][8 * hh + g], acc[tt][8 * hh + g + 1]};
                  carry = __builtin_elementwise_fma(a2, a2, carry);
                }
            }
            part[tt] = carry[0];
            scale[tt] = carry[1];
          } else if (part_i == 2) {
            const float summed = part[tt] + scale[tt];
            part[tt] = summed;
            scale[tt] = 1.f;
            const bool last = (item.flags & MF_LAST) != 0;
            const float total = summed + xhalf(summed);
            const float a0 = aux_lds[wave][tt][item.aux & 31][col];
            const float br = aux_lds[wave][tt][(item.aux + 1) & 31][col];
            const float k_quad = a0 + __builtin_amdgcn_sqrtf(fmaxf(total, 0.f));
            // a' x^2 + b' x + c' = 0  (rayen/constraint_module.py:392-396, 339-348), a' < 0
            const float cp = total - a0 * a0;
            const float bp = 2.f * br - 2.f * a0 * item.f0;
            const float disc = bp * bp - 4.f * item.f1 * cp;
            const float root = __builtin_amdgcn_sqrtf(fmaxf(disc, 0.f));
            const float inv2a = 0.5f * __builtin_amdgcn_rcpf(item.f1);
            const float k_soc = disc >= 0.f ? fmaxf((-bp - root) * inv2a, (-bp + root) * inv2a) : 0.f;
            const float kc = item.type == MI_SOC ? k_soc : k_quad;
            const bool up = last && kc > kap[tt];
            kap[tt] = up ? kc : kap[tt];
            acode[tt] = up ? (item.seg << 20) : acode[tt];
          }
        }
    }
  };

  auto load_chunk_at = [&](const char* next_tile, const int idx) {
    const char* sb = next_tile + idx * 1024;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(abuf[idx]) : "v"(lane_off), "s"(sb));
  };
  // ---- the MFMAs of one tile into `acc`, the slices of the previous tile's epilogue (on `prev`) between them
  auto run_tile = [&](auto kind_tag, f32x16 (&acc)[NT], const char* next_tile, const MItem& pitem, const MPack& ppk,
                      f32x16 (&prev)[NT]) {
    constexpr int EPI = decltype(kind_tag)::value;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto load_chunk = [&](const int idx) { load_chunk_at(next_tile, idx); };
    int slice = 0;
    // pass 1: the 2^-16 products of every K-step (12 MFMAs per step at NT = 4)
#pragma unroll
    for (int sp = 0; sp < NS; ++sp) {
      __builtin_amdgcn_sched_barrier(0);
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
      if constexpr (EPI != EPI_NONE) {
        epi_slice(kind_tag, pitem, ppk, prev, slice);
        ++slice;
        if constexpr (NS == 2) { epi_slice(kind_tag, pitem, ppk, prev, slice); ++slice; }
#pragma unroll
        for (int m = 0; m < 3 * NT; ++m) { RAYEN_S4_GROUP(3); }
      }
      __builtin_amdgcn_sched_barrier(0);
      load_chunk(3 * sp + 2);
      __builtin_amdgcn_sched_barrier(0);
    }
    // pass 2: the 2^-8 products (8 MFMAs per step)
#pragma unroll
    for (int sp = 0; sp < NS; ++sp) {
      const bf16x8 a1 = __builtin_bit_cast(bf16x8, abuf[3 * sp + 0]), a2 = __builtin_bit_cast(bf16x8, abuf[3 * sp + 1]);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, vb[t][0][sp], acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, vb[t][1][sp], acc[t], 0, 0, 0);
      if constexpr (EPI != EPI_NONE) {
        epi_slice(kind_tag, pitem, ppk, prev, slice);
        ++slice;
        if constexpr (NS == 2) { epi_slice(kind_tag, pitem, ppk, prev, slice); ++slice; }
#pragma unroll
        for (int m = 0; m < 2 * NT; ++m) { RAYEN_S4_GROUP(4); }
      }
      __builtin_amdgcn_sched_barrier(0);
      load_chunk(3 * sp + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    // pass 3: the leading products (4 MFMAs per step)
#pragma unroll
    for (int sp = 0; sp < NS; ++sp) {
      const bf16x8 a1 = __builtin_bit_cast(bf16x8, abuf[3 * sp + 0]);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, vb[t][0][sp], acc[t], 0, 0, 0);
      if constexpr (EPI != EPI_NONE) {
        epi_slice(kind_tag, pitem, ppk, prev, slice);
        ++slice;
        if constexpr (NS == 2) { epi_slice(kind_tag, pitem, ppk, prev, slice); ++slice; }
#pragma unroll
        for (int m = 0; m < NT; ++m) { RAYEN_S4_GROUP(5); }
      }
      __builtin_amdgcn_sched_barrier(0);
      load_chunk(3 * sp + 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  auto kind_of = [](const MItem& it) {
    return it.type == MI_LIN ? EPI_LIN : it.type == MI_AUX ? EPI_AUX : it.type == MI_PACK ? EPI_PACK
           : (it.type == MI_QFAC || it.type == MI_SOC) ? EPI_SUMSQ : EPI_NONE;
  };
  auto step = [&](f32x16 (&acc)[NT], f32x16 (&prev)[NT], const int it, const MItem& pitem) {
    const char* next_tile = reinterpret_cast<const char*>(Wb) + (size_t)(it + 1 == n_items ? 0 : it + 1) * (NCH * 1024);
    MPack ppk;
    if (pitem.type == MI_PACK) ppk = packs[pitem.aux];
    switch (it == 0 ? EPI_NONE : kind_of(pitem)) {
      case EPI_LIN: run_tile(std::integral_constant<int, EPI_LIN>{}, acc, next_tile, pitem, ppk, prev); break;
      case EPI_AUX:
        run_tile(std::integral_constant<int, EPI_AUX>{}, acc, next_tile, pitem, ppk, prev);
        __builtin_amdgcn_wave_barrier();
        break;
      case EPI_PACK: run_tile(std::integral_constant<int, EPI_PACK>{}, acc, next_tile, pitem, ppk, prev); break;
      case EPI_SUMSQ: run_tile(std::integral_constant<int, EPI_SUMSQ>{}, acc, next_tile, pitem, ppk, prev); break;
      default: run_tile(std::integral_constant<int, EPI_NONE>{}, acc, next_tile, pitem, ppk, prev); break;
    }
  };
  auto tail = [&](f32x16 (&prev)[NT], const MItem& pitem) {   // the last tile's epilogue, nothing to hide behind
    MPack ppk;
    if (pitem.type == MI_PACK) ppk = packs[pitem.aux];
    for (int j = 0; j < NSL; ++j) {
      switch (kind_of(pitem)) {
        case EPI_LIN: epi_slice(std::integral_constant<int, EPI_LIN>{}, pitem, ppk, prev, j); break;
        case EPI_AUX: epi_slice(std::integral_constant<int, EPI_AUX>{}, pitem, ppk, prev, j); break;
        case EPI_PACK: epi_slice(std::integral_constant<int, EPI_PACK>{}, pitem, ppk, prev, j); break;
        case EPI_SUMSQ: epi_slice(std::integral_constant<int, EPI_SUMSQ>{}, pitem, ppk, prev, j); break;
        default: break;
      }
    }
  };

  f32x16 acc_a[NT], acc_b[NT];
  MItem prev_item = items[0];
  for (int it = 0; it < n_items; it += 2) {
    const MItem cur = items[it];
    step(acc_a, acc_b, it, prev_item);
    prev_item = cur;
    if (it + 1 < n_items) {
      const MItem nxt = items[it + 1];
      step(acc_b, acc_a, it + 1, prev_item);
      prev_item = nxt;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next group's first tile has landed
  if (n_items & 1) tail(acc_a, prev_item);
  else tail(acc_b, prev_item);

  // kappa: both halves of the wave
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
  // y = y0 + v / max(1, kappa): v rebuilt from its pieces, v1 + v2 + v3 (exact); two sample tiles at a time
#pragma unroll
  for (int h = 0; h < NT; h += 2) {
    float vr[2][KK];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int sp = 0; sp < NS; ++sp) {
        const u32x4 w1 = __builtin_bit_cast(u32x4, vb[h + t][0][sp]);
        const u32x4 w2 = __builtin_bit_cast(u32x4, vb[h + t][1][sp]);
        const u32x4 w3 = __builtin_bit_cast(u32x4, vb[h + t][2][sp]);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int sh = (i & 1) ? 0 : 16;
          const unsigned m = (i & 1) ? 0xFFFF0000u : 0xFFFFFFFFu;
          const float x1 = __builtin_bit_cast(float, (w1[i >> 1] << sh) & m);
          const float x2 = __builtin_bit_cast(float, (w2[i >> 1] << sh) & m);
          const float x3 = __builtin_bit_cast(float, (w3[i >> 1] << sh) & m);
          vr[t][4 * (2 * sp + (i >> 2)) + (i & 3)] = (x1 + x2) + x3;
        }
      }
    const float sc2[2] = {scale[h], scale[h + 1]};
    const bool lv[2] = {live[h], live[h + 1]};
    bad |= store_rows<2, NKK, LSTR, true>(vr, sc2, y0_lds, y, ldy, k, vec_out, s_base + 32 * h, B, lv, patch, lane);
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

}  // namespace s4

bool mfma_split4_serves(const RayenPack* p, const SplitImage* img) {
  (void)p;
  return img != nullptr && img->identity && (img->nkk == 1 || img->nkk == 2);
}

template <int NKK>
static int launch_split4(const RayenPack* p, const SplitImage* img, const float* v, int64_t B, int64_t ldv, float* y,
                         int64_t ldy, float* kappa, int32_t* active, int32_t* nan_flag, hipStream_t stream) {
  constexpr int per_wave = s4::kNT * 32;
  const int64_t n_groups = (B + per_wave - 1) / per_wave;
  const int64_t slots = (int64_t)img->n_simd;                 // one wave per SIMD
  const int64_t rounds = (n_groups + slots - 1) / slots;
  const int64_t waves = (n_groups + rounds - 1) / rounds;
  const int64_t grid = (waves + s4::kWaves - 1) / s4::kWaves;
  const int vec_in = ((ldv % 4 == 0) && ((reinterpret_cast<uintptr_t>(v) & 15) == 0)) ? 1 : 0;
  const int vec_out = (ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0);
  auto go = [&](auto kern) {
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(s4::kWaves * 64), 0, stream,
                       static_cast<const bf16x8*>(img->Wb), img->items, img->n_items, img->packs, img->y0, p->k, p->n, v,
                       B, ldv, vec_in, y, ldy, vec_out, kappa, active, nan_flag);
  };
  if (active != nullptr) go(s4::mfma_split4_fwd_kernel<NKK, true, s4::kNT>);
  else go(s4::mfma_split4_fwd_kernel<NKK, false, s4::kNT>);
  return hipGetLastError() == hipSuccess ? RAYEN_OK : RAYEN_E_LAUNCH;
}

int mfma_split4_forward(const RayenPack* p, const SplitImage* img, const float* v, int64_t B, int64_t ldv, float* y,
                        int64_t ldy, float* kappa, int32_t* active, int32_t* nan_flag, hipStream_t stream) {
  if (B == 0) return RAYEN_OK;
  if (img->nkk == 1) return launch_split4<1>(p, img, v, B, ldv, y, ldy, kappa, active, nan_flag, stream);
  if (img->nkk == 2) return launch_split4<2>(p, img, v, B, ldv, y, ldy, kappa, active, nan_flag, stream);
  return RAYEN_E_UNSUPPORTED;
}

}  // namespace rayen
