// fp32 backward of the projection for sets of DENSE quadratic / cone forms at n = k = 64 (config 3), the forward's structure
// (rayen_mfma_pair_wl.hip, round 6) instead of the bucketed exact-fp32 walk of rayen_mfma_bwd.hip:
//
//   grad_v = s g - [kappa > 1] s^2 (g . v) grad kappa(v),        s = 1 / max(1, kappa)
//
// which is what autograd produces for rayen/constraint_module.py:351-474 (max -> arg-max, relu); grad kappa belongs to the ONE
// constraint that set kappa (`active`, recorded by the forward): a linear row D_i (:353), phi + S v / sqrt(v'S v) of a
// quadratic (:374), the implicit derivative of the root of a cone (:383-399, 339-348) -- rayen_mfma_bwd.hip's header.
//
// What rayen_mfma_bwd.hip spends its 0.091 ms on at config 3 (B = 262 144; count 5.7 + scatter 8.5 + walk 76.5 us): the
// exact-fp32 MFMA is slow enough (64 clocks per K = 2) that the samples are first SORTED by active constraint, and the walk
// of a sorted group is a chain of dependent round trips (index -> gathered rows -> kappa / active -> rows of W -> scattered
// store) on two waves per SIMD with ~120 spilled registers.  Here
//   * S_s v of EVERY dense form is evaluated for every sample on v_mfma_f32_32x32x16_f16 with f16-pair operands (three
//     products per fp32 product, fp32-grade: DESIGN.md 4.0) -- twelve tiles = 144 MFMAs of 32 clocks per 32 samples, less
//     matrix time than the forward -- so nothing is sorted: the batch is streamed in order, rows move as whole lines;
//   * the image of the forms (config 3: 12 tiles = 96 KiB), the linear rows (gathered per lane by the arg-max record) and
//     the aux rows (phi | c, M'beta) are copied into LDS once per workgroup: the walk has no vector-memory instruction
//     (a partner's MFMAs keep those from issuing, scripts/ubench/mfma_coissue.hip);
//   * 32-sample groups dealt to the waves of a workgroup on demand.
// ONE launch.  Accepted per pack by the creation-time measurement against the fp64 lane backward, next to the exact kernel
// (rayen_abi.hip::bwd32_selfcheck).  RAYEN_old's head, other shapes and small batches stay on rayen_mfma_bwd.hip.
#include "rayen_bwd_tiles.h"
#include "rayen_split_image.h"

#include <cstdlib>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>
#include <vector>

namespace rayen {

struct MfmaBwddImage {
  f16x8* Sh = nullptr;     // [n_tiles][4 K-steps][2 pieces][64] x 8 f16: pairs of gS S, MFMA fragment order
  BItem* items = nullptr;  // [n_tiles] (the exact kernel's list without its padding)
  float* Wrow = nullptr;   // [n_rows + 2][64] fp32 rows of W (linear rows, aux rows)
  int n_tiles = 0, n_dense = 0;
  int lin_lo = 0, lin_n = 0;   // the range of W rows that holds every linear segment (copied into LDS)
  int n_simd = 1024;
  float s_inv = 1.f;       // 1 / gS
  int lds_bytes = 0;
  bool ready = false;      // the kernel was promised its dynamic LDS
  int64_t bytes = 0;
};

namespace {
#ifndef RAYEN_BWDD_WAVES
#define RAYEN_BWDD_WAVES 8
#endif
constexpr int kBdWaves = RAYEN_BWDD_WAVES;
// (measured and not kept, gpurun_out/r06zq: a light-register instance for three waves per SIMD -- A operands read at the top of
// their item, the rows of g read a second time for the final combination -- 36 spilled registers at 168, 0.093 against 0.084 ms)
constexpr int kBdStage = 2048;     // a wave's staging bytes: 16 rows x one 128-byte line
}  // namespace

// developer build (-DRAYEN_BWDD_STAMPS; scripts/ubench/bwdd_stamps.py): s_memtime at the phases of the SECOND group of waves 0 and 4
// of workgroup 0.  Nothing in the library build.
#ifdef RAYEN_BWDD_STAMPS
__device__ unsigned long long bwdd_stamp_buf[2 * 32];
extern "C" int rayen_debug_bwdd_stamps(void* dst, size_t bytes) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(bwdd_stamp_buf), bytes < sizeof(bwdd_stamp_buf) ? bytes : sizeof(bwdd_stamp_buf)) == hipSuccess ? 0 : -1;
}
#define RAYEN_BD_STAMP(slot) do { if (stamp_on && lane == 0) bwdd_stamp_buf[(wave >> 2) * 32 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define RAYEN_BD_STAMP(slot) do { } while (0)
#endif

template <bool DUMMY>
__global__ __launch_bounds__(kBdWaves * 64, 1) void mfma_bwdd_kernel(
    const f16x8* __restrict__ Sh, const BItem* __restrict__ items, int n_tiles, int n_dense,
    const float* __restrict__ Wrow, int lin_lo, int lin_n, const float* __restrict__ v, int64_t B, int64_t ldv,
    const float* __restrict__ kappa, const int32_t* __restrict__ active, const float* __restrict__ gy, int64_t ldg,
    float* __restrict__ gv, int64_t ldgv, const float s_inv, const int n) {
  constexpr int NS = 4, NCH = 8, NQ = 8, NP = 64, SR = 16;
  extern __shared__ __attribute__((aligned(1024))) char bd_smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int col = lane & 31;
  const int hi = lane >> 5;
  char* const simg = bd_smem;                                                       // the forms
  char* const lin = bd_smem + (size_t)n_tiles * (NCH * 1024);                      // [lin_n][16 pieces], piece p of row r in slot p ^ (r & 15)
  float* const auxr = reinterpret_cast<float*>(lin + (size_t)lin_n * 256);         // [n_dense][2][64]: phi | c, M'beta of form d
  char* const stage = reinterpret_cast<char*>(auxr + (size_t)n_dense * 128) + wave * kBdStage;
  unsigned* const take_lds = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(auxr + (size_t)n_dense * 128) + kBdWaves * kBdStage);
  float* const ftab = reinterpret_cast<float*>(take_lds + 4);                      // [n_dense][4]: type | tau | a' | - of form d
  const int64_t n_groups = (B + 31) / 32;
  const int64_t grp_stride = (int64_t)gridDim.x;
  int64_t grp = (int64_t)blockIdx.x + (int64_t)wave * grp_stride;

#ifdef RAYEN_BWDD_STAMPS
  if (blockIdx.x == 0 && (wave == 0 || wave == 4) && lane == 0) {
    bwdd_stamp_buf[(wave >> 2) * 32 + 8] = __builtin_amdgcn_s_memtime();
    bwdd_stamp_buf[(wave >> 2) * 32 + 9] = __builtin_amdgcn_s_memrealtime();
  }
#endif
  // ---- once per workgroup: forms, linear rows, aux rows -> LDS
  {
    const int n_chunks = n_tiles * NCH;
    const char* src = reinterpret_cast<const char*>(Sh) + lane * 16;
#pragma unroll 4
    for (int c = wave; c < n_chunks; c += kBdWaves)
      *reinterpret_cast<u32x4*>(simg + c * 1024 + lane * 16) = *reinterpret_cast<const u32x4*>(src + (size_t)c * 1024);
    for (int i = threadIdx.x; i < lin_n * 16; i += kBdWaves * 64) {
      const int r = i >> 4, pc = i & 15;
      *reinterpret_cast<f32x4*>(lin + r * 256 + ((pc ^ (r & 15)) * 16)) =
          *reinterpret_cast<const f32x4*>(Wrow + (size_t)(lin_lo + r) * NP + 4 * pc);
    }
    for (int i = threadIdx.x; i < n_dense * 128; i += kBdWaves * 64)
      auxr[i] = Wrow[(size_t)items[(i >> 7) * 2].aux_row * NP + (i & 127)];     // (rows aux_row, aux_row + 1 are contiguous)
    if (threadIdx.x == 0) *take_lds = kBdWaves;
    if (threadIdx.x < n_dense) {
      const BItem it0 = items[threadIdx.x * 2];
      ftab[threadIdx.x * 4 + 0] = __builtin_bit_cast(float, it0.type);
      ftab[threadIdx.x * 4 + 1] = it0.f0;
      ftab[threadIdx.x * 4 + 2] = it0.f1;
      ftab[threadIdx.x * 4 + 3] = 0.f;
    }
  }
  __syncthreads();  // the only workgroup barrier
#ifdef RAYEN_BWDD_STAMPS
  if (blockIdx.x == 0 && (wave == 0 || wave == 4) && lane == 0) bwdd_stamp_buf[(wave >> 2) * 32 + 10] = __builtin_amdgcn_s_memtime();
#endif

  // this lane's pieces 2 q + hi of its sample's rows of v and g (zero beyond the batch), kappa and the arg-max record.  A
  // group's rows are requested at the END of the previous group, in front of its staged stores: by then nothing but the
  // finished gradient is live, and the round trip runs under the stores and the next group's first instructions
  float vr[32], gr[32], kap_in = 0.f;
  int aseg_in = -1, arow_in = 0;
  // Buffer addressing (rayen_mfma_pair_wl.hip, round 6): descriptors whose extent is the batch, the lane's byte offset inside
  // a group a constant, one uniform term per group; rows beyond the batch are out of range -- loads return 0, stores are dropped
  const __amdgpu_buffer_rsrc_t v_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(v), 0, (int)(unsigned)((uint64_t)B * (uint64_t)ldv * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t g_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gy), 0, (int)(unsigned)((uint64_t)B * (uint64_t)ldg * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t o_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(gv, 0, (int)(unsigned)((uint64_t)B * (uint64_t)ldgv * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t k_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(kappa), 0, (int)(unsigned)((uint64_t)B * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t a_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(active), 0, (int)(unsigned)((uint64_t)B * 8u), 0x00020000);
  const unsigned v_lane_off = (unsigned)col * (unsigned)ldv * 4u + 16u * (unsigned)hi;
  const unsigned g_lane_off = (unsigned)col * (unsigned)ldg * 4u + 16u * (unsigned)hi;
  const unsigned o_lane_off = (unsigned)(lane >> 3) * (unsigned)ldgv * 4u + 16u * (unsigned)((lane & 7) ^ ((lane >> 3) & 7));
  auto request = [&](const int64_t g_) {
    const unsigned g32 = (unsigned)g_ * 32u;
    const unsigned voff = v_lane_off + g32 * (unsigned)ldv * 4u, goff = g_lane_off + g32 * (unsigned)ldg * 4u;
    const bool live = g_ * 32 + col < B;
    // (n = k below 64, in whole 16-byte pieces: the pieces beyond a row's n columns are sent out of range -- loads return 0, stores are
    // dropped; the forms and the linear rows are zero there.  Behind a wave-uniform branch, the empty statement keeps the copies apart.)
    if (__builtin_expect(n < NP, 0)) {
      asm volatile("; ragged width" ::: "memory");
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const bool in = 8 * q + 4 * hi < n;
        const f32x4 x = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(v_rsrc, in ? voff + 32u * q : 0xFFFFFFF0u, 0, 0));
        const f32x4 g = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(g_rsrc, in ? goff + 32u * q : 0xFFFFFFF0u, 0, 0));
#pragma unroll
        for (int c = 0; c < 4; ++c) { vr[4 * q + c] = x[c]; gr[4 * q + c] = g[c]; }
      }
    } else {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const f32x4 x = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(v_rsrc, voff + 32u * q, 0, 0));
        const f32x4 g = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(g_rsrc, goff + 32u * q, 0, 0));
#pragma unroll
        for (int c = 0; c < 4; ++c) { vr[4 * q + c] = x[c]; gr[4 * q + c] = g[c]; }
      }
    }
    const unsigned roff = (g32 + (unsigned)col) * 4u;
    kap_in = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(k_rsrc, roff, 0, 0));
    const int a0 = (int)__builtin_amdgcn_raw_buffer_load_b32(a_rsrc, 2u * roff, 0, 0);
    const int a1 = (int)__builtin_amdgcn_raw_buffer_load_b32(a_rsrc, 2u * roff + 4u, 0, 0);
    aseg_in = live ? a0 : -1;
    arow_in = a1;
  };
  if (grp < n_groups) request(grp);

#ifdef RAYEN_BWDD_STAMPS
  int stamp_round = 0;
  bool stamp_on = false;
#endif
  while (grp < n_groups) {
#ifdef RAYEN_BWDD_STAMPS
    stamp_on = blockIdx.x == 0 && (wave == 0 || wave == 4) && stamp_round == 1;
    ++stamp_round;
#endif
    RAYEN_BD_STAMP(0);
    const int64_t row = grp * 32 + col;
    const bool live = row < B;
    int taken = 0;
    if (lane == 0) taken = (int)atomicAdd(take_lds, 1u);
    const int64_t next = (int64_t)blockIdx.x + (int64_t)__builtin_amdgcn_readfirstlane(taken) * grp_stride;
    const float kap = kap_in;
    const int aseg = aseg_in, arow = arow_in;
    const bool clipped = live && kap > 1.f && aseg >= 0;
    const float sc = 1.f / fmaxf(1.f, kap);
    float tv;
    {
      float d4[4] = {0.f, 0.f, 0.f, 0.f};      // (four chains: one chain of 32 dependent FMAs is 32 x the FMA latency)
#pragma unroll
      for (int i = 0; i < 32; ++i) d4[i & 3] = fmaf(gr[i], vr[i], d4[i & 3]);
      const float dot = (d4[0] + d4[1]) + (d4[2] + d4[3]);
      tv = dot + xhalf(dot);
    }
    RAYEN_BD_STAMP(1);     // rows have landed, g . v
    float ur[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) ur[i] = 0.f;

    if (__ballot(clipped) != 0) {  // wave-uniform: a wave of interior samples skips the walk
      // ---- v -> scaled f16 pairs (the forward's split: per-row power of two)
      f16x8 vb[2][NS];
      float v_inv;
      {
        float m = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) m = fmaxf(m, __builtin_fabsf(vr[i]));
        m = fmaxf(m, xhalf(m));
        float sv;
        int sv_exp;
        pow2_scale(m, sv, v_inv, sv_exp);
#pragma unroll
        for (int sp = 0; sp < NS; ++sp) {
          u32x4 w1, w2;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int q = 2 * sp + (j >> 1), c = 2 * (j & 1);
            unsigned a, b;
            pair_split_lo(a, b, vr[4 * q + c], sv);
            pair_split_hi(a, b, vr[4 * q + c + 1], sv);
            w1[j] = a;
            w2[j] = b;
          }
          vb[0][sp] = __builtin_bit_cast(f16x8, w1);
          vb[1][sp] = __builtin_bit_cast(f16x8, w2);
        }
      }
      RAYEN_BD_STAMP(2);   // split
      float part = 0.f, tot = 0.f;
      int form = -1;       // the dense form this lane's active constraint is (-1: a linear row, or not clipped)
      f32x16 acc;
      const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      u32x4 abuf[NCH];
      auto fetch = [&](const int tile) {
        const char* tb = simg + (size_t)tile * (NCH * 1024) + lane * 16;
#pragma unroll
        for (int c = 0; c < NCH; ++c) abuf[c] = *reinterpret_cast<const u32x4*>(tb + c * 1024);
      };
      fetch(0);
      for (int it = 0; it < n_tiles; ++it) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int sp = 0; sp < NS; ++sp) {
          const f16x8 a1 = __builtin_bit_cast(f16x8, abuf[2 * sp + 0]), a2 = __builtin_bit_cast(f16x8, abuf[2 * sp + 1]);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, vb[0][sp], sp == 0 ? zero : acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, vb[1][sp], acc, 0, 0, 0);
        }
#pragma unroll
        for (int sp = 0; sp < NS; ++sp) {
          const f16x8 a1 = __builtin_bit_cast(f16x8, abuf[2 * sp + 0]);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, vb[0][sp], acc, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        const BItem item = items[it];      // (requested BEHIND the MFMAs: LDS and scalar loads share a counter)
        if (it + 1 < n_tiles) fetch(it + 1);
        __builtin_amdgcn_sched_barrier(0);
        const bool sel = clipped && aseg == item.seg;
        if (__ballot(sel) == 0) continue;          // nobody in this wave sits on this form
        // acc = gS sv (S v)[32 tp + rows of this lane]: the sum v'S v and, for the lanes of this form, the vector itself
        {
          float s4[4] = {(item.flags & MF_FIRST) ? 0.f : part, 0.f, 0.f, 0.f};
          if (item.tp == 0) {
#pragma unroll
            for (int g = 0; g < 16; ++g) {
              s4[g & 3] = fmaf(acc[g], vr[g], s4[g & 3]);
              ur[g] = sel ? acc[g] : ur[g];
            }
          } else {
#pragma unroll
            for (int g = 0; g < 16; ++g) {
              s4[g & 3] = fmaf(acc[g], vr[16 + g], s4[g & 3]);
              ur[16 + g] = sel ? acc[g] : ur[16 + g];
            }
          }
          part = (s4[0] + s4[1]) + (s4[2] + s4[3]);
        }
        if (item.flags & MF_LAST) {
          // the form's lanes only RECORD what their closer needs (v'S v and which form): the closers themselves run once per
          // group, below, every lane on its own form -- six forms closing one after the other on all 32 lanes were 1 200 of
          // the ~2 000 vector instructions of a group
          const float total = part + xhalf(part);
          tot = sel ? total : tot;
          form = sel ? (it >> 1) : form;
        }
      }
      RAYEN_BD_STAMP(3);   // walk
      // ---- grad kappa of every lane's own constraint, in one pass: u = cw (S v) + c0 x0 + c1 x1 with
      //   quadratic   x0 = phi,  cw = 1 / sqrt(v'S v), c0 = 1, c1 = 0                                    (:374)
      //   cone        x0 = c, x1 = M'beta: the implicit derivative of the root (rayen_mfma_bwd.hip)      (:383-399)
      //   linear row  x0 = D_i, cw = 0, c0 = 1, c1 = 0                                                   (:353)
      // x0 | x1 are rows in LDS at a per-lane address (the form's aux rows, or the arg-max row of the linear block, whose
      // pieces sit in slots p ^ (r & 15)); lanes whose linear row is outside the LDS block gather it from memory.
      {
        const bool dense = form >= 0;
        const int r = arow - lin_lo;
        const bool lin_lds = !dense && r >= 0 && r < lin_n;
        const f32x4 ft = *reinterpret_cast<const f32x4*>(ftab + (dense ? form : 0) * 4);     // type | tau | a' | -
        const bool soc = dense && __builtin_bit_cast(int, ft[0]) == BI_SOC;
        const char* base = dense ? reinterpret_cast<const char*>(auxr + (size_t)form * 128) : lin + (lin_lds ? r : 0) * 256;
        const int sw = dense ? 0 : (r & 15);
        float c4[4] = {0.f, 0.f, 0.f, 0.f}, b4[4] = {0.f, 0.f, 0.f, 0.f};
        f32x4 x0[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          x0[q] = *reinterpret_cast<const f32x4*>(base + (((2 * q + hi) ^ sw) * 16));
          const f32x4 x1 = *reinterpret_cast<const f32x4*>(base + 256 + (2 * q + hi) * 16);      // (read by every lane, used by cones)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            c4[c] = fmaf(x0[q][c], vr[4 * q + c], c4[c]);
            b4[c] = fmaf(x1[c], vr[4 * q + c], b4[c]);
          }
        }
        if (clipped && !dense && !lin_lds) {      // (a linear row beyond the block in LDS)
          const float* rw = Wrow + (int64_t)arow * NP + 4 * hi;
#pragma unroll
          for (int q = 0; q < NQ; ++q) x0[q] = *reinterpret_cast<const f32x4*>(rw + 8 * q);
        }
        const float cr = (c4[0] + c4[1]) + (c4[2] + c4[3]), br = (b4[0] + b4[1]) + (b4[2] + b4[3]);
        const float crs = cr + xhalf(cr), brs = br + xhalf(br);
        // quadratic
        const float total = (tot * s_inv) * v_inv;                    // v'S v in natural units
        const float cw_q = total > 0.f ? 1.f / sqrtf(total) : 0.f;
        // cone
        const float tau = ft[1], ap = ft[2];
        const float bp = 2.f * brs - 2.f * crs * tau;
        const float den = 2.f * ap * kap + bp;                        // dF/dkappa at the root
        const float inv = den != 0.f ? -1.f / den : 0.f;
        const float cw_n = dense ? (soc ? 2.f * inv : cw_q) : 0.f;    // d c'/dv = 2 M'Mv - 2 (c.v) c
        const float c0 = soc ? inv * (-2.f * crs - 2.f * tau * kap) : 1.f;
        const float c1 = soc ? inv * 2.f * kap : 0.f;                 // kappa * d b'/dv = kappa (2 M'beta - 2 tau c)
        const float cw = (cw_n * s_inv) * v_inv;                      // (S v sits in ur with the scales gS sv on it)
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const f32x4 x1 = *reinterpret_cast<const f32x4*>(base + 256 + (2 * q + hi) * 16);
#pragma unroll
          for (int c = 0; c < 4; ++c) ur[4 * q + c] = fmaf(cw, ur[4 * q + c], fmaf(c0, x0[q][c], c1 * x1[c]));
        }
      }
    }

    RAYEN_BD_STAMP(4);     // closers
    // ---- grad_v = s g - coef grad kappa, out as whole 128-byte lines through the wave's 2 KiB of LDS (rayen_mfma_pair_wl.hip)
    {
      const float coef = clipped ? sc * sc * tv : 0.f;
#pragma unroll
      for (int i = 0; i < 32; ++i) ur[i] = fmaf(sc, gr[i], -coef * ur[i]);
      __builtin_amdgcn_sched_barrier(0);
      RAYEN_BD_STAMP(5);   // combined
      if (next < n_groups) request(next);      // (v, g, kappa and the record of this group are dead)
      __builtin_amdgcn_sched_barrier(0);
      const unsigned o_goff = o_lane_off + (unsigned)grp * 32u * (unsigned)ldgv * 4u;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int part16 = 0; part16 < 32 / SR; ++part16) {
          if ((col / SR) == part16) {
            char* slot = stage + (col % SR) * 128;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
              const int q = 4 * h + qq;
              *reinterpret_cast<f32x4*>(slot + (((2 * qq + hi) ^ (col & 7)) * 16)) = f32x4{ur[4 * q], ur[4 * q + 1], ur[4 * q + 2], ur[4 * q + 3]};
            }
          }
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int i = 0; i < SR / 8; ++i) {
            const int r = 8 * i + (lane >> 3);
            const f32x4 x = *reinterpret_cast<const f32x4*>(stage + r * 128 + (lane & 7) * 16);
            (void)r;   // (row (lane >> 3) of the eight, slot (lane & 7) ^ (row & 7): o_lane_off)
            unsigned off = o_goff + (unsigned)(part16 * SR + 8 * i) * (unsigned)ldgv * 4u + 128u * h;
            if (__builtin_expect(n < NP, 0)) {
              asm volatile("; ragged width" ::: "memory");
              off = (32 * h + 4 * ((lane & 7) ^ ((lane >> 3) & 7)) < n) ? off : 0xFFFFFFF0u;
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, x), o_rsrc, off, 0, 2);
          }
          __builtin_amdgcn_wave_barrier();
        }
      }
    }
    RAYEN_BD_STAMP(6);     // next rows requested, this group's lines staged and stored
    grp = next;
#ifdef RAYEN_BWDD_STAMPS
    if (blockIdx.x == 0 && (wave == 0 || wave == 4) && lane == 0) bwdd_stamp_buf[(wave >> 2) * 32 + 13] = stamp_round;
#endif
  }
#ifdef RAYEN_BWDD_STAMPS
  if (blockIdx.x == 0 && (wave == 0 || wave == 4) && lane == 0) {
    bwdd_stamp_buf[(wave >> 2) * 32 + 11] = __builtin_amdgcn_s_memtime();
    bwdd_stamp_buf[(wave >> 2) * 32 + 12] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

// ---------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------
bool mfma_bwdd_eligible(const RayenPack* p) {
  // (n = k = 64, or -- round 6 -- anything above 32 in whole 16-byte pieces: the tiles are padded to 64 columns anyway)
  return bwd_tiles_eligible(p) && p->n > 32 && p->n <= 64 && (p->n % 4) == 0 && p->k == p->n && p->out_identity;
}

void mfma_bwdd_free(MfmaBwddImage* img);

int mfma_bwdd_build(const RayenPack* p, MfmaBwddImage** out, int64_t* bytes) {
  const int n = p->n, np = n_pad_of(n);
  TileLayout b(n);
  std::vector<BItem> items;
  (void)layout_bwd_tiles(p, b, items);
  int n_tiles = 0;
  for (const BItem& it : items)
    if (it.type == BI_QUAD || it.type == BI_SOC) ++n_tiles;
  *out = nullptr;
  *bytes = 0;
  if (n_tiles == 0 || (n_tiles & 1)) return RAYEN_OK;   // (two row tiles per form at n = 64; nothing dense: the exact kernel keeps the pack)
  for (int i = 0; i < n_tiles; ++i)
    if (items[i].type != BI_QUAD && items[i].type != BI_SOC) return RAYEN_OK;
  items.resize((size_t)n_tiles);
  const std::vector<float> frag = b.fragments_f32();
  // one global power of two that puts the largest entry of the forms into [2^13, 2^14)
  const int nq = b.nq(), ns = nq / 2;
  float big = 0.f;
  for (size_t i = 0; i < (size_t)n_tiles * nq * 64 * 4; ++i) big = std::fmax(big, std::fabs(frag[i]));
  int ex = 0;
  if (big > 0.f) (void)std::frexp(big, &ex);
  int shift = big > 0.f ? 14 - ex : 0;
  shift = shift > 100 ? 100 : (shift < -100 ? -100 : shift);
  const float s_scale = std::ldexp(1.0f, shift);
  std::vector<_Float16> sh((size_t)n_tiles * ns * 2 * 64 * 8);
  for (int t = 0; t < n_tiles; ++t)
    for (int sp = 0; sp < ns; ++sp)
      for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 8; ++i) {
          const float x = frag[(((size_t)t * nq + 2 * sp + (i >> 2)) * 64 + l) * 4 + (i & 3)] * s_scale;
          const _Float16 h1 = (_Float16)x;
          const _Float16 h2 = (_Float16)(x - (float)h1);
          const size_t base = (((size_t)t * ns + sp) * 2) * 64 * 8 + (size_t)l * 8 + i;
          sh[base] = h1;
          sh[base + 64 * 8] = h2;
        }
  std::vector<float> wrow((size_t)(p->n_rows + 2) * np, 0.f);
  for (int r = 0; r < p->n_rows; ++r)
    for (int j = 0; j < n; ++j) wrow[(size_t)r * np + j] = (float)p->W[(size_t)r * n + j];
  MfmaBwddImage* img = new MfmaBwddImage();
  img->n_tiles = n_tiles;
  img->n_dense = n_tiles / 2;
  img->s_inv = std::ldexp(1.0f, -shift);
  int lo = 1 << 30, hi = -1;
  for (const RayenSegment& g : p->segs)
    if (g.type == RAYEN_SEG_LIN) { lo = std::min(lo, (int)g.row0); hi = std::max(hi, (int)(g.row0 + g.nrows)); }
  img->lin_lo = hi > lo ? lo : 0;
  img->lin_n = hi > lo ? hi - lo : 0;
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, p->device) == hipSuccess && prop.multiProcessorCount > 0)
      img->n_simd = prop.multiProcessorCount * 4;
  }
  auto lds_of = [&](int lin_n) { return n_tiles * 8192 + lin_n * 256 + img->n_dense * 512 + kBdWaves * kBdStage + 16 + img->n_dense * 16; };
  if (lds_of(img->lin_n) > 160 * 1024) img->lin_n = 0;      // (the linear rows stay in L2: gathered from there)
  img->lds_bytes = lds_of(img->lin_n);
  if (img->lds_bytes > 160 * 1024) { delete img; return RAYEN_OK; }
  const bool ok =
      hipMalloc(&img->Sh, sh.size() * 2) == hipSuccess &&
      hipMemcpy(img->Sh, sh.data(), sh.size() * 2, hipMemcpyHostToDevice) == hipSuccess &&
      hipMalloc(&img->items, items.size() * sizeof(BItem)) == hipSuccess &&
      hipMemcpy(img->items, items.data(), items.size() * sizeof(BItem), hipMemcpyHostToDevice) == hipSuccess &&
      hipMalloc(&img->Wrow, wrow.size() * sizeof(float)) == hipSuccess &&
      hipMemcpy(img->Wrow, wrow.data(), wrow.size() * sizeof(float), hipMemcpyHostToDevice) == hipSuccess;
  if (!ok) { mfma_bwdd_free(img); return RAYEN_E_ALLOC; }
  // (the attribute belongs to the kernel instance: every pack asks for the running maximum; pack creation only)
  {
    static std::mutex mu;
    static int promised = 0;
    std::lock_guard<std::mutex> hold(mu);
    promised = std::max(promised, img->lds_bytes);
    img->ready = hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_bwdd_kernel<false>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, promised) == hipSuccess;
  }
  img->bytes = (int64_t)(sh.size() * 2 + items.size() * sizeof(BItem) + wrow.size() * sizeof(float));
  *bytes = img->bytes;
  *out = img;
  return RAYEN_OK;
}

void mfma_bwdd_free(MfmaBwddImage* img) {
  if (img == nullptr) return;
  if (img->Sh) (void)hipFree(img->Sh);
  if (img->items) (void)hipFree(img->items);
  if (img->Wrow) (void)hipFree(img->Wrow);
  delete img;
}

bool mfma_bwdd_serves(const RayenPack* p, const MfmaBwddImage* img, const float* v, int64_t B, int64_t ldv,
                      const float* gy, int64_t ldg, const float* gv, int64_t ldgv) {
  (void)p;
  if (img == nullptr || !img->ready) return false;
  auto aligned = [](const void* ptr, int64_t ld) { return (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(ptr) & 15) == 0); };
  if (!aligned(v, ldv) || !aligned(gy, ldg) || !aligned(gv, ldgv)) return false;
  // (buffer addressing with 32-bit byte offsets; the rows of the ragged last group beyond the batch must not wrap)
  auto fits = [&](int64_t ld) { return (uint64_t)(B + 64) * (uint64_t)ld * 4u < (1ull << 32); };
  if (!fits(ldv) || !fits(ldg) || !fits(ldgv)) return false;
  // Every batch size (round 6, gpurun_out/r06zzh; config 3, ms per call): one workgroup per CU as soon as there is a group for it --
  //   B = 1 024: 0.012 (bucketed exact-fp32 walk: 0.072) | 16 384: 0.014 (0.076) | 32 768: 0.015 (0.039) | 65 536: 0.021 (0.043) | 131 072: 0.035 (0.068)
  // (the bucketed walk pays three launches and a pass over the batch per bucket whatever the batch).
  static const int64_t min_groups_env = [] { const char* e = getenv("RAYEN_BWDD_MIN_GROUPS"); return e ? atoll(e) : 1ll; }();   // developer sweeps
  return (B + 31) / 32 >= min_groups_env;
}

int mfma_bwdd_backward(const RayenPack* p, const MfmaBwddImage* img, const float* v, int64_t B, int64_t ldv,
                       const float* kappa, const int32_t* active, const float* grad_y, int64_t ldg, float* grad_v,
                       int64_t ldgv, hipStream_t stream) {
  if (B == 0) return RAYEN_OK;
  if (img == nullptr || !img->ready) return RAYEN_E_UNSUPPORTED;
  const int64_t n_groups = (B + 31) / 32;
  const int64_t cus = launch_simds(img->n_simd) / 4;
  const unsigned grid = (unsigned)std::min<int64_t>(cus, n_groups);   // (every CU as soon as there is a group for it: rayen_mfma_pair_wl.hip)
  hipLaunchKernelGGL((mfma_bwdd_kernel<false>), dim3(grid), dim3(kBdWaves * 64), img->lds_bytes, stream, img->Sh, img->items,
                     img->n_tiles, img->n_dense, img->Wrow, img->lin_lo, img->lin_n, v, B, ldv, kappa, active, grad_y, ldg,
                     grad_v, ldgv, img->s_inv, p->n);
  return hipGetLastError() == hipSuccess ? RAYEN_OK : RAYEN_E_LAUNCH;
}

}  // namespace rayen
