// fp32 backward on f16 pairs for the sets rayen_mfma_bwdg.hip is slowest on: n <= 32, equality constraints allowed
// (k <= 64), every quadratic a small factor (rank <= 8) that sits in packed tiles -- the corridor sets (config 5 / 5r).
//
//   grad_v = s t - [kappa > 1] s^2 (t . v) grad kappa(v),   t = NA_E' g,   s = 1 / max(1, kappa)
//   (rayen/constraint_module.py:351-474 differentiated; what autograd computes through the reference's op chain)
//
// What rayen_mfma_bwdg.hip spends its time on for these shapes (config 5r, B = 262144: 145 us + 20 us of bucket sort
// against a 60 us forward): the exact-fp32 MFMA at 1/16 of the 16-bit rate (t is formed twice to save registers, the
// masked two-step product of a packed tile is 64 instructions of 64 cycles) and -- because that is so slow -- a
// permutation of the batch by active constraint, which turns the 120- and 180-byte rows into scattered partial-line
// accesses.  Here every product runs on v_mfma_f32_32x32x16_f16 with the operands as two f16 pieces of a power-of-two
// scaled value (three instructions of 32 cycles per K = 16, fp32-grade results), so the WHOLE item list
// is cheap enough to be walked by every group, the batch is streamed in order, and rows stored back to back move as
// whole 16-byte pieces of the group's contiguous block.
//
//   step 0   t = NA_E' g            A = NA_E' (pairs, scale gN), B = g (per-sample power of two sg), once, kept
//   step 1   w = U_tile v           A = packed tile (pairs, scale gU x one power of two f_s per segment), B = v (sv);
//            every lane zeroes the quads that do not belong to ITS active segment and scales the rest by 1 / ||U v||:
//            w is a unit vector whatever the scales were
//   step 2   u += U_tile' w         A = the transposed tile (same scales), B = 2^13 w as pairs
//   out      grad kappa = phi_s + u / (gU f_s 2^13)    (phi_s: a row gather from the fp32 rows, as for linear rows)
//
// Accepted per pack by a creation-time measurement against the fp64 lane backward, next to the exact-fp32 kernel
// (rayen_abi.hip::bwd32_selfcheck); RAYEN_old's head and everything else stay on rayen_mfma_bwdg.hip.
#include "rayen_bwd_tiles.h"
#include "rayen_split_image.h"

#include <cmath>
#include <algorithm>
#include <cstring>
#include <mutex>
#include <vector>

#ifndef RAYEN_BWDP_ABL
#define RAYEN_BWDP_ABL 0   // developer ablation builds (scripts/ubench/tu_variant.sh): 1 no walk | 2 no NA_E' g | 8 no store
#endif

namespace rayen {

struct MfmaBwdpImage {
  f16x8* U = nullptr;        // [tile][2 K-steps][2 pieces][64] x 8 f16: PACK1 tile, its transpose, ...
  f16x8* NT = nullptr;       // NA_E' as [2 nkg K-steps][2 pieces][64] x 8 f16 (rows = subspace coordinates), null when NA_E = I
  int32_t* seg_info = nullptr; // [n_segments + 1][4]: tile pair that holds the segment (-1: none) | quad a | half + 2 x (spans
                               // both halves) | 1 / f_s (float bits)
  int32_t* seg_aux = nullptr;  // [n_segments + 1] W row of phi for factor segments, -1 otherwise
  float* Wrow = nullptr;       // [n_rows + 2][32] fp32 rows (linear rows and phi: gathered, never multiplied on the MFMA)
  int n_pairs = 0, nkg = 0, n_simd = 1024;   // n_pairs: (packed tile, its transpose) pairs = tiles / 2
  int lds_bytes = 0;           // dynamic LDS of the resident instance (U image + NA_E' image + descriptors), 0: does not fit
  float u_unscale = 1.f;       // 1 / (gU 2^13)
  float n_inv = 1.f;           // 1 / gN
  int64_t bytes = 0;
};

namespace {

// rows stored back to back (ld == width) behind a 16-byte aligned base: the 32 rows of a sample tile are ONE block of
// 32 width floats (a multiple of 16 bytes), moved as whole 16-byte pieces through the patch as a flat array
// ... in two steps, so that a group can have all its row loads in flight before it waits for the first of them:
// issue (global -> registers, 16-byte pieces idx = lane + 64 j of each tile's block) and land (through the patch into the
// fragment layout of an MFMA B operand: this lane's 4-column groups of ITS sample's row)
template <int NT, int NK>
__device__ __forceinline__ void issue_rows_flat(f32x4 (&piece)[NT][NK * 4], const float* __restrict__ src, const int width,
                                                const int64_t s_base, const int64_t B, const int lane) {
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int64_t row0 = s_base + 32 * t;
    const int64_t left = B - row0;
    const int nfl = (int)(left >= 32 ? 32 : (left > 0 ? left : 0)) * width;
    const float* blk = src + row0 * (int64_t)width;
#pragma unroll
    for (int jj = 0; jj < NK * 4; ++jj) {
      const int i4 = lane + 64 * jj;
      f32x4 x = {0.f, 0.f, 0.f, 0.f};
      if (i4 < 8 * width) {
        if (4 * i4 + 3 < nfl) {
          x = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(blk + 4 * i4));
        } else {
          if (4 * i4 + 0 < nfl) x[0] = blk[4 * i4 + 0];
          if (4 * i4 + 1 < nfl) x[1] = blk[4 * i4 + 1];
          if (4 * i4 + 2 < nfl) x[2] = blk[4 * i4 + 2];
        }
      }
      piece[t][jj] = x;
    }
  }
}

template <int NT, int NK, int LSTR>
__device__ __forceinline__ void land_rows_flat(float (&dst)[NT][NK * 16], const f32x4 (&piece)[NT][NK * 4], const int width,
                                               float (*patch)[LSTR], const int lane) {
  float* flat = &patch[0][0];
  const int col = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int jj = 0; jj < NK * 4; ++jj) {
      const int i4 = lane + 64 * jj;
      if (i4 < 8 * width) *reinterpret_cast<f32x4*>(flat + 4 * i4) = piece[t][jj];
    }
    __builtin_amdgcn_wave_barrier();
    const float* myrow = flat + col * width + 4 * hi;
#pragma unroll
    for (int q = 0; q < NK * 4; ++q)
#pragma unroll
      for (int c = 0; c < 4; ++c) dst[t][4 * q + c] = (8 * q + 4 * hi + c < width) ? myrow[8 * q + c] : 0.f;
    __builtin_amdgcn_wave_barrier();
  }
}

template <int NT, int NK, int LSTR>
__device__ __forceinline__ void store_rows_flat(const float (&val)[NT][NK * 16], float* __restrict__ dst, const int width,
                                                const int64_t s_base, const int64_t B, float (*patch)[LSTR], const int lane) {
  float* flat = &patch[0][0];
  const int col = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int64_t row0 = s_base + 32 * t;
    const int64_t left = B - row0;
    const int nfl = (int)(left >= 32 ? 32 : (left > 0 ? left : 0)) * width;
    float* myrow = flat + col * width + 4 * hi;
#pragma unroll
    for (int q = 0; q < NK * 4; ++q)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (8 * q + 4 * hi + c < width) myrow[8 * q + c] = val[t][4 * q + c];
    __builtin_amdgcn_wave_barrier();
    float* blk = dst + row0 * (int64_t)width;
#pragma unroll
    for (int jj = 0; jj < NK * 4; ++jj) {
      const int i4 = lane + 64 * jj;
      if (i4 < 8 * width) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(flat + 4 * i4);
        if (4 * i4 + 3 < nfl) {
          __builtin_nontemporal_store(x, reinterpret_cast<f32x4*>(blk + 4 * i4));
        } else {
          if (4 * i4 + 0 < nfl) blk[4 * i4 + 0] = x[0];
          if (4 * i4 + 1 < nfl) blk[4 * i4 + 1] = x[1];
          if (4 * i4 + 2 < nfl) blk[4 * i4 + 2] = x[2];
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// the eight values a lane holds of one K-step of an MFMA B operand -> two f16 pieces of scale x value
__device__ __forceinline__ void split8(const float* x, const float scale, f16x8& p1, f16x8& p2) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float s = x[i] * scale;
    const _Float16 h = (_Float16)s;
    p1[i] = h;
    p2[i] = (_Float16)(s - (float)h);
  }
}

}  // namespace

// NKG: 32-column blocks of the incoming gradient (k_pad / 32; 0 = NA_E is the identity).  FLAT: the rows of v, grad_v and
// grad_y are stored back to back, each tensor behind a 16-byte aligned base.
// RES: the whole image (packed tiles, NA_E', descriptors) is copied into LDS once per workgroup and every wave reads its A
// operands from there -- streamed from L2 instead, the eight waves of a CU pull 80 KB per 64 samples through the
// vector-memory path (config 5r: 330 MB per launch, the walk 54 us of a 95 us kernel; resident: DESIGN.md 4.3).
template <int NKG, bool RES, bool FLAT>
__global__ __launch_bounds__(kMfmaWaves * 64, kMfmaWaves / 4) void mfma_bwdp_kernel(
    const f16x8* __restrict__ Uimg, const f16x8* __restrict__ NTimg, const int n_pairs,
    const int32_t* __restrict__ seg_info, const int32_t* __restrict__ seg_aux,
    const float* __restrict__ Wrow, const int n, const int k, const float* __restrict__ v, const int64_t B,
    const int64_t ldv, const int vec_v, const float* __restrict__ kappa, const int32_t* __restrict__ active,
    const float* __restrict__ gy, const int64_t ldg, const int vec_g, float* __restrict__ gv, const int64_t ldgv,
    const int vec_o, const float u_unscale, const float n_inv) {
  constexpr int NT = 2, KK = 16, NP = 32;
  constexpr int NKL = NKG > 1 ? NKG : 1, LSTR = NKL * 32 + 4;
  constexpr int KG = NKG > 0 ? NKG * 16 : 16, NSG = NKG * 2;
  __shared__ __attribute__((aligned(16))) float line_lds[kMfmaWaves][32][LSTR];
  extern __shared__ __attribute__((aligned(16))) char res_lds[];   // RES: [2 n_pairs + 2 tiles x 4 KiB][NA_E': NSG x 2 KiB]
  const f16x8* u_res = reinterpret_cast<const f16x8*>(res_lds);
  const f16x8* nt_res = u_res + (size_t)(n_pairs * 2 + 2) * 4 * 64;   // (two spare tiles: the walk looks ahead)
  if constexpr (RES) {
    f16x8* dst = reinterpret_cast<f16x8*>(res_lds);
    const int n_u = (n_pairs * 2 + 2) * 4 * 64, n_nt = NSG * 2 * 64;
    for (int i = threadIdx.x; i < n_u; i += kMfmaWaves * 64) dst[i] = Uimg[i];
    for (int i = threadIdx.x; i < n_nt; i += kMfmaWaves * 64) dst[n_u + i] = NTimg[i];
    __syncthreads();   // the only workgroup barrier
  }

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int col = lane & 31;
  const int hi = lane >> 5;
  const int64_t n_groups = (B + NT * 32 - 1) / (NT * 32);
  const int64_t wave_id = (int64_t)blockIdx.x * kMfmaWaves + wave;
  const int64_t wave_stride = (int64_t)gridDim.x * kMfmaWaves;
  float (*patch)[LSTR] = line_lds[wave];
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  for (int64_t grp = wave_id; grp < n_groups; grp += wave_stride) {
    const int64_t s_base = grp * (NT * 32);
    bool live[NT], clipped[NT], pmatched[NT], mine_half[NT];
    int mp[NT], myq[NT];         // tile pair and quad of the sample's active segment (-1: a linear row / interior)
    float tr[NT][KK];
    f16x8 vb[NT][2][2];          // v as MFMA B operand: [tile][piece][K-step]
    float tv[NT], sc[NT], sinv[NT];
    int aseg[NT], arow[NT];
    f32x16 u16[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) live[t] = (s_base + t * 32 + col) < B;

    // ---- every load of the group goes out first: the rows of g and v (flat: 16-byte pieces of the group's blocks, up
    // to 96 registers -- nothing else is live yet), kappa and the arg-max record
    f32x4 g_piece[FLAT ? NT : 1][NKL * 4], v_piece[FLAT ? NT : 1][4];
    if constexpr (FLAT) {
      issue_rows_flat<NT, NKL>(g_piece, gy, NKG == 0 ? n : k, s_base, B, lane);
      issue_rows_flat<NT, 1>(v_piece, v, n, s_base, B, lane);
    }
    float kap_in[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int64_t smp = live[t] ? s_base + t * 32 + col : 0;
      kap_in[t] = live[t] ? kappa[smp] : 0.f;
      aseg[t] = live[t] ? active[2 * smp] : -1;
      arow[t] = live[t] ? active[2 * smp + 1] : 0;
    }

    // ---- t = NA_E' g (or g itself), in the register layout of v
    if constexpr (NKG == 0) {
      if constexpr (FLAT) land_rows_flat<NT, 1, LSTR>(tr, g_piece, n, patch, lane);
      else load_rows<NT, 1, LSTR, true>(tr, gy, ldg, n, vec_g, s_base, B, live, patch, lane);
    } else {
      float gr[NT][KG];
      if constexpr (FLAT) land_rows_flat<NT, NKL, LSTR>(gr, g_piece, k, patch, lane);
      else load_rows<NT, NKL, LSTR, true>(gr, gy, ldg, k, vec_g, s_base, B, live, patch, lane);
      f16x8 a[NSG][2];
      const f16x8* nsrc = (RES ? nt_res : NTimg) + lane;
#pragma unroll
      for (int sp = 0; sp < NSG; ++sp) {
        a[sp][0] = nsrc[(sp * 2 + 0) * 64];
        a[sp][1] = nsrc[(sp * 2 + 1) * 64];
      }
      f16x8 gb[NT][2][NSG];
      float sg_inv[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        float m = 0.f;
#pragma unroll
        for (int i = 0; i < KG; ++i) m = fmaxf(m, __builtin_fabsf(gr[t][i]));
        m = fmaxf(m, xhalf(m));
        float sg;
        int sg_exp;
        pow2_scale(m, sg, sg_inv[t], sg_exp);
#pragma unroll
        for (int sp = 0; sp < NSG; ++sp) split8(&gr[t][8 * sp], sg, gb[t][0][sp], gb[t][1][sp]);
      }
      f32x16 acc[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = zero;
      // (cross products first, leading products last: DESIGN.md 4.0b; the two sample tiles alternate -- two independent
      // accumulator chains keep the pipe busy)
      if constexpr (!(RAYEN_BWDP_ABL & 2)) {
#pragma unroll
        for (int sp = 0; sp < NSG; ++sp) {
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[sp][1], gb[t][0][sp], acc[t], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[sp][0], gb[t][1][sp], acc[t], 0, 0, 0);
        }
#pragma unroll
        for (int sp = 0; sp < NSG; ++sp)
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[sp][0], gb[t][0][sp], acc[t], 0, 0, 0);
      } else {
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t][0] = (float)gb[t][0][0][0] + (float)gb[t][1][NSG - 1][7];
      }
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 16; ++g) tr[t][g] = (acc[t][g] * n_inv) * sg_inv[t];
    }

    // ---- v: t . v in fp32, then the pieces
    float v_inv[NT];
    {
      float vr[NT][KK];
      if constexpr (FLAT) land_rows_flat<NT, 1, LSTR>(vr, v_piece, n, patch, lane);
      else load_rows<NT, 1, LSTR, true>(vr, v, ldv, n, vec_v, s_base, B, live, patch, lane);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        float dot = 0.f, m = 0.f;
#pragma unroll
        for (int i = 0; i < KK; ++i) {
          dot = fmaf(tr[t][i], vr[t][i], dot);
          m = fmaxf(m, __builtin_fabsf(vr[t][i]));
        }
        tv[t] = dot + xhalf(dot);
        m = fmaxf(m, xhalf(m));
        float sv;
        int sv_exp;
        pow2_scale(m, sv, v_inv[t], sv_exp);
        split8(&vr[t][0], sv, vb[t][0][0], vb[t][1][0]);
        split8(&vr[t][8], sv, vb[t][0][1], vb[t][1][1]);
      }
    }
    bool any = false;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float kap = kap_in[t];
      clipped[t] = live[t] && kap > 1.f && aseg[t] >= 0;
      sc[t] = 1.f / fmaxf(1.f, kap);
      // where the active segment sits (every quadratic of these packs is in exactly one packed tile)
      // (four plain loads: read as one int4 under the `clipped` predicate, hipcc 7.2 returned element 0 for element 3)
      const int32_t* si = seg_info + 4 * (clipped[t] ? aseg[t] : 0);
      const int where = clipped[t] ? si[2] : 0;
      mp[t] = clipped[t] ? si[0] : -1;
      myq[t] = clipped[t] ? si[1] : 0;
      mine_half[t] = (where & 2) != 0 || (where & 1) == hi;
      sinv[t] = clipped[t] ? reinterpret_cast<const float*>(si)[3] : 0.f;
      pmatched[t] = mp[t] >= 0;
      u16[t] = zero;
      any |= pmatched[t];
    }

    if (!(RAYEN_BWDP_ABL & 1) && __ballot(any) != 0) {  // wave-uniform: a wave of interior samples skips the walk
      f16x8 wb[NT][2][2];               // step-1 result, masked and normalised, as the B operand of step 2
      // one tile's A operands [K-step][piece]: LDS (resident) or L2
      auto fetch_tile = [&](f16x8 (&buf)[2][2], const int tile) {
        const f16x8* src = (RES ? u_res : Uimg) + (size_t)tile * (4 * 64) + lane;
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
          buf[sp][0] = src[(sp * 2 + 0) * 64];
          buf[sp][1] = src[(sp * 2 + 1) * 64];
        }
      };
      // acc[t] (+)= A x b[t] for both sample tiles, alternating (two independent accumulator chains)
      auto product2 = [&](const f16x8 (&a)[2][2], const f16x8 (&b)[NT][2][2], f32x16 (&acc)[NT]) {
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[sp][1], b[t][0][sp], acc[t], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[sp][0], b[t][1][sp], acc[t], 0, 0, 0);
        }
#pragma unroll
        for (int sp = 0; sp < 2; ++sp)
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[sp][0], b[t][0][sp], acc[t], 0, 0, 0);
      };
      f16x8 buf_a[2][2], buf_b[2][2];
      // ---- phase A: w = U_tile v for every packed tile; a sample keeps the result of ITS tile (16 selects per tile and
      // sample tile -- the normalisation and the split into pieces happen once per group, not once per tile)
      f32x16 wsel[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) wsel[t] = zero;
      fetch_tile(buf_a, 0);
      for (int pr = 0; pr < n_pairs; pr += 2) {   // (two spare tiles behind the list: the look-ahead of an odd count is harmless)
        f32x16 acc[NT];
        fetch_tile(buf_b, 2 * pr + 2);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = zero;
        product2(buf_a, vb, acc);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int g = 0; g < 16; ++g) wsel[t][g] = mp[t] == pr ? acc[t][g] : wsel[t][g];
        if (pr + 1 < n_pairs) {
          fetch_tile(buf_a, 2 * pr + 4);
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t] = zero;
          product2(buf_b, vb, acc);
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int g = 0; g < 16; ++g) wsel[t][g] = mp[t] == pr + 1 ? acc[t][g] : wsel[t][g];
        }
      }
      // ---- the unit vector w = U_s v / ||U_s v|| of the sample's own segment (the rows of its quad; of both halves for a
      // segment of rank 5..8), as pieces of 2^13 w
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        float qs = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          const bool keep = pmatched[t] && mine_half[t] && (g >> 2) == myq[t];
          wsel[t][g] = keep ? wsel[t][g] : 0.f;
          qs = fmaf(wsel[t][g], wsel[t][g], qs);
        }
        qs += xhalf(qs);   // (the other half holds the rest of a rank 5..8 segment, zeros otherwise)
        const float cw = qs > 0.f ? 8192.f * __builtin_amdgcn_rsqf(qs) : 0.f;
        float w[16];
        // (the product is NOT a power-of-two scaling: it must exist as ONE rounded fp32 value before it is split.  Left to
        // itself hipcc fuses it into the conversions -- piece 1 from the rounded product, piece 2 as
        // fma(acc, cw, -f16(acc cw)) with a singly rounded f16 -- and near a rounding tie the two disagree about piece 1
        // by one f16 ulp: 6e-5 of the unit vector, one row in four thousand)
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          float prod = wsel[t][g] * cw;
          asm volatile("" : "+v"(prod));
          w[g] = prod;
        }
        split8(&w[0], 1.f, wb[t][0][0], wb[t][1][0]);
        split8(&w[8], 1.f, wb[t][0][1], wb[t][1][1]);
      }
      // ---- phase B: u += U_tile' w, every tile with the w of the samples that belong to it (zeros for the others)
      const f16x8 none = {0, 0, 0, 0, 0, 0, 0, 0};
      fetch_tile(buf_a, 1);
      for (int pr = 0; pr < n_pairs; pr += 2) {
        f16x8 bsel[NT][2][2];
        fetch_tile(buf_b, 2 * pr + 3);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int pc = 0; pc < 2; ++pc)
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) bsel[t][pc][sp] = mp[t] == pr ? wb[t][pc][sp] : none;
        product2(buf_a, bsel, u16);
        if (pr + 1 < n_pairs) {
          fetch_tile(buf_a, 2 * pr + 5);
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc)
#pragma unroll
              for (int sp = 0; sp < 2; ++sp) bsel[t][pc][sp] = mp[t] == pr + 1 ? wb[t][pc][sp] : none;
          product2(buf_b, bsel, u16);
        }
      }
    }

    // ---- grad kappa: u back to natural units + phi of the packed quadratic, or the active linear row
    float out[NT][KK];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float us = u_unscale * sinv[t];
      float gk[KK];
#pragma unroll
      for (int i = 0; i < KK; ++i) gk[i] = pmatched[t] ? u16[t][i] * us : 0.f;
      if (clipped[t]) {
        const int rowi = pmatched[t] ? seg_aux[aseg[t]] : arow[t];
        const float* row = Wrow + (int64_t)rowi * NP + 4 * hi;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 x = *reinterpret_cast<const f32x4*>(row + 8 * q);
#pragma unroll
          for (int c = 0; c < 4; ++c) gk[4 * q + c] += x[c];
        }
      }
      const float coef = clipped[t] ? sc[t] * sc[t] * tv[t] : 0.f;
#pragma unroll
      for (int i = 0; i < KK; ++i) out[t][i] = fmaf(sc[t], tr[t][i], -coef * gk[i]);
    }
    if (RAYEN_BWDP_ABL & 8) {
      if (out[0][0] == 123.456f) gv[0] = out[1][3];
    } else if constexpr (FLAT) {
      store_rows_flat<NT, 1, LSTR>(out, gv, n, s_base, B, patch, lane);
    } else {
      float one[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) one[t] = 1.f;
      (void)store_rows<NT, 1, LSTR, true>(out, one, nullptr, gv, ldgv, n, vec_o, s_base, B, live, patch, lane);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------

// n <= 32, k <= 64, no LMI, and every quadratic-like segment a small factor (packed tiles only; at least one)
bool mfma_bwdp_eligible(const RayenPack* p) {
  if (p->n > 32 || p->k > 64 || (p->out_identity && p->k != p->n)) return false;
  int small = 0;
  for (const RayenSegment& g : p->segs) {
    if (g.type == RAYEN_SEG_LMI) return false;
    if (!bwd_quad_like(g)) continue;
    if (!is_small_factor(g)) return false;
    ++small;
  }
  return small > 0 && 2 * ((small + 3) / 4) <= 96;
}

void mfma_bwdp_free(MfmaBwdpImage* img) {
  if (img == nullptr) return;
  if (img->U) (void)hipFree(img->U);
  if (img->NT) (void)hipFree(img->NT);
  if (img->seg_info) (void)hipFree(img->seg_info);
  if (img->seg_aux) (void)hipFree(img->seg_aux);
  if (img->Wrow) (void)hipFree(img->Wrow);
  delete img;
}

namespace {

template <typename T>
bool upload(const std::vector<T>& host, T** dev, int64_t* bytes) {
  if (hipMalloc(dev, host.size() * sizeof(T)) != hipSuccess) return false;
  if (hipMemcpy(*dev, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return false;
  *bytes += (int64_t)(host.size() * sizeof(T));
  return true;
}

// power of two that puts `big` into [2^13, 2^14)
void pow2_for(const double big, float* scale, float* inv) {
  int ex = 0;
  if (big > 0.0) (void)std::frexp(big, &ex);
  int shift = big > 0.0 ? 14 - ex : 0;
  shift = shift > 100 ? 100 : (shift < -100 ? -100 : shift);
  *scale = std::ldexp(1.0f, shift);
  *inv = std::ldexp(1.0f, -shift);
}

// two f16 pieces of scale x every entry, in the fragment order of v_mfma_f32_32x32x16_f16: chunk (tile, K-step s, piece)
// = 64 lanes x 8 elements, element i of lane l = column 16 s + 8 (i >> 2) + 4 (l >> 5) + (i & 3) of row l & 31
std::vector<_Float16> pair_chunks(const TileLayout& b, const float scale) {
  const std::vector<float> frag = b.fragments_f32();
  const int n_tiles = b.n_tiles(), nq = b.nq(), ns = nq / 2;
  std::vector<_Float16> wh((size_t)n_tiles * ns * 2 * 64 * 8);
  for (int t = 0; t < n_tiles; ++t)
    for (int sp = 0; sp < ns; ++sp)
      for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 8; ++i) {
          const float x = frag[(((size_t)t * nq + 2 * sp + (i >> 2)) * 64 + l) * 4 + (i & 3)] * scale;
          const _Float16 h1 = (_Float16)x;
          const _Float16 h2 = (_Float16)(x - (float)h1);
          const size_t base = (((size_t)t * ns + sp) * 2) * 64 * 8 + (size_t)l * 8 + i;
          wh[base] = h1;
          wh[base + 64 * 8] = h2;
        }
  return wh;
}

}  // namespace

int mfma_bwdp_build(const RayenPack* p, MfmaBwdpImage** out, int64_t* bytes) {
  const int n = p->n, k = p->k, np = 32;
  const double* W = p->W.data();
  TileLayout b(n);
  std::vector<BItem> items;
  std::vector<BPack> packs;
  std::vector<int32_t> seg_aux;
  const int n_real = layout_bwdg_tiles(p, b, items, packs, seg_aux);
  // (packed tile, transpose) pairs and nothing else: pair i = items 2 i, 2 i + 1 = pack i
  if (n_real == 0 || n_real % 2 != 0) return RAYEN_E_UNSUPPORTED;
  for (int i = 0; i < n_real; i += 2)
    if (items[i].type != BI_PACK1 || items[i + 1].type != BI_PACK2 || items[i].aux_row != i / 2) return RAYEN_E_UNSUPPORTED;

  // one power of two per segment on top of the image's (rayen_mfma_pair.hip): rows of the packed tile, columns of its
  // transpose
  std::vector<float> pack_inv(packs.size() * 8, 1.f);
  {
    double image_big = 0.0;
    std::vector<double> seg_big(p->segs.size() + 1, 0.0);
    for (int i = 0; i < n_real; ++i) {
      if (items[i].type != BI_PACK1) continue;
      const BPack& pk = packs[items[i].aux_row];
      for (int a = 0; a < 4; ++a)
        for (int h = 0; h < 2; ++h) {
          const int s = pk.seg[a][h];
          if (s < 0) continue;
          for (int c = 0; c < 4; ++c)
            for (int j = 0; j < np; ++j) {
              const double x = std::fabs(b.raw[((size_t)i * 32 + 8 * a + 4 * h + c) * np + j]);
              if (!std::isfinite(x)) continue;
              image_big = x > image_big ? x : image_big;
              seg_big[s] = x > seg_big[s] ? x : seg_big[s];
            }
        }
    }
    for (int i = 0; i < n_real; ++i) {
      if (items[i].type != BI_PACK1) continue;
      const BPack& pk = packs[items[i].aux_row];
      for (int a = 0; a < 4; ++a)
        for (int h = 0; h < 2; ++h) {
          const int s = pk.seg[a][h];
          if (s < 0 || !(seg_big[s] > 0.0) || !(image_big > 0.0)) continue;
          int ex_seg = 0, ex_img = 0;
          (void)std::frexp(seg_big[s], &ex_seg);
          (void)std::frexp(image_big, &ex_img);
          int e = ex_img - ex_seg;
          e = e < 0 ? 0 : (e > 60 ? 60 : e);
          const double boost = std::ldexp(1.0, e);
          pack_inv[(size_t)items[i].aux_row * 8 + 2 * a + h] = (float)std::ldexp(1.0, -e);
          if (e == 0) continue;
          for (int c = 0; c < 4; ++c) {
            const int r = 8 * a + 4 * h + c;
            for (int j = 0; j < np; ++j) {
              b.raw[((size_t)i * 32 + r) * np + j] *= boost;          // row r of the tile
              b.raw[((size_t)(i + 1) * 32 + j) * np + r] *= boost;    // column r of its transpose
            }
          }
        }
    }
  }

  MfmaBwdpImage* img = new MfmaBwdpImage();
  img->nkg = p->out_identity ? 0 : n_pad_of(k) / 32;
  img->n_pairs = n_real / 2;
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, p->device) == hipSuccess && prop.multiProcessorCount > 0)
      img->n_simd = prop.multiProcessorCount * 4;
  }
  std::vector<int32_t> seg_info((p->segs.size() + 1) * 4, 0);
  for (size_t sidx = 0; sidx <= p->segs.size(); ++sidx) seg_info[4 * sidx] = -1;
  for (int pr = 0; pr < img->n_pairs; ++pr)
    for (int a = 0; a < 4; ++a)
      for (int h = 0; h < 2; ++h) {
        const int sidx = packs[pr].seg[a][h];
        if (sidx < 0) continue;
        const bool both = (packs[pr].pair_bits >> a) & 1;
        int32_t* d = &seg_info[4 * (size_t)sidx];
        d[0] = pr;
        d[1] = a;
        d[2] = (both ? 2 : 0) | (both ? 0 : h);
        std::memcpy(&d[3], &pack_inv[(size_t)pr * 8 + 2 * a + h], 4);
      }
  double big = 0.0;
  for (const double x : b.raw)
    if (std::isfinite(x)) big = std::fabs(x) > big ? std::fabs(x) : big;
  float u_scale = 1.f, u_inv = 1.f;
  pow2_for(big, &u_scale, &u_inv);
  img->u_unscale = u_inv * (1.0f / 8192.f);
  const std::vector<_Float16> uh = pair_chunks(b, u_scale);
  std::vector<float> wrow((size_t)(p->n_rows + 2) * np, 0.f);
  for (int r = 0; r < p->n_rows; ++r)
    for (int j = 0; j < n; ++j) wrow[(size_t)r * np + j] = (float)W[(size_t)r * n + j];
  bool ok = true;
  {
    _Float16* d = nullptr;
    ok = ok && upload(uh, &d, &img->bytes);
    img->U = reinterpret_cast<f16x8*>(d);
  }
  ok = ok && upload(wrow, &img->Wrow, &img->bytes) && upload(seg_aux, &img->seg_aux, &img->bytes) &&
       upload(seg_info, &img->seg_info, &img->bytes);
  if (ok && !p->out_identity) {
    // NA_E' : rows = the n subspace coordinates, K = the k ambient coordinates
    TileLayout bn(k);
    std::vector<std::vector<double>> nt(n, std::vector<double>(k, 0.0));
    double nbig = 0.0;
    for (int i = 0; i < k; ++i)
      for (int e = 0; e < n; ++e) {
        nt[e][i] = p->NA_E[(size_t)i * n + e];
        if (std::isfinite(nt[e][i])) nbig = std::fabs(nt[e][i]) > nbig ? std::fabs(nt[e][i]) : nbig;
      }
    std::vector<const double*> rows;
    for (int r = 0; r < n; ++r) rows.push_back(nt[r].data());
    bn.add_tile(rows, k);
    float n_scale = 1.f;
    pow2_for(nbig, &n_scale, &img->n_inv);
    const std::vector<_Float16> nh = pair_chunks(bn, n_scale);
    _Float16* d = nullptr;
    ok = upload(nh, &d, &img->bytes);
    img->NT = reinterpret_cast<f16x8*>(d);
  }
  if (!ok) { mfma_bwdp_free(img); return RAYEN_E_ALLOC; }
  // the resident instance: everything in LDS next to the kernel's static row patches (160 KiB per CU, one workgroup)
  {
    const int nkl = img->nkg > 1 ? img->nkg : 1;
    const int64_t patch = (int64_t)kMfmaWaves * 32 * (nkl * 32 + 4) * 4;
    const int64_t need = (int64_t)(img->n_pairs * 2 + 2) * 4096 + (int64_t)img->nkg * 2 * 2048;
    img->lds_bytes = (need + patch + 512 <= 160 * 1024) ? (int)need : 0;
    if (img->lds_bytes > 0) {
      // (per kernel instance, never lowered: a later pack with fewer pairs must not take back what an earlier one was
      // promised -- every pack asks for the running maximum of its instance)
      static std::mutex mu;
      static int promised[3] = {0, 0, 0};
      std::lock_guard<std::mutex> hold(mu);
      int ask = img->lds_bytes;
      if (img->nkg >= 0 && img->nkg <= 2) { promised[img->nkg] = std::max(promised[img->nkg], ask); ask = promised[img->nkg]; }
      const void* fns[2] = {nullptr, nullptr};
      if (img->nkg == 0) { fns[0] = reinterpret_cast<const void*>(&mfma_bwdp_kernel<0, true, true>); fns[1] = reinterpret_cast<const void*>(&mfma_bwdp_kernel<0, true, false>); }
      if (img->nkg == 1) { fns[0] = reinterpret_cast<const void*>(&mfma_bwdp_kernel<1, true, true>); fns[1] = reinterpret_cast<const void*>(&mfma_bwdp_kernel<1, true, false>); }
      if (img->nkg == 2) { fns[0] = reinterpret_cast<const void*>(&mfma_bwdp_kernel<2, true, true>); fns[1] = reinterpret_cast<const void*>(&mfma_bwdp_kernel<2, true, false>); }
      for (const void* fn : fns)
        if (fn == nullptr || hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, ask) != hipSuccess) {
          (void)hipGetLastError();
          img->lds_bytes = 0;
        }
    }
  }
  *bytes = img->bytes;
  *out = img;
  return RAYEN_OK;
}

template <int NKG>
static int launch_bwdp(const RayenPack* p, const MfmaBwdpImage* img, const float* v, int64_t B, int64_t ldv,
                       const float* kappa, const int32_t* active, const float* gy, int64_t ldg, float* gv,
                       int64_t ldgv, hipStream_t stream) {
  const int64_t n_groups = (B + 63) / 64;
  const int64_t slots = (int64_t)launch_simds(img->n_simd) * kMfmaWavesPerSimd;
  const int64_t rounds = (n_groups + slots - 1) / slots;
  const int64_t waves = (n_groups + rounds - 1) / rounds;
  const int64_t grid = (waves + kMfmaWaves - 1) / kMfmaWaves;
  auto aligned = [](const void* ptr, int64_t ld) { return (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(ptr) & 15) == 0); };
  auto base16 = [](const void* ptr) { return (reinterpret_cast<uintptr_t>(ptr) & 15) == 0; };
  // rows of v, grad_v and grad_y stored back to back behind 16-byte aligned bases: whole 16-byte pieces of the group's blocks
  const bool flat = ldv == p->n && ldgv == p->n && base16(v) && base16(gv) && ldg == p->k && base16(gy);
  auto go = [&](auto kern, const unsigned lds) {
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kMfmaWaves * 64), lds, stream, img->U, img->NT, img->n_pairs,
                       img->seg_info, img->seg_aux, img->Wrow, p->n, p->k, v, B, ldv, aligned(v, ldv) ? 1 : 0, kappa, active,
                       gy, ldg, aligned(gy, ldg) ? 1 : 0, gv, ldgv, aligned(gv, ldgv) ? 1 : 0, img->u_unscale,
                       img->n_inv);
  };
  // (resident wherever the image fits: one copy per workgroup against one stream per wave and group)
  if (img->lds_bytes > 0 && n_groups >= 2) {
    if (flat) go(mfma_bwdp_kernel<NKG, true, true>, (unsigned)img->lds_bytes);
    else go(mfma_bwdp_kernel<NKG, true, false>, (unsigned)img->lds_bytes);
  } else {
    go(mfma_bwdp_kernel<NKG, false, false>, 0u);
  }
  return hipGetLastError() == hipSuccess ? RAYEN_OK : RAYEN_E_LAUNCH;
}

int mfma_bwdp_backward(const RayenPack* p, const MfmaBwdpImage* img, const float* v, int64_t B, int64_t ldv,
                       const float* kappa, const int32_t* active, const float* grad_y, int64_t ldg,
                       float* grad_v, int64_t ldgv, hipStream_t stream) {
  if (B == 0) return RAYEN_OK;
  if (img->nkg == 0) return launch_bwdp<0>(p, img, v, B, ldv, kappa, active, grad_y, ldg, grad_v, ldgv, stream);
  if (img->nkg == 1) return launch_bwdp<1>(p, img, v, B, ldv, kappa, active, grad_y, ldg, grad_v, ldgv, stream);
  if (img->nkg == 2) return launch_bwdp<2>(p, img, v, B, ldv, kappa, active, grad_y, ldg, grad_v, ldgv, stream);
  return RAYEN_E_UNSUPPORTED;
}

}  // namespace rayen
