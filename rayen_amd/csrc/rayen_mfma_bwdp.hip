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
#include <cstring>
#include <vector>

namespace rayen {

struct MfmaBwdpImage {
  f16x8* U = nullptr;        // [tile][2 K-steps][2 pieces][64] x 8 f16: PACK1 tile, its transpose, ...
  f16x8* NT = nullptr;       // NA_E' as [2 nkg K-steps][2 pieces][64] x 8 f16 (rows = subspace coordinates), null when NA_E = I
  BItem* items = nullptr;
  BPack* packs = nullptr;
  float* pack_inv = nullptr;   // [n_packs][4][2]: 1 / f_s of the segment sitting in that half-quad
  int32_t* seg_aux = nullptr;  // [n_segments + 1] W row of phi for factor segments, -1 otherwise
  float* Wrow = nullptr;       // [n_rows + 2][32] fp32 rows (linear rows and phi: gathered, never multiplied on the MFMA)
  int n_items = 0, nkg = 0, n_simd = 1024;
  float u_unscale = 1.f;       // 1 / (gU 2^13)
  float n_inv = 1.f;           // 1 / gN
  int64_t bytes = 0;
};

namespace {

// rows stored back to back (ld == width) behind a 16-byte aligned base: the 32 rows of a sample tile are ONE block of
// 32 width floats (a multiple of 16 bytes), moved as whole 16-byte pieces through the patch as a flat array
template <int NT, int NK, int LSTR>
__device__ __forceinline__ void load_rows_flat(float (&dst)[NT][NK * 16], const float* __restrict__ src, const int width,
                                               const int64_t s_base, const int64_t B, float (*patch)[LSTR], const int lane) {
  float* flat = &patch[0][0];
  const int col = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int64_t row0 = s_base + 32 * t;
    const int64_t left = B - row0;
    const int nfl = (int)(left >= 32 ? 32 : (left > 0 ? left : 0)) * width;
    const float* blk = src + row0 * (int64_t)width;
#pragma unroll
    for (int jj = 0; jj < NK * 4; ++jj) {
      const int i4 = lane + 64 * jj;
      if (i4 < 8 * width) {
        f32x4 x = {0.f, 0.f, 0.f, 0.f};
        if (4 * i4 + 3 < nfl) {
          x = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(blk + 4 * i4));
        } else {
          if (4 * i4 + 0 < nfl) x[0] = blk[4 * i4 + 0];
          if (4 * i4 + 1 < nfl) x[1] = blk[4 * i4 + 1];
          if (4 * i4 + 2 < nfl) x[2] = blk[4 * i4 + 2];
        }
        *reinterpret_cast<f32x4*>(flat + 4 * i4) = x;
      }
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

// NKG: 32-column blocks of the incoming gradient (k_pad / 32; 0 = NA_E is the identity).  FLAT bits: 1 = v and grad_v
// rows back to back, 2 = grad_y rows back to back (each behind a 16-byte aligned base).
template <int NKG>
__global__ __launch_bounds__(kMfmaWaves * 64, kMfmaWaves / 4) void mfma_bwdp_kernel(
    const f16x8* __restrict__ Uimg, const f16x8* __restrict__ NTimg, const BItem* __restrict__ items, const int n_items,
    const BPack* __restrict__ packs, const float* __restrict__ pack_inv, const int32_t* __restrict__ seg_aux,
    const float* __restrict__ Wrow, const int n, const int k, const float* __restrict__ v, const int64_t B,
    const int64_t ldv, const int vec_v, const float* __restrict__ kappa, const int32_t* __restrict__ active,
    const float* __restrict__ gy, const int64_t ldg, const int vec_g, float* __restrict__ gv, const int64_t ldgv,
    const int vec_o, const int flat, const float u_unscale, const float n_inv) {
  constexpr int NT = 2, KK = 16, NP = 32;
  constexpr int NKL = NKG > 1 ? NKG : 1, LSTR = NKL * 32 + 4;
  constexpr int KG = NKG > 0 ? NKG * 16 : 16, NSG = NKG * 2;
  __shared__ __attribute__((aligned(16))) float line_lds[kMfmaWaves][32][LSTR];

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
    bool live[NT], clipped[NT], pmatched[NT];
    float tr[NT][KK];
    f16x8 vb[NT][2][2];          // v as MFMA B operand: [tile][piece][K-step]
    float tv[NT], sc[NT], sinv[NT];
    int aseg[NT], arow[NT];
    f32x16 u16[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) live[t] = (s_base + t * 32 + col) < B;

    // ---- t = NA_E' g (or g itself), in the register layout of v
    if constexpr (NKG == 0) {
      if (flat & 2) load_rows_flat<NT, 1, LSTR>(tr, gy, n, s_base, B, patch, lane);
      else load_rows<NT, 1, LSTR, true>(tr, gy, ldg, n, vec_g, s_base, B, live, patch, lane);
    } else {
      float gr[NT][KG];
      if (flat & 2) load_rows_flat<NT, NKL, LSTR>(gr, gy, k, s_base, B, patch, lane);
      else load_rows<NT, NKL, LSTR, true>(gr, gy, ldg, k, vec_g, s_base, B, live, patch, lane);
      f16x8 a[NSG][2];
#pragma unroll
      for (int sp = 0; sp < NSG; ++sp) {
        a[sp][0] = NTimg[(sp * 2 + 0) * 64 + lane];
        a[sp][1] = NTimg[(sp * 2 + 1) * 64 + lane];
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        float m = 0.f;
#pragma unroll
        for (int i = 0; i < KG; ++i) m = fmaxf(m, __builtin_fabsf(gr[t][i]));
        m = fmaxf(m, xhalf(m));
        float sg, sg_inv;
        int sg_exp;
        pow2_scale(m, sg, sg_inv, sg_exp);
        f16x8 gb[2][NSG > 0 ? NSG : 1];
#pragma unroll
        for (int sp = 0; sp < NSG; ++sp) split8(&gr[t][8 * sp], sg, gb[0][sp], gb[1][sp]);
        f32x16 acc = zero;
        // (cross products first, leading products last: DESIGN.md 4.0b)
#pragma unroll
        for (int sp = 0; sp < NSG; ++sp) {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[sp][1], gb[0][sp], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[sp][0], gb[1][sp], acc, 0, 0, 0);
        }
#pragma unroll
        for (int sp = 0; sp < NSG; ++sp) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[sp][0], gb[0][sp], acc, 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 16; ++g) tr[t][g] = (acc[g] * n_inv) * sg_inv;
      }
    }

    // ---- v: t . v in fp32, then the pieces
    float v_inv[NT];
    {
      float vr[NT][KK];
      if (flat & 1) load_rows_flat<NT, 1, LSTR>(vr, v, n, s_base, B, patch, lane);
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
      const int64_t smp = live[t] ? s_base + t * 32 + col : 0;
      const float kap = live[t] ? kappa[smp] : 0.f;
      aseg[t] = live[t] ? active[2 * smp] : -1;
      arow[t] = live[t] ? active[2 * smp + 1] : 0;
      clipped[t] = live[t] && kap > 1.f && aseg[t] >= 0;
      sc[t] = 1.f / fmaxf(1.f, kap);
      pmatched[t] = false;
      sinv[t] = 0.f;
      u16[t] = zero;
      any |= clipped[t];
    }

    if (__ballot(any) != 0 && n_items > 0) {  // wave-uniform: a wave of interior samples skips the walk
      const f16x8* up = Uimg + lane;
      f16x8 buf_a[2][2], buf_b[2][2];   // [K-step][piece]
      f16x8 wb[NT][2][2];               // step-1 result, masked and normalised, as the B operand of step 2
      auto fetch_tile = [&](f16x8 (&buf)[2][2]) {
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
          buf[sp][0] = up[(sp * 2 + 0) * 64];
          buf[sp][1] = up[(sp * 2 + 1) * 64];
        }
        up += 4 * 64;
        __builtin_amdgcn_sched_barrier(0);
      };
      auto product = [&](const f16x8 (&a)[2][2], const f16x8 (&b)[2][2], f32x16 acc) {
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[sp][1], b[0][sp], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[sp][0], b[1][sp], acc, 0, 0, 0);
        }
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[sp][0], b[0][sp], acc, 0, 0, 0);
        return acc;
      };
      auto process = [&](const BItem item, const f16x8 (&a)[2][2]) {
        if (item.type == BI_PACK2) {
          // u += U_tile' w (zero for every sample whose active segment is not in this tile)
#pragma unroll
          for (int t = 0; t < NT; ++t) u16[t] = product(a, wb[t], u16[t]);
          return;
        }
        if (item.type != BI_PACK1) return;
        const BPack pk = packs[item.aux_row];
        const float* pinv = pack_inv + (size_t)item.aux_row * 8;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const f32x16 acc = product(a, vb[t], zero);
          float w[16];
          bool got = false;
#pragma unroll
          for (int a4 = 0; a4 < 4; ++a4) {
            const int sid = hi ? pk.seg[a4][1] : pk.seg[a4][0];
            const bool mine = clipped[t] && sid >= 0 && sid == aseg[t];
            float qs = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) qs = fmaf(acc[4 * a4 + c], acc[4 * a4 + c], qs);
            qs = mine ? qs : 0.f;
            if ((pk.pair_bits >> a4) & 1) qs += xhalf(qs);  // the segment's other rows sit in the other half
            // (2^13: the unit vector as pieces of an f16-range number; undone with the image's scale at the end)
            const float cw = (mine && qs > 0.f) ? 8192.f * __builtin_amdgcn_rsqf(qs) : 0.f;
            // (the product is NOT a power-of-two scaling: it must exist as ONE rounded fp32 value before it is split.  Left to
            // itself hipcc fuses it into the conversions -- piece 1 from the rounded product, piece 2 as
            // fma(acc, cw, -f16(acc cw)) with a singly rounded f16 -- and near a rounding tie the two disagree about piece 1
            // by one f16 ulp: 6e-5 of the unit vector, one row in four thousand)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              float prod = acc[4 * a4 + c] * cw;
              asm volatile("" : "+v"(prod));
              w[4 * a4 + c] = prod;
            }
            if (mine) sinv[t] = pinv[2 * a4 + hi];
            got |= mine;
          }
          split8(&w[0], 1.f, wb[t][0][0], wb[t][1][0]);
          split8(&w[8], 1.f, wb[t][0][1], wb[t][1][1]);
          const int both = (got ? 1 : 0) | __shfl_xor(got ? 1 : 0, 32);
          pmatched[t] |= both != 0;
        }
      };
      fetch_tile(buf_a);
      for (int it = 0; it < n_items; it += 2) {  // (the item count is even; two spare tiles behind the list)
        fetch_tile(buf_b);
        process(items[it], buf_a);
        fetch_tile(buf_a);
        process(items[it + 1], buf_b);
      }
    }

    // ---- grad kappa: u back to natural units + phi of the packed quadratic, or the active linear row
    float out[NT][KK];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float s_other = xhalf(sinv[t]);
      const float us = u_unscale * fmaxf(sinv[t], s_other);
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
    if (flat & 1) {
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
  if (img->items) (void)hipFree(img->items);
  if (img->packs) (void)hipFree(img->packs);
  if (img->pack_inv) (void)hipFree(img->pack_inv);
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
  for (int i = 0; i < n_real; ++i)
    if (items[i].type != BI_PACK1 && items[i].type != BI_PACK2 && items[i].type != BI_NOP) return RAYEN_E_UNSUPPORTED;

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
  img->n_items = n_real;
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, p->device) == hipSuccess && prop.multiProcessorCount > 0)
      img->n_simd = prop.multiProcessorCount * 4;
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
  ok = ok && upload(wrow, &img->Wrow, &img->bytes) && upload(items, &img->items, &img->bytes) &&
       upload(packs, &img->packs, &img->bytes) && upload(seg_aux, &img->seg_aux, &img->bytes) &&
       upload(pack_inv, &img->pack_inv, &img->bytes);
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
  const int flat = ((ldv == p->n && ldgv == p->n && base16(v) && base16(gv)) ? 1 : 0) |
                   ((ldg == p->k && base16(gy)) ? 2 : 0);
  hipLaunchKernelGGL((mfma_bwdp_kernel<NKG>), dim3((unsigned)grid), dim3(kMfmaWaves * 64), 0, stream, img->U, img->NT,
                     img->items, img->n_items, img->packs, img->pack_inv, img->seg_aux, img->Wrow, p->n, p->k, v, B, ldv,
                     aligned(v, ldv) ? 1 : 0, kappa, active, gy, ldg, aligned(gy, ldg) ? 1 : 0, gv, ldgv,
                     aligned(gv, ldgv) ? 1 : 0, flat, img->u_unscale, img->n_inv);
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
