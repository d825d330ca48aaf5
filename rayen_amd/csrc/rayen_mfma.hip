// fp32 MFMA path: T = W . V' on v_mfma_f32_32x32x2_f32 with the whole epilogue in registers.
//
// One wave owns NT tiles of 32 samples and walks every 32-row tile of W.  Per
// tile it issues n_pad/2 MFMAs per sample tile (A = 32 rows of W, B = 32
// samples of V) and reduces the 32x32 result at once:
//
//   D[row][sample]: lane l holds sample l&31 and rows (g&3) + 8(g>>2) + 4(l>>5), g = 0..15
//
// so every reduction of computeKappa (rayen/constraint_module.py:351-458) runs
// along the lane's own 16 registers plus ONE exchange between the two half-waves.
// The K order of the MFMA chain is free, and is chosen so that the B operand of
// K-step kk IS the direction element that matches result register kk&15 of row
// tile kk>>4:  element(kk, half) = 8*(kk>>2) + 4*half + (kk&3).  Consequences:
//   * a lane loads its half of v as float4s and keeps it in registers for the
//     whole kernel (B operands of every tile, n/2 VGPRs);
//   * the quadratic form v'Gv needs no second pass: acc[g] * v[16t+g] summed;
//   * with NA_E = I the output y = y0 + v/max(1,kappa) is written straight from
//     those registers, again as float4s.
// W is stored in fragment order ([tile][k-group][lane] float4): each A fetch is
// one contiguous 1 KiB global_load_dwordx4 per wave, served by L2 (the image is
// <= a few hundred KiB and shared by every wave), prefetched a whole tile ahead into a
// second register buffer.  No LDS staging of W, no workgroup barriers in the tile walk.
//
// HBM traffic per sample: n loads + k stores, the algorithmic minimum.
#include "rayen_internal.h"
#include "rayen_tiles.h"

#include <cstring>
#include <vector>

namespace rayen {

using f32x16 = float __attribute__((ext_vector_type(16)));
using f32x4 = float __attribute__((ext_vector_type(4)));

#ifndef RAYEN_MFMA_NT
#define RAYEN_MFMA_NT 2
#endif
#ifndef RAYEN_MFMA_WPS
#define RAYEN_MFMA_WPS 2
#endif
constexpr int kMfmaWavesPerSimd = RAYEN_MFMA_WPS;

struct MfmaImage {
  MPack* packs = nullptr;
  void* Wb = nullptr;      // split-operand image: [n_tiles][NS][3][64] x 8 bf16 (null unless requested)
  int n_cu = 256;
  f32x4* W = nullptr;      // [n_tiles + 1][NQ][64] float4, fragment order (one spare tile for the prefetch)
  MItem* items = nullptr;
  float* y0 = nullptr;     // [k_pad]
  int n_items = 0;
  int nkk = 0;             // n_pad / 32
  int identity = 0;
  int n_simd = 1024;       // SIMDs on the device (CUs x 4)
  int waves_per_simd = kMfmaWavesPerSimd;  // resident waves per SIMD the kernel is built for
  int64_t bytes = 0;
};

template <int NKK>
struct MfmaCfg {
  static constexpr int NT = (NKK <= 2) ? RAYEN_MFMA_NT : 1;  // sample tiles per wave
  static constexpr int NQ = NKK * 4;             // k-groups (4 MFMA steps each) per row tile
  static constexpr int KK = NKK * 16;            // MFMA steps per row tile = registers of v per sample tile
};

#ifndef RAYEN_MFMA_WAVES
#define RAYEN_MFMA_WAVES 8
#endif
constexpr int kMfmaWaves = RAYEN_MFMA_WAVES;  // waves per workgroup (independent; the workgroup is only a launch unit)

__device__ __forceinline__ float xhalf(float x) { return __shfl_xor(x, 32); }

template <int NKK, bool TRACK>
__global__ __launch_bounds__(kMfmaWaves * 64, kMfmaWaves / 4) void mfma_fwd_kernel(
    const f32x4* __restrict__ Wimg, const MItem* __restrict__ items, int n_items,
    const MPack* __restrict__ packs, const float* __restrict__ y0, int identity, int k, int n, const float* __restrict__ v, int64_t B,
    int64_t ldv, int vec_in, float* __restrict__ y, int64_t ldy, int vec_out,
    float* __restrict__ kappa_out, int32_t* __restrict__ active_out, int32_t* __restrict__ nan_flag,
    int old_mode) {
  using C = MfmaCfg<NKK>;
  constexpr int NT = C::NT, NQ = C::NQ, KK = C::KK;
  __shared__ float aux_lds[kMfmaWaves][NT][32][32];  // [wave][sample tile][aux row][sample]
  __shared__ __attribute__((aligned(16))) float y0_lds[NKK * 32];  // output offset for the NA_E = I write-out
  // Transposition patch for v and y (one sample tile per wave at a time).  A lane needs 16-byte
  // pieces of ITS sample's row (fragment-shaped access: 32 B per 128-B line per instruction); going
  // through LDS lets every global load/store instruction move four whole rows (1 KiB, full lines).
  // Row stride n_pad + 4 floats keeps both the row-wise and the fragment-wise LDS accesses conflict-free.
  constexpr bool kLines = NKK <= 2;  // LDS budget: 8 waves x 32 x (n_pad + 4) floats next to aux_lds
  constexpr int LSTR = NKK * 32 + 4;
  __shared__ __attribute__((aligned(16))) float line_lds[kLines ? kMfmaWaves : 1][kLines ? 32 : 1][kLines ? LSTR : 4];

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int col = lane & 31;  // sample within the tile
  const int hi = lane >> 5;   // which half of the rows / k pairs this lane holds
  const int64_t n_groups = (B + NT * 32 - 1) / (NT * 32);
  const int64_t wave_id = (int64_t)blockIdx.x * kMfmaWaves + wave;
  const int64_t wave_stride = (int64_t)gridDim.x * kMfmaWaves;
  bool bad = false;
  for (int i = threadIdx.x; i < NKK * 32; i += kMfmaWaves * 64) y0_lds[i] = y0[i];  // y0 is zero-padded
  __syncthreads();  // the only workgroup barrier; from here on the waves are independent

  // persistent walk over groups of NT*32 samples (no workgroup barriers anywhere)
  for (int64_t grp = wave_id; grp < n_groups; grp += wave_stride) {
  const int64_t s_base = grp * (NT * 32);

  // ---- this lane's half of v for each of its samples, as B operands
  float vr[NT][KK];
  bool live[NT];
  const bool lines_in = kLines && vec_in && n == NKK * 32;   // wave-uniform
#pragma unroll
  for (int t = 0; t < NT; ++t) live[t] = (s_base + t * 32 + col) < B;
  if (lines_in) {
    f32x4 piece[NT][NQ];  // piece idx = lane + 64 j of the [32 rows][n/4 pieces] tile: row idx / (n/4)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int j = 0; j < NQ; ++j) {
        const int idx = lane + 64 * j;
        const int64_t s = s_base + t * 32 + idx / (NKK * 8);
        piece[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (s < B) piece[t][j] = *reinterpret_cast<const f32x4*>(v + s * ldv + 4 * (idx % (NKK * 8)));
      }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int j = 0; j < NQ; ++j) {
        const int idx = lane + 64 * j;
        *reinterpret_cast<f32x4*>(&line_lds[wave][idx / (NKK * 8)][4 * (idx % (NKK * 8))]) = piece[t][j];
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(&line_lds[wave][col][8 * q + 4 * hi]);
        vr[t][4 * q + 0] = x[0];
        vr[t][4 * q + 1] = x[1];
        vr[t][4 * q + 2] = x[2];
        vr[t][4 * q + 3] = x[3];
      }
      __builtin_amdgcn_wave_barrier();
    }
  } else {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float* row = v + (live[t] ? (s_base + t * 32 + col) : 0) * ldv;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int c0 = 8 * q + 4 * hi;
        f32x4 x = {0.f, 0.f, 0.f, 0.f};
        if (live[t]) {
          if (vec_in && c0 + 3 < n) {
            x = *reinterpret_cast<const f32x4*>(row + c0);
          } else {
            if (c0 + 0 < n) x[0] = row[c0 + 0];
            if (c0 + 1 < n) x[1] = row[c0 + 1];
            if (c0 + 2 < n) x[2] = row[c0 + 2];
            if (c0 + 3 < n) x[3] = row[c0 + 3];
          }
        }
        vr[t][4 * q + 0] = x[0];
        vr[t][4 * q + 1] = x[1];
        vr[t][4 * q + 2] = x[2];
        vr[t][4 * q + 3] = x[3];
      }
    }
  }

  // RAYEN_old head (rayen/constraint_module.py:460-466): y = y0 + N v / (||v|| e^beta + kappa(v)),
  // beta = column n of the input
  float old_den[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) old_den[t] = 0.f;
  if (old_mode) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float nrm2 = 0.f;
#pragma unroll
      for (int i = 0; i < KK; ++i) nrm2 = fmaf(vr[t][i], vr[t][i], nrm2);
      nrm2 += xhalf(nrm2);
      const float beta = live[t] ? v[(s_base + t * 32 + col) * ldv + n] : 0.f;
      old_den[t] = sqrtf(nrm2) * __expf(beta);   // ||v|| e^beta (0 exactly when v = 0)
    }
  }
  float kap[NT], part[NT], scale[NT];
  int aseg[NT], arow[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) { kap[t] = 0.f; part[t] = 0.f; scale[t] = 1.f; aseg[t] = -1; arow[t] = 0; }

  // A fragments: two whole-tile register buffers.  While the MFMAs of tile t run out of one
  // buffer, the NQ loads of tile t+1 (issued at the top of tile t, a full tile = NQ*4*NT MFMAs
  // ahead) land in the other, so an L2 or Infinity-Cache round trip never reaches the MFMA
  // stream.  The item walk is unrolled by two to keep the buffer choice static; the
  // sched_barrier keeps hipcc from sinking the loads next to their uses.
  const f32x4* wp = Wimg + lane;
  f32x4 buf_a[NQ], buf_b[NQ];
  auto fetch_tile = [&](f32x4 (&buf)[NQ]) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) buf[q] = wp[q * 64];
    wp += NQ * 64;
    __builtin_amdgcn_sched_barrier(0);
  };
  fetch_tile(buf_a);

  // one 32-row tile: NQ k-groups of 4 MFMA steps on every sample tile
  auto run_tile = [&](f32x16 (&acc)[NT], const f32x4 (&a)[NQ], const int qbegin) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[t][g] = 0.f;
#pragma unroll
    for (int qb = 0; qb < NKK; ++qb) {  // one 32-column block = 4 k-groups
      if (4 * qb < qbegin) continue;    // wave-uniform: the block was folded into its transpose
#pragma unroll
      for (int q = 4 * qb; q < 4 * qb + 4; ++q)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int t = 0; t < NT; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][c], vr[t][4 * q + c], acc[t], 0, 0, 0);
    }
  };

  auto finish_kappa = [&]() {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float other = xhalf(kap[t]);
      if (TRACK) {
        const int oseg = __shfl_xor(aseg[t], 32), orow = __shfl_xor(arow[t], 32);
        // deterministic tie-break so both halves agree
        if (other > kap[t] || (other == kap[t] && hi == 1)) { aseg[t] = oseg; arow[t] = orow; }
      }
      kap[t] = fmaxf(kap[t], other);
      scale[t] = 1.0f / fmaxf(1.0f, kap[t]);
      if (old_mode) scale[t] = old_den[t] > 0.f ? 1.0f / (old_den[t] + kap[t]) : 0.f;
    }
  };

  f32x16 acc[NT];
  auto process = [&](const MItem item, const f32x4 (&a)[NQ]) {
    if (item.type == MI_NOP) return;  // pairing filler: no MFMAs, no epilogue
    // rows of NA_E come last: kappa is final once the first of those tiles is reached
    if (item.type == MI_OUT && (item.flags & MF_FIRST)) finish_kappa();
    run_tile(acc, a, item.qbegin);
    if (item.type == MI_LIN) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (TRACK) {
#pragma unroll
          for (int g = 0; g < 16; ++g)
            if (acc[t][g] > kap[t]) {
              kap[t] = acc[t][g];
              aseg[t] = item.seg;
              arow[t] = item.row0 + (g & 3) + 8 * (g >> 2) + 4 * hi;
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
    } else if (item.type == MI_OUT) {
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
    } else if (item.type == MI_PACK) {
      // eight small factor segments in one tile: ||U v||^2 of each is a 4-register sum
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
          const float kc = aux_lds[wave][t][slot & 31][col] + sqrtf(qs);
          if (sid >= 0 && kc > kap[t]) { kap[t] = kc; aseg[t] = sid; arow[t] = 0; }
        }
      }
    } else {
      // QSYM / QFAC / SOC: a running sum over the segment's tiles, closed on its last tile
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        float sum = (item.flags & MF_FIRST) ? 0.f : part[t];
        if (item.flags & MF_SYM) {
          // radicand v'Gv = sum_j (G v)_j v_j ; v_j of row tile tp is register 16*tp+g of vr
#pragma unroll
          for (int tp = 0; tp < NKK; ++tp)
            if (item.row0 == tp) {
#pragma unroll
              for (int g = 0; g < 16; ++g) sum = fmaf(acc[t][g], vr[t][16 * tp + g], sum);
            }
        } else {
#pragma unroll
          for (int g = 0; g < 16; ++g) sum = fmaf(acc[t][g], acc[t][g], sum);
        }
        part[t] = sum;
      }
      if (item.flags & MF_LAST) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const float total = part[t] + xhalf(part[t]);
          const float a0 = aux_lds[wave][t][item.aux][col];
          float kc;
          if (item.type != MI_SOC) {
            kc = a0 + sqrtf(fmaxf(total, 0.f));
          } else {
            // a' x^2 + b' x + c' = 0  (rayen/constraint_module.py:392-396, 339-348), a' < 0
            const float br = aux_lds[wave][t][item.aux + 1][col];
            const float cp = total - a0 * a0;
            const float bp = 2.f * br - 2.f * a0 * item.f0;
            const float disc = bp * bp - 4.f * item.f1 * cp;
            kc = 0.f;
            if (disc >= 0.f) {
              const float root = sqrtf(disc);
              const float inv2a = 0.5f / item.f1;
              kc = fmaxf((-bp - root) * inv2a, (-bp + root) * inv2a);
            }
          }
          if (kc > kap[t]) { kap[t] = kc; aseg[t] = item.seg; arow[t] = 0; }
        }
      }
    }
  };
  for (int it = 0; it < n_items; it += 2) {  // n_items is even (padded with a no-op tile)
    fetch_tile(buf_b);
    process(items[it], buf_a);
    fetch_tile(buf_a);
    process(items[it + 1], buf_b);
  }

  if (identity) {
    finish_kappa();
    // y = y0 + v / max(1, kappa), straight from the B-operand registers.  y0 comes from LDS (a
    // global load per piece would put an L2 round trip in front of every store).
    if (kLines && vec_out && k == NKK * 32) {  // wave-uniform: full-line stores through the LDS patch
#pragma unroll
      for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const f32x4 off = *reinterpret_cast<const f32x4*>(&y0_lds[8 * q + 4 * hi]);
          f32x4 o;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            o[c] = fmaf(vr[t][4 * q + c], scale[t], off[c]);
            bad |= live[t] && (o[c] != o[c]);
          }
          *reinterpret_cast<f32x4*>(&line_lds[wave][col][8 * q + 4 * hi]) = o;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
          const int idx = lane + 64 * j;
          const int64_t s = s_base + t * 32 + idx / (NKK * 8);
          const f32x4 o = *reinterpret_cast<const f32x4*>(&line_lds[wave][idx / (NKK * 8)][4 * (idx % (NKK * 8))]);
          if (s < B) *reinterpret_cast<f32x4*>(y + s * ldy + 4 * (idx % (NKK * 8))) = o;
        }
        __builtin_amdgcn_wave_barrier();
      }
    } else if (vec_out && (k & 3) == 0) {  // whole float4 pieces, no per-piece branches
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (!live[t]) continue;
        float* yrow = y + (s_base + t * 32 + col) * ldy;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int c0 = 8 * q + 4 * hi;
          const f32x4 off = *reinterpret_cast<const f32x4*>(&y0_lds[c0]);
          f32x4 o;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            o[c] = fmaf(vr[t][4 * q + c], scale[t], off[c]);
            bad |= (o[c] != o[c]);
          }
          if (c0 < k) *reinterpret_cast<f32x4*>(yrow + c0) = o;
        }
      }
    } else {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (!live[t]) continue;
        float* yrow = y + (s_base + t * 32 + col) * ldy;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int c0 = 8 * q + 4 * hi;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float o = fmaf(vr[t][4 * q + c], scale[t], y0_lds[c0 + c]);
            if (c0 + c < k) { bad |= (o != o); yrow[c0 + c] = o; }
          }
        }
      }
    }
  }

  if (hi == 0) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (!live[t]) continue;
      const int64_t s = s_base + t * 32 + col;
      if (kappa_out) kappa_out[s] = kap[t];
      if (TRACK) { active_out[2 * s] = aseg[t]; active_out[2 * s + 1] = arow[t]; }
    }
  }
  }  // persistent loop over sample groups
  if (nan_flag && bad) atomicOr(nan_flag, 1);
}

// ---------------------------------------------------------------------------------------------
// Split-operand variant: the same tile walk on v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate).
// Every fp32 operand is split exactly into three bf16 pieces x = x1 + x2 + x3 (8 significant bits
// each, bf16 has fp32's exponent range, so no scaling is involved); the product is rebuilt from
// the six piece products of order <= 2^-16 (x1y1, x1y2, x2y1, x1y3, x2y2, x3y1), each exact in the
// fp32 accumulator.  The dropped terms are <= 2^-24 relative, i.e. below the rounding error of an
// fp32 FMA chain: the result is fp32-grade at 6/16 of the fp32 MFMA time.  A operands (three bf16
// images of W, 12 KiB per tile at n = 64) are staged through LDS once per workgroup per tile --
// the workgroup's waves walk the tiles in lockstep, one barrier per tile (two 4-wave workgroups
// share a CU so that one can compute while the other sits at a barrier or a group boundary) --
// because at this MFMA rate the per-wave L2 stream of the fp32 kernel would exceed the L1 bandwidth.
// EXPERIMENTAL (opt-in with RAYEN_SPLIT_BF16=1 at pack creation): parity-tested like the fp32
// kernel, but with the matrix work 2.7x cheaper the group boundaries and epilogues dominate and the
// measured gain is only 5-25 %; DESIGN.md lists what it takes to turn it into the default.
// ---------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#ifndef RAYEN_SPLIT_WAVES
#define RAYEN_SPLIT_WAVES 4
#endif
constexpr int kSplitWaves = RAYEN_SPLIT_WAVES;
constexpr int kSplitStage = (12 + kSplitWaves - 1) / kSplitWaves;  // chunks a wave stages per tile (<= 12 chunks)

template <int NKK, bool TRACK>
__global__ __launch_bounds__(kSplitWaves * 64, 2) void mfma_split_kernel(
    const bf16x8* __restrict__ Wb, const MItem* __restrict__ items, int n_items,
    const MPack* __restrict__ packs, const float* __restrict__ y0, int identity, int k, int n,
    const float* __restrict__ v, int64_t B, int64_t ldv, int vec_in, float* __restrict__ y, int64_t ldy,
    int vec_out, float* __restrict__ kappa_out, int32_t* __restrict__ active_out,
    int32_t* __restrict__ nan_flag) {
  constexpr int NT = 2, NQ = NKK * 4, NS = NKK * 2, NCH = NS * 3;  // NS K-steps of 16, NCH 1-KiB chunks per tile
  constexpr int kMfmaWaves = kSplitWaves;                          // (the shared epilogue text indexes aux_lds by wave)
  __shared__ float aux_lds[kSplitWaves][NT][32][32];               // [wave][sample tile][aux row][sample]
  __shared__ bf16x8 a_lds[2][NCH][64];                             // two tiles of A fragments, [chunk][lane]

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int col = lane & 31;
  const int hi = lane >> 5;
  const int64_t n_groups = (B + NT * 32 - 1) / (NT * 32);
  const int64_t groups_per_round = (int64_t)gridDim.x * kSplitWaves;
  const int64_t rounds = (n_groups + groups_per_round - 1) / groups_per_round;
  bool bad = false;
  // every wave of a workgroup runs the same number of rounds (the tile loop holds barriers);
  // a wave whose group index is past the end just carries dead samples
  for (int64_t round = 0; round < rounds; ++round) {
  const int64_t grp = (round * gridDim.x + blockIdx.x) * kSplitWaves + wave;
  const int64_t s_base = grp * (NT * 32);

  // ---- this lane's half of v: fp32 (epilogues, output) and split into bf16 pieces
  // vb[t][piece][k-step] = 8 elements = the B operand of one MFMA
  float vr[NT][NKK * 16];
  bf16x8 vb[NT][3][NS];
  bool live[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int64_t s = s_base + t * 32 + col;
    live[t] = s < B;
    const float* row = v + (live[t] ? s : 0) * ldv;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int c0 = 8 * q + 4 * hi;
      f32x4 x = {0.f, 0.f, 0.f, 0.f};
      if (live[t]) {
        if (vec_in && c0 + 3 < n) {
          x = *reinterpret_cast<const f32x4*>(row + c0);
        } else {
          if (c0 + 0 < n) x[0] = row[c0 + 0];
          if (c0 + 1 < n) x[1] = row[c0 + 1];
          if (c0 + 2 < n) x[2] = row[c0 + 2];
          if (c0 + 3 < n) x[3] = row[c0 + 3];
        }
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int i = (q & 1) * 4 + c;  // position inside K-step q >> 1
        const __bf16 p1 = (__bf16)x[c];
        const float r1 = x[c] - (float)p1;
        const __bf16 p2 = (__bf16)r1;
        const float r2 = r1 - (float)p2;
        vr[t][4 * q + c] = x[c];
        vb[t][0][q >> 1][i] = p1;
        vb[t][1][q >> 1][i] = p2;
        vb[t][2][q >> 1][i] = (__bf16)r2;
      }
    }
  }
  auto vget = [&](int t, int idx) -> float { return vr[t][idx]; };

  float kap[NT], part[NT], scale[NT];
  int aseg[NT], arow[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) { kap[t] = 0.f; part[t] = 0.f; scale[t] = 1.f; aseg[t] = -1; arow[t] = 0; }

  // ---- A staging: wave w carries chunks w, w + #waves, ... of a tile from global memory into LDS
  bf16x8 stage[kSplitStage];
  auto stage_load = [&](int tile) {
    const bf16x8* src = Wb + (size_t)tile * NCH * 64 + lane;
#pragma unroll
    for (int j = 0; j < kSplitStage; ++j)
      if (wave + kSplitWaves * j < NCH) stage[j] = src[(size_t)(wave + kSplitWaves * j) * 64];
  };
  auto stage_store = [&](int buf) {
#pragma unroll
    for (int j = 0; j < kSplitStage; ++j)
      if (wave + kSplitWaves * j < NCH) a_lds[buf][wave + kSplitWaves * j][lane] = stage[j];
  };
  stage_load(0);
  stage_store(0);
  if (n_items > 1) stage_load(1);
  __syncthreads();

  auto finish_kappa = [&]() {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float other = xhalf(kap[t]);
      if (TRACK) {
        const int oseg = __shfl_xor(aseg[t], 32), orow = __shfl_xor(arow[t], 32);
        if (other > kap[t] || (other == kap[t] && hi == 1)) { aseg[t] = oseg; arow[t] = orow; }
      }
      kap[t] = fmaxf(kap[t], other);
      scale[t] = 1.0f / fmaxf(1.0f, kap[t]);
    }
  };

  f32x16 acc[NT];
  for (int it = 0; it < n_items; ++it) {
    const int buf = it & 1;
    // tile it+1 (loaded into `stage` one tile ago) goes to the other LDS buffer, whose last readers
    // all passed the barrier that closed tile it-1; tile it+2 starts its trip from L2
    if (it + 1 < n_items) stage_store(buf ^ 1);
    if (it + 2 < n_items) stage_load(it + 2);
    const MItem item = items[it];
    if (item.type != MI_NOP) {
    if (item.type == MI_OUT && (item.flags & MF_FIRST)) finish_kappa();
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[t][g] = 0.f;
#pragma unroll
    for (int sp = 0; sp < NS; ++sp) {
      if (2 * sp < item.qbegin) continue;  // 32-column blocks folded into their transpose (wave-uniform)
      const bf16x8 a1 = a_lds[buf][sp * 3 + 0][lane];
      const bf16x8 a2 = a_lds[buf][sp * 3 + 1][lane];
      const bf16x8 a3 = a_lds[buf][sp * 3 + 2][lane];
      // smallest products first
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, vb[t][0][sp], acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, vb[t][1][sp], acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, vb[t][2][sp], acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, vb[t][0][sp], acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, vb[t][1][sp], acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, vb[t][0][sp], acc[t], 0, 0, 0);
    }
    if (item.type == MI_LIN) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (TRACK) {
#pragma unroll
          for (int g = 0; g < 16; ++g)
            if (acc[t][g] > kap[t]) {
              kap[t] = acc[t][g];
              aseg[t] = item.seg;
              arow[t] = item.row0 + (g & 3) + 8 * (g >> 2) + 4 * hi;
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
    } else if (item.type == MI_OUT) {
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
    } else if (item.type == MI_PACK) {
      // eight small factor segments in one tile: ||U v||^2 of each is a 4-register sum
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
          const float kc = aux_lds[wave][t][slot & 31][col] + sqrtf(qs);
          if (sid >= 0 && kc > kap[t]) { kap[t] = kc; aseg[t] = sid; arow[t] = 0; }
        }
      }
    } else {
      // QSYM / QFAC / SOC: a running sum over the segment's tiles, closed on its last tile
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        float sum = (item.flags & MF_FIRST) ? 0.f : part[t];
        if (item.flags & MF_SYM) {
          // radicand v'Gv = sum_j (G v)_j v_j ; v_j of row tile tp is register 16*tp+g of vr
#pragma unroll
          for (int tp = 0; tp < NKK; ++tp)
            if (item.row0 == tp) {
#pragma unroll
              for (int g = 0; g < 16; ++g) sum = fmaf(acc[t][g], vget(t, 16 * tp + g), sum);
            }
        } else {
#pragma unroll
          for (int g = 0; g < 16; ++g) sum = fmaf(acc[t][g], acc[t][g], sum);
        }
        part[t] = sum;
      }
      if (item.flags & MF_LAST) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const float total = part[t] + xhalf(part[t]);
          const float a0 = aux_lds[wave][t][item.aux][col];
          float kc;
          if (item.type != MI_SOC) {
            kc = a0 + sqrtf(fmaxf(total, 0.f));
          } else {
            // a' x^2 + b' x + c' = 0  (rayen/constraint_module.py:392-396, 339-348), a' < 0
            const float br = aux_lds[wave][t][item.aux + 1][col];
            const float cp = total - a0 * a0;
            const float bp = 2.f * br - 2.f * a0 * item.f0;
            const float disc = bp * bp - 4.f * item.f1 * cp;
            kc = 0.f;
            if (disc >= 0.f) {
              const float root = sqrtf(disc);
              const float inv2a = 0.5f / item.f1;
              kc = fmaxf((-bp - root) * inv2a, (-bp + root) * inv2a);
            }
          }
          if (kc > kap[t]) { kap[t] = kc; aseg[t] = item.seg; arow[t] = 0; }
        }
      }
    }
    }  // not a filler tile
    __syncthreads();  // everyone is done reading a_lds[buf] and writing a_lds[buf ^ 1]
  }

  if (identity) {
    finish_kappa();
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (!live[t]) continue;
      float* yrow = y + (s_base + t * 32 + col) * ldy;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int c0 = 8 * q + 4 * hi;
        if (c0 >= k) continue;
        f32x4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          o[c] = fmaf(vget(t, 4 * q + c), scale[t], y0[c0 + c]);
          bad |= (o[c] != o[c]) && (c0 + c < k);
        }
        if (vec_out && c0 + 3 < k) {
          *reinterpret_cast<f32x4*>(yrow + c0) = o;
        } else {
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (c0 + c < k) yrow[c0 + c] = o[c];
        }
      }
    }
  }

  if (hi == 0) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (!live[t]) continue;
      const int64_t s = s_base + t * 32 + col;
      if (kappa_out) kappa_out[s] = kap[t];
      if (TRACK) { active_out[2 * s] = aseg[t]; active_out[2 * s + 1] = arow[t]; }
    }
  }
  }  // rounds
  if (nan_flag && bad) atomicOr(nan_flag, 1);
}

// ---------------------------------------------------------------------------------------------
// host: eligibility, image construction, launch
// ---------------------------------------------------------------------------------------------


bool mfma_eligible(const RayenPack* p) {
  if (p->n > 128) return false;  // v lives in registers: n_pad/2 VGPRs per sample tile
  for (const RayenSegment& g : p->segs)
    if (g.type == RAYEN_SEG_LMI) return false;  // eigen-solve epilogue lives on the generic path
  TileLayout b(p->n);
  if (layout_tiles(p, b, /*allow_pack=*/true) != RAYEN_OK || b.items.empty()) return false;
  // 32-row tiles must be reasonably full, and columns not mostly padding; otherwise the
  // 8-row generic path wastes less
  const int64_t padded = (int64_t)b.items.size() * 32;
  return b.useful_rows * 2 >= padded && p->n * 2 >= b.n_pad;
}

int mfma_build(const RayenPack* p, MfmaImage** out, int64_t* bytes) {
  TileLayout b(p->n);
  const int rc = layout_tiles(p, b, /*allow_pack=*/true);
  if (rc != RAYEN_OK) return rc;
  if (b.items.size() % 2) {  // the kernel walks tiles in pairs
    MItem it;
    std::memset(&it, 0, sizeof(it));
    it.type = MI_NOP;
    b.items.push_back(it);
    b.add_tile({}, p->n);
  }
  b.add_tile({}, p->n);  // spare tile: the prefetch runs one tile past the end (no item refers to it)
  const std::vector<float> frag = b.fragments_f32();
  if (b.packs.empty()) b.packs.push_back(MPack());

  MfmaImage* img = new MfmaImage();
  img->nkk = b.n_pad / 32;
  img->identity = p->out_identity;
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, p->device) == hipSuccess && prop.multiProcessorCount > 0) {
      img->n_simd = prop.multiProcessorCount * 4;
      img->n_cu = prop.multiProcessorCount;
    }
  }
  img->n_items = (int)b.items.size();
  const int k_tiles = (p->k + 31) / 32;
  std::vector<float> y0((size_t)k_tiles * 32 + 32, 0.f);
  for (int i = 0; i < p->k; ++i) y0[i] = (float)p->y0[i];
  bool ok = hipMalloc(&img->W, frag.size() * sizeof(float)) == hipSuccess &&
            hipMemcpy(img->W, frag.data(), frag.size() * sizeof(float), hipMemcpyHostToDevice) == hipSuccess &&
            hipMalloc(&img->y0, y0.size() * sizeof(float)) == hipSuccess &&
            hipMemcpy(img->y0, y0.data(), y0.size() * sizeof(float), hipMemcpyHostToDevice) == hipSuccess &&
            hipMalloc(&img->items, b.items.size() * sizeof(MItem)) == hipSuccess &&
            hipMemcpy(img->items, b.items.data(), b.items.size() * sizeof(MItem), hipMemcpyHostToDevice) == hipSuccess &&
            hipMalloc(&img->packs, b.packs.size() * sizeof(MPack)) == hipSuccess &&
            hipMemcpy(img->packs, b.packs.data(), b.packs.size() * sizeof(MPack), hipMemcpyHostToDevice) == hipSuccess;
  if (ok && p->split_bf16 && img->nkk <= 2) {
    // three bf16 pieces of every entry, in the fragment order of v_mfma_f32_32x32x16_bf16:
    // chunk (tile, k-step s, piece p) = 64 lanes x 8 elements, element i of lane l = column
    // 16s + 8(i>>2) + 4(l>>5) + (i&3) of row l&31, i.e. entry [2s + (i>>2)][l][i&3] of the fp32 image
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
    const int n_tiles = (int)b.items.size(), ns = b.nq() / 2;
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
    ok = hipMalloc(&img->Wb, wb.size() * 2) == hipSuccess &&
         hipMemcpy(img->Wb, wb.data(), wb.size() * 2, hipMemcpyHostToDevice) == hipSuccess;
    if (ok) img->bytes += (int64_t)wb.size() * 2;
  }
  if (!ok) { mfma_free(img); return RAYEN_E_ALLOC; }
  img->bytes += (int64_t)(frag.size() * sizeof(float) + y0.size() * sizeof(float) +
                         b.items.size() * sizeof(MItem) + b.packs.size() * sizeof(MPack));
  *bytes = img->bytes;
  *out = img;
  return RAYEN_OK;
}

void mfma_free(MfmaImage* img) {
  if (img == nullptr) return;
  if (img->W) (void)hipFree(img->W);
  if (img->items) (void)hipFree(img->items);
  if (img->packs) (void)hipFree(img->packs);
  if (img->Wb) (void)hipFree(img->Wb);
  if (img->y0) (void)hipFree(img->y0);
  delete img;
}

template <int NKK>
static int launch_mfma(const RayenPack* p, const MfmaImage* img, const float* v, int64_t B, int64_t ldv,
                       float* y, int64_t ldy, float* kappa, int32_t* active, int32_t* nan_flag,
                       int old_mode, hipStream_t stream) {
  // persistent waves: at most `slots` waves are resident (VGPR-limited waves per SIMD x SIMDs);
  // give every wave the same number of sample groups so that no SIMD idles in a ragged last round
  constexpr int per_wave = MfmaCfg<NKK>::NT * 32;
  const int64_t n_groups = (B + per_wave - 1) / per_wave;
  const int64_t slots = (int64_t)img->n_simd * img->waves_per_simd;
  const int64_t rounds = (n_groups + slots - 1) / slots;
  const int64_t waves = (n_groups + rounds - 1) / rounds;
  const int64_t grid = (waves + kMfmaWaves - 1) / kMfmaWaves;
  const int vec_in = (ldv % 4 == 0) && ((reinterpret_cast<uintptr_t>(v) & 15) == 0);
  const int vec_out = (ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0);
  if (active != nullptr) {
    hipLaunchKernelGGL((mfma_fwd_kernel<NKK, true>), dim3((unsigned)grid), dim3(kMfmaWaves * 64), 0, stream,
                       img->W, img->items, img->n_items, img->packs, img->y0, img->identity, p->k, p->n, v, B, ldv,
                       vec_in, y, ldy, vec_out, kappa, active, nan_flag, old_mode);
  } else {
    hipLaunchKernelGGL((mfma_fwd_kernel<NKK, false>), dim3((unsigned)grid), dim3(kMfmaWaves * 64), 0, stream,
                       img->W, img->items, img->n_items, img->packs, img->y0, img->identity, p->k, p->n, v, B, ldv,
                       vec_in, y, ldy, vec_out, kappa, active, nan_flag, old_mode);
  }
  return hipGetLastError() == hipSuccess ? RAYEN_OK : RAYEN_E_LAUNCH;
}

template <int NKK>
static int launch_split(const RayenPack* p, const MfmaImage* img, const float* v, int64_t B, int64_t ldv,
                        float* y, int64_t ldy, float* kappa, int32_t* active, int32_t* nan_flag,
                        hipStream_t stream) {
  // one workgroup (8 waves x 64 samples) per CU, every workgroup the same number of rounds
  const int64_t blocks_needed = (B + kSplitWaves * 64 - 1) / (kSplitWaves * 64);
  const int64_t slots = (int64_t)img->n_cu * (8 / kSplitWaves);  // two 4-wave workgroups share a CU
  const int64_t rounds = (blocks_needed + slots - 1) / slots;
  const int64_t grid = (blocks_needed + rounds - 1) / rounds;
  const int vec_in = (ldv % 4 == 0) && ((reinterpret_cast<uintptr_t>(v) & 15) == 0);
  const int vec_out = (ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0);
  const bf16x8* wb = static_cast<const bf16x8*>(img->Wb);
  if (active != nullptr) {
    hipLaunchKernelGGL((mfma_split_kernel<NKK, true>), dim3((unsigned)grid), dim3(kSplitWaves * 64), 0, stream,
                       wb, img->items, img->n_items, img->packs, img->y0, img->identity, p->k, p->n, v, B, ldv,
                       vec_in, y, ldy, vec_out, kappa, active, nan_flag);
  } else {
    hipLaunchKernelGGL((mfma_split_kernel<NKK, false>), dim3((unsigned)grid), dim3(kSplitWaves * 64), 0, stream,
                       wb, img->items, img->n_items, img->packs, img->y0, img->identity, p->k, p->n, v, B, ldv,
                       vec_in, y, ldy, vec_out, kappa, active, nan_flag);
  }
  return hipGetLastError() == hipSuccess ? RAYEN_OK : RAYEN_E_LAUNCH;
}

int mfma_forward(const RayenPack* p, const MfmaImage* img, const float* v, int64_t B, int64_t ldv, float* y,
                 int64_t ldy, float* kappa, int32_t* active, int32_t* nan_flag, int old_mode,
                 hipStream_t stream) {
  if (B == 0) return RAYEN_OK;
  if (img->Wb != nullptr && !old_mode) {
    if (img->nkk == 1) return launch_split<1>(p, img, v, B, ldv, y, ldy, kappa, active, nan_flag, stream);
    if (img->nkk == 2) return launch_split<2>(p, img, v, B, ldv, y, ldy, kappa, active, nan_flag, stream);
  }
  switch (img->nkk) {
    case 1: return launch_mfma<1>(p, img, v, B, ldv, y, ldy, kappa, active, nan_flag, old_mode, stream);
    case 2: return launch_mfma<2>(p, img, v, B, ldv, y, ldy, kappa, active, nan_flag, old_mode, stream);
    case 3: return launch_mfma<3>(p, img, v, B, ldv, y, ldy, kappa, active, nan_flag, old_mode, stream);
    case 4: return launch_mfma<4>(p, img, v, B, ldv, y, ldy, kappa, active, nan_flag, old_mode, stream);
    default: return RAYEN_E_UNSUPPORTED;
  }
}

}  // namespace rayen
