// fp32 MFMA forward kernel (shared by rayen_mfma.hip and rayen_mfma_mapped.hip).
//
// T = W . V' on v_mfma_f32_32x32x2_f32 with the whole epilogue in registers.
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
//     those registers, again as float4s;
//   * the result registers of one MFMA chain ARE valid B operands of the next: with NKX > 0 the
//     kernel first evaluates the module's mapper v = W_m x + b_m (rayen/constraint_module.py:259-263,
//     525) on the same instruction and feeds the result straight into the constraint walk -- v never
//     exists in HBM unless the caller asks for it (training).
// W is stored in fragment order ([tile][k-group][lane] float4): each A fetch is
// one contiguous 1 KiB global_load_dwordx4 per wave, served by L2 (the image is
// <= a few hundred KiB and shared by every wave), prefetched a whole tile ahead into a
// second register buffer.  No LDS staging of W, no workgroup barriers in the tile walk.
//
// HBM traffic per sample: n loads + k stores, the algorithmic minimum.
#pragma once

#include <type_traits>

#include "rayen_internal.h"
#include "rayen_tiles.h"

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

// The mapper of the module, v = Wm x + bias, evaluated in front of the constraint walk (NKX > 0 kernels).
struct MapperArgs {
  const float* w = nullptr;     // [n, in_dim] row-major (torch.nn.Linear.weight), 16-byte aligned rows
  int64_t ldw = 0;
  const float* bias = nullptr;  // [n] or null
  int in_dim = 0;               // multiple of 4, <= 32 NKX
  float* v_out = nullptr;       // [B, ldvo] receives v (the backward needs it) or null
  int64_t ldvo = 0;
};

__device__ __forceinline__ float xhalf(float x) { return __shfl_xor(x, 32); }

// Rows of a [B, ld] matrix -> B-operand registers: dst[t][4q + c] = src[sample][8q + 4hi + c].
// LINES: every global load instruction moves four whole rows (full 128-B lines) and the fragment
// shape is produced by a trip through the wave's LDS patch (row stride LSTR floats).
// STREAM: the rows are read once -- non-temporal loads (they do not displace the W image in the caches).
template <int NT, int NK, int LSTR, bool LINES, bool STREAM = false>
__device__ __forceinline__ void load_rows(float (&dst)[NT][NK * 16], const float* __restrict__ src, int64_t ld,
                                          int width, int vec, int64_t s_base, int64_t B,
                                          const bool (&live_t)[NT], float (*patch)[LSTR], int lane) {
  constexpr int NQ = NK * 4;
  const int col = lane & 31, hi = lane >> 5;
  const bool lines = LINES && vec && width == NK * 32;  // wave-uniform
  if (lines) {
    f32x4 piece[NT][NQ];  // piece idx = lane + 64 j of the [32 rows][width/4 pieces] tile: row idx / (width/4)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int j = 0; j < NQ; ++j) {
        const int idx = lane + 64 * j;
        const int64_t s = s_base + t * 32 + idx / (NK * 8);
        piece[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (s < B) {
          const f32x4* from = reinterpret_cast<const f32x4*>(src + s * ld + 4 * (idx % (NK * 8)));
          if constexpr (STREAM) piece[t][j] = __builtin_nontemporal_load(from);
          else piece[t][j] = *from;
        }
      }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int j = 0; j < NQ; ++j) {
        const int idx = lane + 64 * j;
        *reinterpret_cast<f32x4*>(&patch[idx / (NK * 8)][4 * (idx % (NK * 8))]) = piece[t][j];
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(&patch[col][8 * q + 4 * hi]);
        dst[t][4 * q + 0] = x[0];
        dst[t][4 * q + 1] = x[1];
        dst[t][4 * q + 2] = x[2];
        dst[t][4 * q + 3] = x[3];
      }
      __builtin_amdgcn_wave_barrier();
    }
  } else {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const bool live = live_t[t];
      const float* row = src + (live ? (s_base + t * 32 + col) : 0) * ld;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int c0 = 8 * q + 4 * hi;
        f32x4 x = {0.f, 0.f, 0.f, 0.f};
        if (live) {
          if (vec && c0 + 3 < width) {
            x = *reinterpret_cast<const f32x4*>(row + c0);
          } else {
            if (c0 + 0 < width) x[0] = row[c0 + 0];
            if (c0 + 1 < width) x[1] = row[c0 + 1];
            if (c0 + 2 < width) x[2] = row[c0 + 2];
            if (c0 + 3 < width) x[3] = row[c0 + 3];
          }
        }
        dst[t][4 * q + 0] = x[0];
        dst[t][4 * q + 1] = x[1];
        dst[t][4 * q + 2] = x[2];
        dst[t][4 * q + 3] = x[3];
      }
    }
  }
}

// The inverse trip: dst[sample][8q + 4hi + c] = off[8q + 4hi + c] + scale[t] * val[t][4q + c] for the
// first `width` columns (`off` = an LDS vector padded to NK*32, or null).  Returns true when a NaN was
// written (the fused form of rayen/constraint_module.py:531).
// STREAM: non-temporal stores (the output is not read again by this kernel).
template <int NT, int NK, int LSTR, bool LINES, bool STREAM = false>
__device__ __forceinline__ bool store_rows(const float (&val)[NT][NK * 16], const float (&scale)[NT],
                                           const float* off, float* __restrict__ dst, int64_t ld, int width,
                                           int vec, int64_t s_base, int64_t B, const bool (&live_t)[NT],
                                           float (*patch)[LSTR], int lane) {
  constexpr int NQ = NK * 4;
  const int col = lane & 31, hi = lane >> 5;
  bool bad = false;
  if (LINES && vec && width == NK * 32) {  // wave-uniform: full-line stores through the LDS patch
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const bool live = live_t[t];
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        f32x4 o4 = {0.f, 0.f, 0.f, 0.f};
        if (off != nullptr) o4 = *reinterpret_cast<const f32x4*>(&off[8 * q + 4 * hi]);
        f32x4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          o[c] = fmaf(val[t][4 * q + c], scale[t], o4[c]);
          bad |= live && (o[c] != o[c]);
        }
        *reinterpret_cast<f32x4*>(&patch[col][8 * q + 4 * hi]) = o;
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int j = 0; j < NQ; ++j) {
        const int idx = lane + 64 * j;
        const int64_t s = s_base + t * 32 + idx / (NK * 8);
        const f32x4 o = *reinterpret_cast<const f32x4*>(&patch[idx / (NK * 8)][4 * (idx % (NK * 8))]);
        if (s < B) {
          f32x4* to = reinterpret_cast<f32x4*>(dst + s * ld + 4 * (idx % (NK * 8)));
          if constexpr (STREAM) __builtin_nontemporal_store(o, to);
          else *to = o;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  } else if (vec && (width & 3) == 0) {  // whole float4 pieces, no per-piece branches
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (!live_t[t]) continue;
      float* row = dst + (s_base + t * 32 + col) * ld;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int c0 = 8 * q + 4 * hi;
        f32x4 o4 = {0.f, 0.f, 0.f, 0.f};
        if (off != nullptr) o4 = *reinterpret_cast<const f32x4*>(&off[c0]);
        f32x4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          o[c] = fmaf(val[t][4 * q + c], scale[t], o4[c]);
          bad |= (o[c] != o[c]);
        }
        if (c0 < width) *reinterpret_cast<f32x4*>(row + c0) = o;
      }
    }
  } else {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (!live_t[t]) continue;
      float* row = dst + (s_base + t * 32 + col) * ld;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int c0 = 8 * q + 4 * hi;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float o = fmaf(val[t][4 * q + c], scale[t], off != nullptr ? off[c0 + c] : 0.f);
          if (c0 + c < width) { bad |= (o != o); row[c0 + c] = o; }
        }
      }
    }
  }
  return bad;
}

// STAGED: the NA_E output tiles leave through an LDS transposition (row-coalesced stores).  Packs with
// NA_E = I never execute that code and are launched with STAGED = false, which keeps their instance
// exactly the code the headline numbers were tuned on (hipcc's register allocation of this kernel is
// sensitive to code it never runs).
template <int NKK, int NKX, bool TRACK, bool STAGED>
__global__ __launch_bounds__(kMfmaWaves * 64, kMfmaWaves / 4) void mfma_fwd_kernel(
    const f32x4* __restrict__ Wimg, const MItem* __restrict__ items, int n_items,
    const MPack* __restrict__ packs, const float* __restrict__ y0, int identity, int k, int n, const float* __restrict__ v, int64_t B,
    int64_t ldv, int vec_in, float* __restrict__ y, int64_t ldy, int vec_out,
    float* __restrict__ kappa_out, int32_t* __restrict__ active_out, int32_t* __restrict__ nan_flag,
    int old_mode, MapperArgs mp) {
  using C = MfmaCfg<NKK>;
  constexpr int NT = C::NT, NQ = C::NQ, KK = C::KK;
  __shared__ float aux_lds[kMfmaWaves][NT][32][32];  // [wave][sample tile][aux row][sample]
  __shared__ __attribute__((aligned(16))) float y0_lds[NKK * 32];  // output offset for the NA_E = I write-out
  __shared__ __attribute__((aligned(16))) float bias_lds[NKX > 0 ? NKK * 32 : 4];  // mapper bias, zero-padded
  // Transposition patch for v and y (one sample tile per wave at a time).  A lane needs 16-byte
  // pieces of ITS sample's row (fragment-shaped access: 32 B per 128-B line per instruction); going
  // through LDS lets every global load/store instruction move four whole rows (1 KiB, full lines).
  // Row stride n_pad + 4 floats keeps both the row-wise and the fragment-wise LDS accesses conflict-free.
  constexpr int NKL = NKK > NKX ? NKK : NKX;
  constexpr bool kLines = NKL <= 2;  // LDS budget: 8 waves x 32 x (n_pad + 4) floats next to aux_lds
  constexpr int LSTR = NKL * 32 + 4;
  __shared__ __attribute__((aligned(16))) float line_lds[kLines ? kMfmaWaves : 1][kLines ? 32 : 1][LSTR];

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int col = lane & 31;  // sample within the tile
  const int hi = lane >> 5;   // which half of the rows / k pairs this lane holds
  const int64_t n_groups = (B + NT * 32 - 1) / (NT * 32);
  const int64_t wave_id = (int64_t)blockIdx.x * kMfmaWaves + wave;
  const int64_t wave_stride = (int64_t)gridDim.x * kMfmaWaves;
  bool bad = false;
  for (int i = threadIdx.x; i < NKK * 32; i += kMfmaWaves * 64) y0_lds[i] = y0[i];  // y0 is zero-padded
  if (NKX > 0)
    for (int i = threadIdx.x; i < NKK * 32; i += kMfmaWaves * 64)
      bias_lds[i] = (mp.bias != nullptr && i < n) ? mp.bias[i] : 0.f;
  __syncthreads();  // the only workgroup barrier; from here on the waves are independent
  float (*patch)[LSTR] = line_lds[kLines ? wave : 0];

  // persistent walk over groups of NT*32 samples (no workgroup barriers anywhere)
  for (int64_t grp = wave_id; grp < n_groups; grp += wave_stride) {
  const int64_t s_base = grp * (NT * 32);

  // ---- this lane's half of v for each of its samples, as B operands
  float vr[NT][KK];
  bool live[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) live[t] = (s_base + t * 32 + col) < B;
  if constexpr (NKX == 0) {
    load_rows<NT, NKK, LSTR, kLines>(vr, v, ldv, n, vec_in, s_base, B, live, patch, lane);
  } else {
    // mapper: rows of x as B operands, 32-row tiles of Wm as A operands read in place (row-major
    // weight: a lane's four consecutive K-steps are one 16-byte piece of its row), accumulators
    // started at the bias; the result registers are the walk's B operands.
    constexpr int NQX = NKX * 4;
    f32x4 ma[NQX], mb[NQX];
    auto fetch_map = [&](f32x4 (&buf)[NQX], const int tp) {
      const int row = 32 * tp + col;
#pragma unroll
      for (int q = 0; q < NQX; ++q) {
        const int c0 = 8 * q + 4 * hi;
        buf[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (row < n && c0 < mp.in_dim) buf[q] = *reinterpret_cast<const f32x4*>(mp.w + row * mp.ldw + c0);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    fetch_map(ma, 0);
    float xr[NT][NKX * 16];
    load_rows<NT, NKX, LSTR, kLines>(xr, v, ldv, mp.in_dim, vec_in, s_base, B, live, patch, lane);
    auto map_tile = [&](const f32x4 (&a)[NQX], const int tp) {
      f32x16 macc[NT];
#pragma unroll
      for (int a4 = 0; a4 < 4; ++a4) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(&bias_lds[32 * tp + 8 * a4 + 4 * hi]);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int c = 0; c < 4; ++c) macc[t][4 * a4 + c] = b4[c];
      }
#pragma unroll
      for (int q = 0; q < NQX; ++q)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int t = 0; t < NT; ++t)
            macc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][c], xr[t][4 * q + c], macc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 16; ++g) vr[t][16 * tp + g] = macc[t][g];
    };
#pragma unroll
    for (int tp = 0; tp < NKK; tp += 2) {
      if (tp + 1 < NKK) fetch_map(mb, tp + 1);
      map_tile(ma, tp);
      if (tp + 1 < NKK) {
        if (tp + 2 < NKK) fetch_map(ma, tp + 2);
        map_tile(mb, tp + 1);
      }
    }
    if (mp.v_out != nullptr) {
      float one[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) one[t] = 1.f;
      (void)store_rows<NT, NKK, LSTR, kLines>(vr, one, nullptr, mp.v_out, mp.ldvo, n,
                                             (mp.ldvo % 4 == 0) && ((reinterpret_cast<uintptr_t>(mp.v_out) & 15) == 0),
                                             s_base, B, live, patch, lane);
    }
  }

  // RAYEN_old head (rayen/constraint_module.py:460-466): y = y0 + N v / (||v|| e^beta + kappa(v)),
  // beta = column n of the input
  float old_den[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) old_den[t] = 0.f;
  if (NKX == 0 && old_mode) {
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

  // one 32-row tile: NQ k-groups of 4 MFMA steps on every sample tile.  FB = first 32-column block
  // the tile needs (the blocks before it were folded into their transposes).  The first MFMA of a
  // chain takes the constant 0 as its C operand: no accumulator initialisation instructions (VALU
  // work and MFMA issue of a SIMD are serial).
  auto run_from = [&](f32x16 (&acc)[NT], const f32x4 (&a)[NQ], auto fb_tag) {
    constexpr int FB = decltype(fb_tag)::value;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 4 * FB; q < NQ; ++q)
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int t = 0; t < NT; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][c], vr[t][4 * q + c],
                                                        (q == 4 * FB && c == 0) ? zero : acc[t], 0, 0, 0);
  };
  auto run_tile = [&](f32x16 (&acc)[NT], const f32x4 (&a)[NQ], const int qbegin) {
    if (qbegin == 0) {
      run_from(acc, a, std::integral_constant<int, 0>{});
    } else if (NKK > 1 && qbegin == 4) {
      run_from(acc, a, std::integral_constant<int, (NKK > 1 ? 1 : 0)>{});
    } else if (NKK > 2 && qbegin == 8) {
      run_from(acc, a, std::integral_constant<int, (NKK > 2 ? 2 : 0)>{});
    } else {
      run_from(acc, a, std::integral_constant<int, (NKK > 3 ? 3 : 0)>{});
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
    } else if (STAGED && item.type == MI_OUT) {
      // rows of NA_E: the 32 x 32 result block goes through this wave's aux patch (dead by now: every
      // segment has closed), XOR-swizzled so that both the fragment-shaped writes and the row-shaped
      // reads are conflict-free, and leaves as row-coalesced stores (two samples x 32 consecutive
      // outputs per instruction instead of 64 scattered words)
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
    } else if (!STAGED && item.type == MI_OUT) {
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
          // (v_sqrt_f32, 1 ulp: the IEEE-exact sqrtf costs ~10 more VALU instructions per constraint, and
          // VALU work is serial with the MFMA stream; 72 small constraints per group in config 5)
          // (kept to the general-shape instance so that the NA_E = I instances stay the code they were tuned as)
          const float kc = aux_lds[wave][t][slot & 31][col] + (STAGED ? __builtin_amdgcn_sqrtf(qs) : sqrtf(qs));
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
    bad |= store_rows<NT, NKK, LSTR, kLines>(vr, scale, y0_lds, y, ldy, k, vec_out, s_base, B, live, patch, lane);
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

// persistent launch: at most `slots` waves are resident (VGPR-limited waves per SIMD x SIMDs);
// every wave gets the same number of sample groups so that no SIMD idles in a ragged last round
template <int NKK, int NKX>
static int launch_mfma(const RayenPack* p, const MfmaImage* img, const float* v, int64_t B, int64_t ldv,
                       float* y, int64_t ldy, float* kappa, int32_t* active, int32_t* nan_flag,
                       int old_mode, const MapperArgs& mp, hipStream_t stream) {
  constexpr int per_wave = MfmaCfg<NKK>::NT * 32;
  const int64_t n_groups = (B + per_wave - 1) / per_wave;
  const int64_t slots = (int64_t)launch_simds(img->n_simd) * img->waves_per_simd;
  const int64_t rounds = (n_groups + slots - 1) / slots;
  const int64_t waves = (n_groups + rounds - 1) / rounds;
  const int64_t grid = (waves + kMfmaWaves - 1) / kMfmaWaves;
  const int vec_in = (ldv % 4 == 0) && ((reinterpret_cast<uintptr_t>(v) & 15) == 0);
  const int vec_out = (ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0);
  auto go = [&](auto kern) {
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kMfmaWaves * 64), 0, stream, img->W, img->items,
                       img->n_items, img->packs, img->y0, img->identity, p->k, p->n, v, B, ldv, vec_in, y, ldy,
                       vec_out, kappa, active, nan_flag, old_mode, mp);
  };
  if (img->identity) {
    if (active != nullptr) go(mfma_fwd_kernel<NKK, NKX, true, false>);
    else go(mfma_fwd_kernel<NKK, NKX, false, false>);
  } else {
    if (active != nullptr) go(mfma_fwd_kernel<NKK, NKX, true, true>);
    else go(mfma_fwd_kernel<NKK, NKX, false, true>);
  }
  return hipGetLastError() == hipSuccess ? RAYEN_OK : RAYEN_E_LAUNCH;
}

}  // namespace rayen
