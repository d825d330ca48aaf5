// fp32 MFMA backward of the projection (packs with NA_E = I, n <= 64, no LMI):
//
//   grad_v = s g - [kappa > 1] s^2 (g . v) grad kappa(v),        s = 1 / max(1, kappa)
//
// which is what autograd produces for rayen/constraint_module.py:351-474 (max -> arg-max, relu); the
// RAYEN_old head (:460-466) adds the terms of its step column.  grad kappa belongs to the ONE
// constraint that set kappa (`active`, recorded by the forward):
//   linear row i          D_i                                              (:353)
//   quadratic             phi + S v / sqrt(v'S v),   S = G  or  U'U        (:374)
//   second-order cone     implicit derivative of the root of a'x^2 + b'x + c' = 0 (:383-399, 339-348):
//                         -(2 S v + (-2 c.v - 2 tau kappa) c + 2 kappa M'beta) / (2 a' kappa + b'),  S = M'M
// The generic backward lets every lane walk its own constraint (per-lane gathers of an n x n matrix).
// Here S_s v is evaluated for EVERY quadratic / cone s on the matrix cores -- one dense 32-row tile walk
// like the forward's, the same v-in-registers B operands -- and each lane keeps the rows of its own
// segment with a select; the linear case is a gather of one row.  A wave owns 64 samples (v and the
// gradient of kappa live in registers, n/2 VGPRs per 32 samples each).
#include "rayen_mfma_kernel.h"
#include "rayen_bwd_tiles.h"
#include "rayen_bwd_bucket.h"

#include <cstring>
#include <vector>

namespace rayen {

#ifndef RAYEN_BWD_NT
#define RAYEN_BWD_NT 2
#endif

struct MfmaBwdImage {
  f32x4* S = nullptr;      // [n_items + 1][NQ][64] float4, fragment order, one dense n_pad x n_pad form per segment
  BItem* items = nullptr;
  float* Wrow = nullptr;   // [n_rows][n_pad] row-major copy of W (gathers: linear rows, phi, c, M'beta)
  int32_t* seg_bucket = nullptr;  // [n_segments]: 1 = linear rows, 2 + d = the d-th dense form (bucketed walk)
  int n_items = 0;
  int nkk = 0;
  int n_dense = 0;         // dense forms = segments with tiles
  int n_segs = 0;
  int n_simd = 1024;
  int64_t bytes = 0;
};

// BUCKET: the samples come through the permutation of the bucketed walk (`ws`): a group of 64 belongs to ONE bucket
// and walks only that bucket's tiles.
template <int NKK, bool BUCKET>
__global__ __launch_bounds__(kMfmaWaves * 64, kMfmaWaves / 4) void mfma_bwd_kernel(
    const f32x4* __restrict__ Simg, const BItem* __restrict__ items, int n_items,
    const float* __restrict__ Wrow, int n, const float* __restrict__ v, int64_t B, int64_t ldv, int vec_v,
    const float* __restrict__ kappa, const int32_t* __restrict__ active, const float* __restrict__ gy,
    int64_t ldg, int vec_g, float* __restrict__ gv, int64_t ldgv, int vec_o, int old_mode,
    const int32_t* __restrict__ ws, int nb) {
  // two sample tiles per wave (every A fetch feeds two MFMA chains); grad_y is read twice -- once for
  // g.v, once for the final combination -- instead of occupying n/2 VGPRs per tile during the walk
  constexpr int NT = RAYEN_BWD_NT, NQ = NKK * 4, KK = NKK * 16, NP = NKK * 32, LSTR = NKK * 32 + 4;
  // Per-wave LDS region, used in turn by the transposition patch of load_rows / store_rows and, through the walk,
  // by grad_y parked in lane-private 16-byte slots (it is needed again for the final combination; a second trip
  // to global memory would be one more exposed round trip at the group boundary).
  constexpr int WL = (32 * LSTR > NT * KK * 64) ? 32 * LSTR : NT * KK * 64;
  __shared__ __attribute__((aligned(16))) float wave_lds[kMfmaWaves][WL];

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int col = lane & 31;
  const int hi = lane >> 5;
  static_assert(!BUCKET || NT == 2, "a bucketed group is one wave of 64 samples");
  const int64_t n_groups = BUCKET ? (int64_t)(ws[kWsOffsets + nb] / 64) : (B + NT * 32 - 1) / (NT * 32);
  const int64_t wave_id = (int64_t)blockIdx.x * kMfmaWaves + wave;
  const int64_t wave_stride = (int64_t)gridDim.x * kMfmaWaves;
  float (*patch)[LSTR] = reinterpret_cast<float (*)[LSTR]>(wave_lds[wave]);
  f32x4* gpark = reinterpret_cast<f32x4*>(wave_lds[wave]) + lane;  // piece j of sample tile t: gpark[(t * (KK / 4) + j) * 64]

  for (int64_t grp = wave_id; grp < n_groups; grp += wave_stride) {
    const int64_t s_base = grp * (NT * 32);
    bool live[NT], clipped[NT], matched[NT];
    float vr[NT][KK], ur[NT][KK];
    float kap[NT], tv[NT], sc[NT], r_nrm[NT], e_beta[NT], part[NT];
    int aseg[NT], arow[NT];
    int rowix = -1;        // BUCKET: this lane's entry of the group's 64 sample numbers
    int bucket = -1;       // BUCKET: the group's bucket (wave-uniform)
    int64_t smp_of[NT];    // the sample a lane's column holds, per tile
    if constexpr (BUCKET) {
      rowix = ws[kWsHeader + s_base + lane];
      for (int i = 0; i < nb; ++i)
        if (s_base >= ws[kWsOffsets + i]) bucket = i;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        smp_of[t] = __shfl(rowix, t * 32 + col);
        live[t] = smp_of[t] >= 0;
      }
      load_rows_ix<NT, NKK, LSTR>(vr, v, ldv, n, vec_v, rowix, patch, lane);
    } else {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        smp_of[t] = s_base + t * 32 + col;
        live[t] = smp_of[t] < B;
      }
      load_rows<NT, NKK, LSTR, true>(vr, v, ldv, n, vec_v, s_base, B, live, patch, lane);
    }
    {
      float tr[NT][KK];
      if constexpr (BUCKET) load_rows_ix<NT, NKK, LSTR>(tr, gy, ldg, n, vec_g, rowix, patch, lane);
      else load_rows<NT, NKK, LSTR, true>(tr, gy, ldg, n, vec_g, s_base, B, live, patch, lane);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < KK; ++i) dot = fmaf(tr[t][i], vr[t][i], dot);
        tv[t] = dot + xhalf(dot);
      }
      __builtin_amdgcn_wave_barrier();  // every lane is done reading the patch
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < KK / 4; ++j)
          gpark[(t * (KK / 4) + j) * 64] = f32x4{tr[t][4 * j], tr[t][4 * j + 1], tr[t][4 * j + 2], tr[t][4 * j + 3]};
    }
    bool any = false;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int64_t smp = live[t] ? smp_of[t] : 0;
      kap[t] = live[t] ? kappa[smp] : 0.f;
      aseg[t] = live[t] ? active[2 * smp] : -1;
      arow[t] = live[t] ? active[2 * smp + 1] : 0;
      matched[t] = false;
      part[t] = 0.f;
      // RAYEN: s = 1/max(1,kappa), kappa matters once it clips.  RAYEN_old: s = 1/(r e^beta + kappa),
      // r = ||v||: kappa always matters, and r, beta get gradients too.
      r_nrm[t] = 0.f;
      e_beta[t] = 0.f;
      if (old_mode) {
        float nrm2 = 0.f;
#pragma unroll
        for (int i = 0; i < KK; ++i) nrm2 = fmaf(vr[t][i], vr[t][i], nrm2);
        nrm2 += xhalf(nrm2);
        r_nrm[t] = sqrtf(nrm2);
        e_beta[t] = live[t] ? __expf(v[smp * ldv + n]) : 0.f;   // (old head: never bucketed)
        clipped[t] = live[t] && aseg[t] >= 0 && r_nrm[t] > 0.f;
        sc[t] = r_nrm[t] > 0.f ? 1.f / (r_nrm[t] * e_beta[t] + kap[t]) : 0.f;
      } else {
        clipped[t] = live[t] && kap[t] > 1.f && aseg[t] >= 0;
        sc[t] = 1.f / fmaxf(1.f, kap[t]);
      }
      any |= clipped[t];
#pragma unroll
      for (int i = 0; i < KK; ++i) ur[t][i] = 0.f;
    }

    // bucketed: the tiles of this group's dense form only (bucket 2 + d = items [d NKK, (d + 1) NKK)); buckets 0 / 1 walk nothing
    const int it_lo = BUCKET ? (bucket >= 2 ? (bucket - 2) * NKK : 0) : 0;
    const int it_hi = BUCKET ? (bucket >= 2 ? (bucket - 1) * NKK : 0) : n_items;
    if (__ballot(any) != 0) {  // wave-uniform: a wave of interior samples skips the walk
      const f32x4* wp = Simg + lane + (size_t)it_lo * (NQ * 64);
      f32x4 buf_a[NQ], buf_b[NQ];
      auto fetch_tile = [&](f32x4 (&buf)[NQ]) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) buf[q] = wp[q * 64];
        wp += NQ * 64;
        __builtin_amdgcn_sched_barrier(0);
      };
      auto process = [&](const BItem item, const f32x4 (&a)[NQ]) {
        if (item.type == BI_NOP) return;
        f32x16 acc[NT];
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int t = 0; t < NT; ++t)  // the first MFMA of a chain starts from the constant 0
              acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][c], vr[t][4 * q + c],
                                                            (q == 0 && c == 0) ? zero : acc[t], 0, 0, 0);
        bool sel[NT], any_sel = false;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          sel[t] = clipped[t] && aseg[t] == item.seg;
          any_sel |= sel[t];
          float sum = (item.flags & MF_FIRST) ? 0.f : part[t];
#pragma unroll
          for (int tp = 0; tp < NKK; ++tp)
            if (item.tp == tp) {
#pragma unroll
              for (int g = 0; g < 16; ++g) {
                sum = fmaf(acc[t][g], vr[t][16 * tp + g], sum);
                ur[t][16 * tp + g] = sel[t] ? acc[t][g] : ur[t][16 * tp + g];
              }
            }
          part[t] = sum;
        }
        if ((item.flags & MF_LAST) && __ballot(any_sel) != 0) {
          const float* ax = Wrow + (int64_t)item.aux_row * NP + 4 * hi;
          float cw[NT], c0[NT], c1[NT];
          if (item.type == BI_QUAD) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
              const float total = part[t] + xhalf(part[t]);  // v'S v
              cw[t] = total > 0.f ? 1.f / sqrtf(total) : 0.f;
              c0[t] = 1.f;
              c1[t] = 0.f;
            }
          } else {
            float cr[NT], br[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) { cr[t] = 0.f; br[t] = 0.f; }
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
              const f32x4 x0 = *reinterpret_cast<const f32x4*>(ax + 8 * q);
              const f32x4 x1 = *reinterpret_cast<const f32x4*>(ax + NP + 8 * q);
#pragma unroll
              for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                  cr[t] = fmaf(x0[c], vr[t][4 * q + c], cr[t]);
                  br[t] = fmaf(x1[c], vr[t][4 * q + c], br[t]);
                }
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
              const float crs = cr[t] + xhalf(cr[t]), brs = br[t] + xhalf(br[t]);
              const float tau = item.f0, ap = item.f1;
              const float bp = 2.f * brs - 2.f * crs * tau;
              const float den = 2.f * ap * kap[t] + bp;  // dF/dkappa at the root
              const float inv = den != 0.f ? -1.f / den : 0.f;
              cw[t] = 2.f * inv;                            // d c'/dv = 2 M'Mv - 2 (c.v) c
              c0[t] = inv * (-2.f * crs - 2.f * tau * kap[t]);
              c1[t] = inv * 2.f * kap[t];                   // kappa * d b'/dv = kappa (2 M'beta - 2 tau c)
            }
          }
#pragma unroll
          for (int q = 0; q < NQ; ++q) {
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(ax + 8 * q);
            f32x4 x1 = {0.f, 0.f, 0.f, 0.f};
            if (item.type == BI_SOC) x1 = *reinterpret_cast<const f32x4*>(ax + NP + 8 * q);
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
              for (int t = 0; t < NT; ++t) {
                const float u = fmaf(cw[t], ur[t][4 * q + c], fmaf(c0[t], x0[c], c1[t] * x1[c]));
                ur[t][4 * q + c] = sel[t] ? u : ur[t][4 * q + c];
              }
          }
#pragma unroll
          for (int t = 0; t < NT; ++t) matched[t] |= sel[t];
        }
      };
      if (it_hi > it_lo) {
        fetch_tile(buf_a);
        for (int it = it_lo; it < it_hi; it += 2) {  // n_items is even (padded with a no-op tile); a bucket's NKK tiles
          fetch_tile(buf_b);                         // may be odd: the partner is then skipped (its fetch is a harmless
          process(items[it], buf_a);                 // look-ahead into the next form / the spare tile)
          fetch_tile(buf_a);
          if (it + 1 < it_hi) process(items[it + 1], buf_b);
        }
      }
      // every quadratic / cone is in the item list: what is left is a linear row
#pragma unroll
      for (int t = 0; t < NT; ++t)
        if (clipped[t] && !matched[t]) {
          const float* row = Wrow + (int64_t)arow[t] * NP + 4 * hi;
#pragma unroll
          for (int q = 0; q < NQ; ++q) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(row + 8 * q);
            ur[t][4 * q + 0] = x[0];
            ur[t][4 * q + 1] = x[1];
            ur[t][4 * q + 2] = x[2];
            ur[t][4 * q + 3] = x[3];
          }
        }
    }

    // grad_v = s g - coef grad kappa (in place in ur), with g back from its parking slots
    {
      float tr[NT][KK];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < KK / 4; ++j) {
          const f32x4 g4 = gpark[(t * (KK / 4) + j) * 64];
          tr[t][4 * j] = g4[0]; tr[t][4 * j + 1] = g4[1]; tr[t][4 * j + 2] = g4[2]; tr[t][4 * j + 3] = g4[3];
        }
      __builtin_amdgcn_wave_barrier();  // the region goes back to the patch (store_rows below)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (!old_mode) {
          const float coef = clipped[t] ? sc[t] * sc[t] * tv[t] : 0.f;
#pragma unroll
          for (int i = 0; i < KK; ++i) ur[t][i] = fmaf(sc[t], tr[t][i], -coef * ur[t][i]);
        } else {
          // grad_v = s t - s^2 (t.v) (e^beta v / r + grad kappa),  grad_beta = -s^2 (t.v) r e^beta
          const float coef = sc[t] * sc[t] * tv[t];
          const float dir = r_nrm[t] > 0.f ? e_beta[t] / r_nrm[t] : 0.f;
#pragma unroll
          for (int i = 0; i < KK; ++i) ur[t][i] = fmaf(sc[t], tr[t][i], -coef * fmaf(dir, vr[t][i], ur[t][i]));
          if (live[t] && hi == 0) gv[smp_of[t] * ldgv + n] = -coef * r_nrm[t] * e_beta[t];
        }
      }
    }
    if constexpr (BUCKET) {
      store_rows_ix<NT, NKK, LSTR>(ur, gv, ldgv, n, vec_o, rowix, patch, lane);
    } else {
      float one[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) one[t] = 1.f;
      (void)store_rows<NT, NKK, LSTR, true>(ur, one, nullptr, gv, ldgv, n, vec_o, s_base, B, live, patch, lane);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------

bool mfma_bwd_eligible(const RayenPack* p) {
  return mfma_eligible(p) && bwd_tiles_eligible(p);
}

int mfma_bwd_build(const RayenPack* p, MfmaBwdImage** out, int64_t* bytes) {
  const int n = p->n, np = n_pad_of(n), nkk = np / 32;
  TileLayout b(n);
  std::vector<BItem> items;
  const int n_real = layout_bwd_tiles(p, b, items);
  const std::vector<float> frag = b.fragments_f32();
  std::vector<float> wrow((size_t)(p->n_rows + 2) * np, 0.f);  // (+2: an aux pair may be read past a last row)
  for (int r = 0; r < p->n_rows; ++r)
    for (int j = 0; j < n; ++j) wrow[(size_t)r * np + j] = (float)p->W[(size_t)r * n + j];

  MfmaBwdImage* img = new MfmaBwdImage();
  img->nkk = nkk;
  img->n_items = n_real;
  const std::vector<int32_t> seg_bucket = bucket_table(p, bwd_quad_like, &img->n_dense);
  img->n_segs = (int)p->segs.size();
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, p->device) == hipSuccess && prop.multiProcessorCount > 0)
      img->n_simd = prop.multiProcessorCount * 4;
  }
  const bool ok =
      hipMalloc(&img->S, frag.size() * sizeof(float)) == hipSuccess &&
      hipMemcpy(img->S, frag.data(), frag.size() * sizeof(float), hipMemcpyHostToDevice) == hipSuccess &&
      hipMalloc(&img->items, items.size() * sizeof(BItem)) == hipSuccess &&
      hipMemcpy(img->items, items.data(), items.size() * sizeof(BItem), hipMemcpyHostToDevice) == hipSuccess &&
      hipMalloc(&img->Wrow, wrow.size() * sizeof(float)) == hipSuccess &&
      hipMemcpy(img->Wrow, wrow.data(), wrow.size() * sizeof(float), hipMemcpyHostToDevice) == hipSuccess &&
      hipMalloc(&img->seg_bucket, seg_bucket.size() * sizeof(int32_t)) == hipSuccess &&
      hipMemcpy(img->seg_bucket, seg_bucket.data(), seg_bucket.size() * sizeof(int32_t), hipMemcpyHostToDevice) == hipSuccess;
  if (!ok) { mfma_bwd_free(img); return RAYEN_E_ALLOC; }
  img->bytes = (int64_t)(frag.size() * sizeof(float) + items.size() * sizeof(BItem) + wrow.size() * sizeof(float) +
                         seg_bucket.size() * sizeof(int32_t));
  *bytes = img->bytes;
  *out = img;
  return RAYEN_OK;
}

void mfma_bwd_free(MfmaBwdImage* img) {
  if (img == nullptr) return;
  if (img->S) (void)hipFree(img->S);
  if (img->items) (void)hipFree(img->items);
  if (img->Wrow) (void)hipFree(img->Wrow);
  if (img->seg_bucket) (void)hipFree(img->seg_bucket);
  delete img;
}

// The bucketed walk pays three small launches (~10 us): worth it once the dense forms it skips cost more, i.e. from
// two forms up and for batches that fill the chip.
int64_t mfma_bwd_workspace_bytes(const RayenPack* p, const MfmaBwdImage* img, int64_t B) {
  (void)p;
  return img == nullptr ? 0 : bucket_workspace_bytes(img->n_dense, img->nkk, B);
}

template <int NKK>
static int launch_bwd(const RayenPack* p, const MfmaBwdImage* img, const float* v, int64_t B, int64_t ldv,
                      const float* kappa, const int32_t* active, const float* gy, int64_t ldg, float* gv,
                      int64_t ldgv, int old_mode, void* workspace, int64_t workspace_bytes, hipStream_t stream) {
  auto aligned = [](const void* ptr, int64_t ld) { return (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(ptr) & 15) == 0); };
  const int64_t slots = (int64_t)launch_simds(img->n_simd) * kMfmaWavesPerSimd;
  const int64_t need = old_mode ? 0 : mfma_bwd_workspace_bytes(p, img, B);
  if (need > 0 && workspace != nullptr && workspace_bytes >= need) {
    const int nb = img->n_dense + 2;
    int32_t* ws = static_cast<int32_t*>(workspace);
    launch_bucket_sort<float>(kappa, active, B, img->seg_bucket, nb, ws, stream);
    const int64_t max_groups = (B + 63) / 64 + nb;   // (the kernel reads the true count from the workspace)
    const int64_t rounds = (max_groups + slots - 1) / slots;
    const int64_t waves = (max_groups + rounds - 1) / rounds;
    const int64_t grid = (waves + kMfmaWaves - 1) / kMfmaWaves;
    hipLaunchKernelGGL((mfma_bwd_kernel<NKK, true>), dim3((unsigned)grid), dim3(kMfmaWaves * 64), 0, stream, img->S,
                       img->items, img->n_items, img->Wrow, p->n, v, B, ldv, aligned(v, ldv) ? 1 : 0, kappa, active,
                       gy, ldg, aligned(gy, ldg) ? 1 : 0, gv, ldgv, aligned(gv, ldgv) ? 1 : 0, 0, ws, nb);
    return hipGetLastError() == hipSuccess ? RAYEN_OK : RAYEN_E_LAUNCH;
  }
  const int64_t n_groups = (B + RAYEN_BWD_NT * 32 - 1) / (RAYEN_BWD_NT * 32);
  const int64_t rounds = (n_groups + slots - 1) / slots;
  const int64_t waves = (n_groups + rounds - 1) / rounds;
  const int64_t grid = (waves + kMfmaWaves - 1) / kMfmaWaves;
  hipLaunchKernelGGL((mfma_bwd_kernel<NKK, false>), dim3((unsigned)grid), dim3(kMfmaWaves * 64), 0, stream, img->S,
                     img->items, img->n_items, img->Wrow, p->n, v, B, ldv, aligned(v, ldv) ? 1 : 0, kappa, active,
                     gy, ldg, aligned(gy, ldg) ? 1 : 0, gv, ldgv, aligned(gv, ldgv) ? 1 : 0, old_mode,
                     static_cast<const int32_t*>(nullptr), 0);
  return hipGetLastError() == hipSuccess ? RAYEN_OK : RAYEN_E_LAUNCH;
}

int mfma_backward(const RayenPack* p, const MfmaBwdImage* img, const float* v, int64_t B, int64_t ldv,
                  const float* kappa, const int32_t* active, const float* grad_y, int64_t ldg, float* grad_v,
                  int64_t ldgv, int old_mode, void* workspace, int64_t workspace_bytes, hipStream_t stream) {
  if (B == 0) return RAYEN_OK;
  switch (img->nkk) {
    case 1: return launch_bwd<1>(p, img, v, B, ldv, kappa, active, grad_y, ldg, grad_v, ldgv, old_mode, workspace, workspace_bytes, stream);
    case 2: return launch_bwd<2>(p, img, v, B, ldv, kappa, active, grad_y, ldg, grad_v, ldgv, old_mode, workspace, workspace_bytes, stream);
    default: return RAYEN_E_UNSUPPORTED;
  }
}

}  // namespace rayen
