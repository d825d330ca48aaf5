// fp64 matrix-core backward (the reference trains in fp64, examples/main.py:288): the scheme of
// rayen_mfma_bwd.hip on v_mfma_f64_16x16x4_f64, with the lane layout of rayen_mfma_f64.hip --
// a wave owns 32 samples (two 16-sample column blocks); lane l holds sample l&15 and, with q = l>>4,
// element 4s + q of v as the B operand of K-step s, which is also the row it holds in result register
// g = s&3 of row half (s>>2)&1 of tile s>>3.  So S_s v lands in registers that line up with v, and the
// gradient of kappa is selected per lane without any data movement.  grad_y is read twice (once for
// g.v, once for the final combination) and grad kappa is parked in LDS ([element][lane], written only by
// the lanes whose segment a tile belongs to), so that the register budget is the forward kernel's.
#include "rayen_bwd_tiles.h"
#include "rayen_bwd_bucket.h"
#include "rayen_internal.h"

namespace rayen {

using f64x4 = double __attribute__((ext_vector_type(4)));
using f64x2 = double __attribute__((ext_vector_type(2)));

struct Mfma64BwdImage {
  f64x2* S = nullptr;      // [tile][step pair][row half][lane] x 2 doubles (rayen_mfma_f64.hip order)
  BItem* items = nullptr;
  double* Wrow = nullptr;  // [n_rows + 2][n_pad] row-major copy of W
  int32_t* seg_bucket = nullptr;  // [n_segments]: bucket of the bucketed walk (rayen_bwd_bucket.h)
  int n_items = 0;
  int nkk = 0;
  int n_dense = 0;
  int n_simd = 1024;
  int64_t bytes = 0;
};

constexpr int kB64Waves = 8;

__device__ __forceinline__ double bq16(double x) { return __shfl_xor(x, 16); }
__device__ __forceinline__ double bq32(double x) { return __shfl_xor(x, 32); }
__device__ __forceinline__ double quad_sum(double x) {
  x += bq16(x);
  x += bq32(x);
  return x;
}

// BUCKET: the samples come through the permutation of rayen_bwd_bucket.h; a group of 32 belongs to one bucket and
// walks only that bucket's tiles (buckets start on multiples of 64).
template <int NKK, bool BUCKET>
__global__ __launch_bounds__(kB64Waves * 64, 2) void mfma64_bwd_kernel(
    const f64x2* __restrict__ Simg, const BItem* __restrict__ items, int n_items,
    const double* __restrict__ Wrow, int n, const double* __restrict__ v, int64_t B, int64_t ldv,
    const double* __restrict__ kappa, const int32_t* __restrict__ active, const double* __restrict__ gy,
    int64_t ldg, double* __restrict__ gv, int64_t ldgv, int old_mode, const int32_t* __restrict__ ws, int nb) {
  constexpr int NS = NKK * 8, NP = NKK * 32;
  __shared__ double u_lds[kB64Waves][2][NS][64];  // grad kappa: [wave][column block][element 4 st + q][lane]
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int j = lane & 15;
  const int q = lane >> 4;
  const int64_t n_groups = BUCKET ? (int64_t)(ws[kWsOffsets + nb] / 32) : (B + 31) / 32;
  const int64_t wave_id = (int64_t)blockIdx.x * kB64Waves + wave;
  const int64_t wave_stride = (int64_t)gridDim.x * kB64Waves;

  for (int64_t grp = wave_id; grp < n_groups; grp += wave_stride) {
    const int64_t s_base = grp * 32;
    double vb[2][NS];
    bool live[2], clipped[2], matched[2];
    double kap[2], tv[2], sc[2], r_nrm[2], e_beta[2];
    int aseg[2], arow[2];
    int64_t smp_of[2];
    int bucket = -1;
    if constexpr (BUCKET) {
      for (int i = 0; i < nb; ++i)
        if (s_base >= ws[kWsOffsets + i]) bucket = i;
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      int64_t s = s_base + 16 * c + j;
      if constexpr (BUCKET) {
        s = ws[kWsHeader + s];
        live[c] = s >= 0;
      } else {
        live[c] = s < B;
      }
      smp_of[c] = s;
      const double* row = v + (live[c] ? s : 0) * ldv;
      const double* grow = gy + (live[c] ? s : 0) * ldg;
      double dot = 0.0, nrm2 = 0.0;
#pragma unroll
      for (int st = 0; st < NS; ++st) {
        const bool in = live[c] && 4 * st + q < n;
        vb[c][st] = in ? row[4 * st + q] : 0.0;
        const double g = in ? grow[4 * st + q] : 0.0;
        dot = fma(g, vb[c][st], dot);
        nrm2 = fma(vb[c][st], vb[c][st], nrm2);
      }
      tv[c] = quad_sum(dot);
      kap[c] = live[c] ? kappa[s] : 0.0;
      aseg[c] = live[c] ? active[2 * s] : -1;
      arow[c] = live[c] ? active[2 * s + 1] : 0;
      matched[c] = false;
      r_nrm[c] = 0.0;
      e_beta[c] = 0.0;
      if (old_mode) {
        r_nrm[c] = sqrt(quad_sum(nrm2));
        e_beta[c] = live[c] ? exp(row[n]) : 0.0;
        clipped[c] = live[c] && aseg[c] >= 0 && r_nrm[c] > 0.0;
        sc[c] = r_nrm[c] > 0.0 ? 1.0 / (r_nrm[c] * e_beta[c] + kap[c]) : 0.0;
      } else {
        clipped[c] = live[c] && kap[c] > 1.0 && aseg[c] >= 0;
        sc[c] = 1.0 / fmax(1.0, kap[c]);
      }
    }

    const int it_lo = BUCKET ? (bucket >= 2 ? (bucket - 2) * NKK : 0) : 0;
    const int it_hi = BUCKET ? (bucket >= 2 ? (bucket - 1) * NKK : 0) : n_items;
    if (__ballot(clipped[0] || clipped[1]) != 0 && it_hi > it_lo) {  // a wave of interior samples skips the walk
      const f64x2* wp = Simg + lane + (size_t)it_lo * (NS * 64);
      f64x2 buf_lo[NS / 4][2], buf_hi[NS / 4][2];  // [step pair within the half][row half]
      auto fetch_half = [&](f64x2 (&buf)[NS / 4][2]) {
#pragma unroll
        for (int p = 0; p < NS / 4; ++p)
#pragma unroll
          for (int rh = 0; rh < 2; ++rh) buf[p][rh] = wp[(p * 2 + rh) * 64];
        wp += (NS / 4) * 2 * 64;
        __builtin_amdgcn_sched_barrier(0);
      };
      fetch_half(buf_lo);
      fetch_half(buf_hi);
      double part[2] = {0.0, 0.0};
      f64x4 acc[2][2];  // [row half][column block]
      for (int it = it_lo; it < it_hi; ++it) {
        const BItem item = items[it];
        if (item.type == BI_NOP) {
          fetch_half(buf_lo);
          fetch_half(buf_hi);
          continue;
        }
        const f64x4 zero4 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int p = 0; p < NS / 4; ++p)
#pragma unroll
          for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int rh = 0; rh < 2; ++rh)
#pragma unroll
              for (int c = 0; c < 2; ++c)  // the first MFMA of a chain starts from the constant 0
                acc[rh][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(buf_lo[p][rh][e], vb[c][2 * p + e],
                                                                  (p == 0 && e == 0) ? zero4 : acc[rh][c], 0, 0, 0);
        fetch_half(buf_lo);
#pragma unroll
        for (int p = 0; p < NS / 4; ++p)
#pragma unroll
          for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int rh = 0; rh < 2; ++rh)
#pragma unroll
              for (int c = 0; c < 2; ++c)
                acc[rh][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(buf_hi[p][rh][e], vb[c][NS / 2 + 2 * p + e], acc[rh][c], 0, 0, 0);
        fetch_half(buf_hi);

        bool sel[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          sel[c] = clipped[c] && aseg[c] == item.seg;
          double sum = (item.flags & MF_FIRST) ? 0.0 : part[c];
#pragma unroll
          for (int tp = 0; tp < NKK; ++tp)
            if (item.tp == tp) {
#pragma unroll
              for (int rh = 0; rh < 2; ++rh)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                  const int st = 8 * tp + 4 * rh + g;
                  sum = fma(acc[rh][c][g], vb[c][st], sum);
                  if (sel[c]) u_lds[wave][c][st][lane] = acc[rh][c][g];
                }
            }
          part[c] = sum;
        }
        if ((item.flags & MF_LAST) && __ballot(sel[0] || sel[1]) != 0) {
          const double* ax = Wrow + (int64_t)item.aux_row * NP + q;
          double cw[2], c0[2], c1[2];
          if (item.type == BI_QUAD) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
              const double total = quad_sum(part[c]);  // v'S v
              cw[c] = total > 0.0 ? 1.0 / sqrt(total) : 0.0;
              c0[c] = 1.0;
              c1[c] = 0.0;
            }
          } else {
            double cr[2] = {0.0, 0.0}, br[2] = {0.0, 0.0};
#pragma unroll
            for (int st = 0; st < NS; ++st) {
              const double x0 = ax[4 * st], x1 = ax[NP + 4 * st];
#pragma unroll
              for (int c = 0; c < 2; ++c) {
                cr[c] = fma(x0, vb[c][st], cr[c]);
                br[c] = fma(x1, vb[c][st], br[c]);
              }
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
              const double crs = quad_sum(cr[c]), brs = quad_sum(br[c]);
              const double tau = item.f0d, ap = item.f1d;
              const double bp = 2.0 * brs - 2.0 * crs * tau;
              const double den = 2.0 * ap * kap[c] + bp;  // dF/dkappa at the root
              const double inv = den != 0.0 ? -1.0 / den : 0.0;
              cw[c] = 2.0 * inv;                             // d c'/dv = 2 M'Mv - 2 (c.v) c
              c0[c] = inv * (-2.0 * crs - 2.0 * tau * kap[c]);
              c1[c] = inv * 2.0 * kap[c];                    // kappa * d b'/dv = kappa (2 M'beta - 2 tau c)
            }
          }
#pragma unroll
          for (int st = 0; st < NS; ++st) {
            const double x0 = ax[4 * st];
            const double x1 = item.type == BI_SOC ? ax[NP + 4 * st] : 0.0;
#pragma unroll
            for (int c = 0; c < 2; ++c)
              if (sel[c]) u_lds[wave][c][st][lane] = fma(cw[c], u_lds[wave][c][st][lane], fma(c0[c], x0, c1[c] * x1));
          }
#pragma unroll
          for (int c = 0; c < 2; ++c) matched[c] |= sel[c];
        }
      }
    }
    // every quadratic / cone is in the item list: what is left is a linear row
#pragma unroll
    for (int c = 0; c < 2; ++c)
      if (clipped[c] && !matched[c]) {
        const double* row = Wrow + (int64_t)arow[c] * NP + q;
#pragma unroll
        for (int st = 0; st < NS; ++st) u_lds[wave][c][st][lane] = row[4 * st];
      }

#pragma unroll
    for (int c = 0; c < 2; ++c) {
      if (!live[c]) continue;
      const int64_t s = smp_of[c];
      const double* grow = gy + s * ldg;
      double* orow = gv + s * ldgv;
      if (!old_mode) {
        const double coef = clipped[c] ? sc[c] * sc[c] * tv[c] : 0.0;
#pragma unroll
        for (int st = 0; st < NS; ++st) {
          const int r = 4 * st + q;
          const double u = clipped[c] ? u_lds[wave][c][st][lane] : 0.0;  // a clipped lane has written every element
          if (r < n) orow[r] = fma(sc[c], grow[r], -coef * u);
        }
      } else {
        // grad_v = s t - s^2 (t.v) (e^beta v / r + grad kappa),  grad_beta = -s^2 (t.v) r e^beta
        const double coef = sc[c] * sc[c] * tv[c];
        const double dir = r_nrm[c] > 0.0 ? e_beta[c] / r_nrm[c] : 0.0;
#pragma unroll
        for (int st = 0; st < NS; ++st) {
          const int r = 4 * st + q;
          const double u = clipped[c] ? u_lds[wave][c][st][lane] : 0.0;
          if (r < n) orow[r] = fma(sc[c], grow[r], -coef * fma(dir, vb[c][st], u));
        }
        if (q == 0) orow[n] = -coef * r_nrm[c] * e_beta[c];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------

bool mfma64_bwd_eligible(const RayenPack* p) { return mfma64_eligible(p) && bwd_tiles_eligible(p); }

void mfma64_bwd_free(Mfma64BwdImage* img) {
  if (img == nullptr) return;
  if (img->S) (void)hipFree(img->S);
  if (img->items) (void)hipFree(img->items);
  if (img->Wrow) (void)hipFree(img->Wrow);
  if (img->seg_bucket) (void)hipFree(img->seg_bucket);
  delete img;
}

int mfma64_bwd_build(const RayenPack* p, Mfma64BwdImage** out, int64_t* bytes) {
  const int n = p->n, np = n_pad_of(n);
  TileLayout b(n);
  std::vector<BItem> items;
  const int n_real = layout_bwd_tiles(p, b, items);
  const int nt = b.n_tiles(), ns = np / 4;
  // [tile][step pair sg][row half rh][lane l][2]: S[16 rh + (l&15)][4 (2 sg + e) + (l>>4)], e = 0, 1
  std::vector<double> frag((size_t)nt * (ns / 2) * 2 * 64 * 2, 0.0);
  for (int t = 0; t < nt; ++t)
    for (int sg = 0; sg < ns / 2; ++sg)
      for (int rh = 0; rh < 2; ++rh)
        for (int l = 0; l < 64; ++l)
          for (int e = 0; e < 2; ++e)
            frag[((((size_t)t * (ns / 2) + sg) * 2 + rh) * 64 + l) * 2 + e] =
                b.raw[((size_t)t * 32 + 16 * rh + (l & 15)) * np + 4 * (2 * sg + e) + (l >> 4)];
  std::vector<double> wrow((size_t)(p->n_rows + 2) * np, 0.0);
  for (int r = 0; r < p->n_rows; ++r)
    for (int c = 0; c < n; ++c) wrow[(size_t)r * np + c] = p->W[(size_t)r * n + c];

  Mfma64BwdImage* img = new Mfma64BwdImage();
  img->nkk = np / 32;
  img->n_items = n_real;
  const std::vector<int32_t> seg_bucket = bucket_table(p, bwd_quad_like, &img->n_dense);
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, p->device) == hipSuccess && prop.multiProcessorCount > 0)
      img->n_simd = prop.multiProcessorCount * 4;
  }
  const bool ok =
      hipMalloc(&img->S, frag.size() * sizeof(double)) == hipSuccess &&
      hipMemcpy(img->S, frag.data(), frag.size() * sizeof(double), hipMemcpyHostToDevice) == hipSuccess &&
      hipMalloc(&img->items, items.size() * sizeof(BItem)) == hipSuccess &&
      hipMemcpy(img->items, items.data(), items.size() * sizeof(BItem), hipMemcpyHostToDevice) == hipSuccess &&
      hipMalloc(&img->Wrow, wrow.size() * sizeof(double)) == hipSuccess &&
      hipMemcpy(img->Wrow, wrow.data(), wrow.size() * sizeof(double), hipMemcpyHostToDevice) == hipSuccess &&
      hipMalloc(&img->seg_bucket, seg_bucket.size() * sizeof(int32_t)) == hipSuccess &&
      hipMemcpy(img->seg_bucket, seg_bucket.data(), seg_bucket.size() * sizeof(int32_t), hipMemcpyHostToDevice) == hipSuccess;
  if (!ok) { mfma64_bwd_free(img); return RAYEN_E_ALLOC; }
  img->bytes = (int64_t)(frag.size() * sizeof(double) + items.size() * sizeof(BItem) + wrow.size() * sizeof(double) +
                         seg_bucket.size() * sizeof(int32_t));
  *bytes = img->bytes;
  *out = img;
  return RAYEN_OK;
}

int64_t mfma64_bwd_workspace_bytes(const RayenPack* p, const Mfma64BwdImage* img, int64_t B) {
  (void)p;
  return img == nullptr ? 0 : bucket_workspace_bytes(img->n_dense, img->nkk, B);
}

template <int NKK>
static int launch_bwd64(const RayenPack* p, const Mfma64BwdImage* img, const double* v, int64_t B, int64_t ldv,
                        const double* kappa, const int32_t* active, const double* gy, int64_t ldg, double* gv,
                        int64_t ldgv, int old_mode, void* workspace, int64_t workspace_bytes, hipStream_t stream) {
  const int64_t slots = (int64_t)launch_simds(img->n_simd) * 2;
  const int64_t need = old_mode ? 0 : mfma64_bwd_workspace_bytes(p, img, B);
  const bool bucketed = need > 0 && workspace != nullptr && workspace_bytes >= need;
  const int nb = img->n_dense + 2;
  int32_t* ws = static_cast<int32_t*>(workspace);
  if (bucketed) launch_bucket_sort<double>(kappa, active, B, img->seg_bucket, nb, ws, stream);
  const int64_t n_groups = bucketed ? (B + 31) / 32 + 2 * nb : (B + 31) / 32;   // (bucketed: the kernel reads the true count)
  const int64_t rounds = (n_groups + slots - 1) / slots;
  const int64_t waves = (n_groups + rounds - 1) / rounds;
  const int64_t grid = (waves + kB64Waves - 1) / kB64Waves;
  if (bucketed)
    hipLaunchKernelGGL((mfma64_bwd_kernel<NKK, true>), dim3((unsigned)grid), dim3(kB64Waves * 64), 0, stream, img->S,
                       img->items, img->n_items, img->Wrow, p->n, v, B, ldv, kappa, active, gy, ldg, gv, ldgv, 0,
                       static_cast<const int32_t*>(ws), nb);
  else
    hipLaunchKernelGGL((mfma64_bwd_kernel<NKK, false>), dim3((unsigned)grid), dim3(kB64Waves * 64), 0, stream, img->S,
                       img->items, img->n_items, img->Wrow, p->n, v, B, ldv, kappa, active, gy, ldg, gv, ldgv,
                       old_mode, static_cast<const int32_t*>(nullptr), 0);
  return hipGetLastError() == hipSuccess ? RAYEN_OK : RAYEN_E_LAUNCH;
}

int mfma64_backward(const RayenPack* p, const Mfma64BwdImage* img, const double* v, int64_t B, int64_t ldv,
                    const double* kappa, const int32_t* active, const double* grad_y, int64_t ldg,
                    double* grad_v, int64_t ldgv, int old_mode, void* workspace, int64_t workspace_bytes,
                    hipStream_t stream) {
  if (B == 0) return RAYEN_OK;
  if (img->nkk == 1) return launch_bwd64<1>(p, img, v, B, ldv, kappa, active, grad_y, ldg, grad_v, ldgv, old_mode, workspace, workspace_bytes, stream);
  if (img->nkk == 2) return launch_bwd64<2>(p, img, v, B, ldv, kappa, active, grad_y, ldg, grad_v, ldgv, old_mode, workspace, workspace_bytes, stream);
  return RAYEN_E_UNSUPPORTED;
}

}  // namespace rayen
