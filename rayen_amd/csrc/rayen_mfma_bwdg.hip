// fp32 matrix-core backward, general shapes: sets with equality constraints (NA_E != I, so the incoming
// gradient is first pulled back, t = NA_E' g, on the matrix cores) and sets with many small low-rank
// quadratics (the packed tiles of the forward).  rayen_mfma_bwd.hip keeps the tuned kernel for NA_E = I with
// dense forms only; this one serves what that one declines (n, k <= 64, no LMI); with two 32-column blocks
// of v a wave takes one sample tile instead of two (and still spills: 1.6x the lane-per-sample backward).
//
//   grad_v = s t - [kappa > 1] s^2 (t . v) grad kappa(v),   t = NA_E' g,   s = 1 / max(1, kappa)
//
// Packed low-rank quadratics (kappa = phi.v + ||U v||, rank <= 8, eight half-quads of rows per tile as in
// the forward): a per-lane matrix gather is avoided with a MASKED two-step product --
//   step 1   w = U_tile v                    one tile walk step, all eight segments of the tile at once;
//            every lane zeroes the quads that do not belong to ITS active segment and scales the rest by
//            1 / ||U v||  (a 4-register sum, plus one half-wave exchange when the segment spans both halves);
//   step 2   u += U_tile' w                  the transposed tile as A operand, w -- which already sits in
//            the registers an MFMA B operand needs -- as B operand, accumulated over the pack tiles.
// Only the lane's own segment survives the mask, so u = U_s'(U_s v)/||U_s v||; phi_s is a row gather.
#include "rayen_bwd_tiles.h"
#include "rayen_mfma_kernel.h"
#include "rayen_bwd_bucket.h"

#include <cstring>
#include <vector>

namespace rayen {

struct MfmaBwdgImage {
  f32x4* S = nullptr;        // item tiles, fragment order, NQ float4 per lane and tile
  f32x4* NT = nullptr;       // NA_E' as [NKK tiles][NQG][64] float4 (K = k_pad), null when NA_E = I
  BItem* items = nullptr;
  BPack* packs = nullptr;
  int32_t* seg_aux = nullptr;  // [n_segments] W row of phi for factor segments, -1 otherwise
  float* Wrow = nullptr;     // [n_rows + 2][n_pad]
  int32_t* seg_bucket = nullptr;    // [n_segments]: bucket of the bucketed walk (1 linear rows, 2 + g item group g)
  int32_t* group_items = nullptr;   // [n_groups][2]: item range of every group (a dense form / one packed tile pair)
  int n_items = 0, nkk = 0, nkg = 0, n_simd = 1024;
  int n_groups = 0, n_group_tiles = 0;
  int64_t bytes = 0;
};

// BUCKET: the samples come through the permutation of rayen_bwd_bucket.h (grouped by the item group that serves their
// active constraint): a group of NT x 32 walks only that item group's tiles (config 5: one packed tile pair of nine).
template <int NKK, int NKG, bool BUCKET>
__global__ __launch_bounds__(kMfmaWaves * 64, kMfmaWaves / 4) void mfma_bwdg_kernel(
    const f32x4* __restrict__ Simg, const f32x4* __restrict__ NTimg, const BItem* __restrict__ items, int n_items,
    const BPack* __restrict__ packs, const int32_t* __restrict__ seg_aux, const float* __restrict__ Wrow, int n,
    int k, const float* __restrict__ v, int64_t B, int64_t ldv, int vec_v, const float* __restrict__ kappa,
    const int32_t* __restrict__ active, const float* __restrict__ gy, int64_t ldg, int vec_g,
    float* __restrict__ gv, int64_t ldgv, int vec_o, int old_mode, const int32_t* __restrict__ ws, int nb,
    const int32_t* __restrict__ group_items) {
  constexpr int NT = NKK == 1 ? 2 : 1;  // two 32-column blocks of v leave registers for one sample tile only
  constexpr int NQ = NKK * 4, KK = NKK * 16, NP = NKK * 32;
  constexpr int NKL = NKG > NKK ? NKG : NKK, LSTR = NKL * 32 + 4;
  constexpr int NQG = NKG * 4, KG = NKG > 0 ? NKG * 16 : 1;
  __shared__ __attribute__((aligned(16))) float line_lds[kMfmaWaves][32][LSTR];

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int col = lane & 31;
  const int hi = lane >> 5;
  const int64_t n_groups = BUCKET ? (int64_t)(ws[kWsOffsets + nb] / (NT * 32)) : (B + NT * 32 - 1) / (NT * 32);
  const int64_t wave_id = (int64_t)blockIdx.x * kMfmaWaves + wave;
  const int64_t wave_stride = (int64_t)gridDim.x * kMfmaWaves;
  float (*patch)[LSTR] = line_lds[wave];
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  for (int64_t grp = wave_id; grp < n_groups; grp += wave_stride) {
    const int64_t s_base = grp * (NT * 32);
    bool live[NT], clipped[NT], matched[NT], pmatched[NT];
    float vr[NT][KK];
    f32x16 u16[NT][NKK];
    float kap[NT], tv[NT], sc[NT], r_nrm[NT], e_beta[NT], part[NT];
    int aseg[NT], arow[NT];
    int rowix = -1, bucket = -1;
    int64_t smp_of[NT];
    if constexpr (BUCKET) {
      rowix = lane < NT * 32 ? ws[kWsHeader + s_base + lane] : -1;
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

    // t = NA_E' g (or g itself), in the register layout of v
    auto pull_back = [&](float (&tr)[NT][KK]) {
      if constexpr (NKG == 0) {
        if constexpr (BUCKET) load_rows_ix<NT, NKK, LSTR>(tr, gy, ldg, n, vec_g, rowix, patch, lane);
        else load_rows<NT, NKK, LSTR, true>(tr, gy, ldg, n, vec_g, s_base, B, live, patch, lane);
      } else {
        float gr[NT][KG];
        if constexpr (BUCKET) load_rows_ix<NT, NKG, LSTR>(gr, gy, ldg, k, vec_g, rowix, patch, lane);
        else load_rows<NT, NKG, LSTR, true>(gr, gy, ldg, k, vec_g, s_base, B, live, patch, lane);
#pragma unroll
        for (int tp = 0; tp < NKK; ++tp) {
          f32x4 a[NQG];
#pragma unroll
          for (int q = 0; q < NQG; ++q) a[q] = NTimg[((size_t)tp * NQG + q) * 64 + lane];
          f32x16 acc[NT];
#pragma unroll
          for (int q = 0; q < NQG; ++q)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
              for (int t = 0; t < NT; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][c], gr[t][4 * q + c],
                                                              (q == 0 && c == 0) ? zero : acc[t], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int g = 0; g < 16; ++g) tr[t][16 * tp + g] = acc[t][g];
        }
      }
    };
    {
      float tr[NT][KK];
      pull_back(tr);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < KK; ++i) dot = fmaf(tr[t][i], vr[t][i], dot);
        tv[t] = dot + xhalf(dot);
      }
    }
    bool any = false;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int64_t smp = live[t] ? smp_of[t] : 0;
      kap[t] = live[t] ? kappa[smp] : 0.f;
      aseg[t] = live[t] ? active[2 * smp] : -1;
      arow[t] = live[t] ? active[2 * smp + 1] : 0;
      matched[t] = false;
      pmatched[t] = false;
      part[t] = 0.f;
      r_nrm[t] = 0.f;
      e_beta[t] = 0.f;
      if (old_mode) {  // RAYEN_old: s = 1/(r e^beta + kappa), r = ||v||
        float nrm2 = 0.f;
#pragma unroll
        for (int i = 0; i < KK; ++i) nrm2 = fmaf(vr[t][i], vr[t][i], nrm2);
        nrm2 += xhalf(nrm2);
        r_nrm[t] = sqrtf(nrm2);
        e_beta[t] = live[t] ? __expf(v[smp * ldv + n]) : 0.f;
        clipped[t] = live[t] && aseg[t] >= 0 && r_nrm[t] > 0.f;
        sc[t] = r_nrm[t] > 0.f ? 1.f / (r_nrm[t] * e_beta[t] + kap[t]) : 0.f;
      } else {
        clipped[t] = live[t] && kap[t] > 1.f && aseg[t] >= 0;
        sc[t] = 1.f / fmaxf(1.f, kap[t]);
      }
      any |= clipped[t];
#pragma unroll
      for (int tp = 0; tp < NKK; ++tp) u16[t][tp] = zero;
    }

    // bucketed: the items of this group's bucket only (buckets 0 / 1: not clipped / a linear row -- nothing to walk)
    const int it_lo = BUCKET ? (bucket >= 2 ? group_items[2 * (bucket - 2)] : 0) : 0;
    const int it_hi = BUCKET ? (bucket >= 2 ? group_items[2 * (bucket - 2) + 1] : 0) : n_items;
    if (__ballot(any) != 0 && it_hi > it_lo) {  // wave-uniform: a wave of interior samples skips the walk
      const f32x4* wp = Simg + lane + (size_t)it_lo * (NQ * 64);
      f32x4 buf_a[NQ], buf_b[NQ];
      f32x16 wreg[NT];  // step-1 result of a packed tile, masked and scaled: the B operand of step 2
      auto fetch_tile = [&](f32x4 (&buf)[NQ]) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) buf[q] = wp[q * 64];
        wp += NQ * 64;
        __builtin_amdgcn_sched_barrier(0);
      };
      auto process = [&](const BItem item, const f32x4 (&a)[NQ]) {
        if (item.type == BI_NOP) return;
        if (item.type == BI_PACK2) {
          // u += U_tile' w : row tile tp of the transposed tile sits in k-groups 4 tp .. 4 tp + 3
#pragma unroll
          for (int tp = 0; tp < NKK; ++tp)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
              for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                  u16[t][tp] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * tp + q][c], wreg[t][4 * q + c], u16[t][tp], 0, 0, 0);
          return;
        }
        f32x16 acc[NT];
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int t = 0; t < NT; ++t)
              acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][c], vr[t][4 * q + c],
                                                            (q == 0 && c == 0) ? zero : acc[t], 0, 0, 0);
        if (item.type == BI_PACK1) {
          const BPack pk = packs[item.aux_row];
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            bool got = false;
#pragma unroll
            for (int a4 = 0; a4 < 4; ++a4) {
              const int sid = hi ? pk.seg[a4][1] : pk.seg[a4][0];
              const bool mine = clipped[t] && sid >= 0 && sid == aseg[t];
              float qs = 0.f;
#pragma unroll
              for (int c = 0; c < 4; ++c) qs = fmaf(acc[t][4 * a4 + c], acc[t][4 * a4 + c], qs);
              qs = mine ? qs : 0.f;
              if ((pk.pair_bits >> a4) & 1) qs += xhalf(qs);  // the segment's other rows sit in the other half
              const float cw = (mine && qs > 0.f) ? 1.f / sqrtf(qs) : 0.f;
#pragma unroll
              for (int c = 0; c < 4; ++c) wreg[t][4 * a4 + c] = acc[t][4 * a4 + c] * cw;
              got |= mine;
            }
            const int both = (got ? 1 : 0) | __shfl_xor(got ? 1 : 0, 32);
            pmatched[t] |= both != 0;
          }
          return;
        }
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
                u16[t][tp][g] = sel[t] ? acc[t][g] : u16[t][tp][g];
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
              cw[t] = 2.f * inv;
              c0[t] = inv * (-2.f * crs - 2.f * tau * kap[t]);
              c1[t] = inv * 2.f * kap[t];
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
                const int i = 4 * q + c;
                const float u = fmaf(cw[t], u16[t][i / 16][i % 16], fmaf(c0[t], x0[c], c1[t] * x1[c]));
                u16[t][i / 16][i % 16] = sel[t] ? u : u16[t][i / 16][i % 16];
              }
          }
#pragma unroll
          for (int t = 0; t < NT; ++t) matched[t] |= sel[t];
        }
      };
      fetch_tile(buf_a);
      for (int it = it_lo; it < it_hi; it += 2) {  // (an odd count: the partner's fetch is a harmless look-ahead)
        fetch_tile(buf_b);
        process(items[it], buf_a);
        fetch_tile(buf_a);
        if (it + 1 < it_hi) process(items[it + 1], buf_b);
      }
    }
    // a packed quadratic still needs its phi; what matched nothing at all is a linear row
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (!clipped[t] || matched[t]) continue;
      const int rowi = pmatched[t] ? seg_aux[aseg[t]] : arow[t];
      const float* row = Wrow + (int64_t)rowi * NP + 4 * hi;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(row + 8 * q);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int i = 4 * q + c;
          u16[t][i / 16][i % 16] = pmatched[t] ? u16[t][i / 16][i % 16] + x[c] : x[c];
        }
      }
    }

    // grad_v = s t - coef grad kappa, with t formed a second time
    float out[NT][KK];
    {
      float tr[NT][KK];
      pull_back(tr);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (!old_mode) {
          const float coef = clipped[t] ? sc[t] * sc[t] * tv[t] : 0.f;
#pragma unroll
          for (int i = 0; i < KK; ++i) out[t][i] = fmaf(sc[t], tr[t][i], -coef * u16[t][i / 16][i % 16]);
        } else {
          // grad_v = s t - s^2 (t.v) (e^beta v / r + grad kappa),  grad_beta = -s^2 (t.v) r e^beta
          const float coef = sc[t] * sc[t] * tv[t];
          const float dir = r_nrm[t] > 0.f ? e_beta[t] / r_nrm[t] : 0.f;
#pragma unroll
          for (int i = 0; i < KK; ++i)
            out[t][i] = fmaf(sc[t], tr[t][i], -coef * fmaf(dir, vr[t][i], u16[t][i / 16][i % 16]));
          if (live[t] && hi == 0) gv[smp_of[t] * ldgv + n] = -coef * r_nrm[t] * e_beta[t];
        }
      }
    }
    if constexpr (BUCKET) {
      store_rows_ix<NT, NKK, LSTR>(out, gv, ldgv, n, vec_o, rowix, patch, lane);
    } else {
      float one[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) one[t] = 1.f;
      (void)store_rows<NT, NKK, LSTR, true>(out, one, nullptr, gv, ldgv, n, vec_o, s_base, B, live, patch, lane);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------

bool mfma_bwdg_eligible(const RayenPack* p) { return mfma_eligible(p) && bwdg_tiles_eligible(p); }

void mfma_bwdg_free(MfmaBwdgImage* img) {
  if (img == nullptr) return;
  if (img->S) (void)hipFree(img->S);
  if (img->NT) (void)hipFree(img->NT);
  if (img->items) (void)hipFree(img->items);
  if (img->packs) (void)hipFree(img->packs);
  if (img->seg_aux) (void)hipFree(img->seg_aux);
  if (img->Wrow) (void)hipFree(img->Wrow);
  if (img->seg_bucket) (void)hipFree(img->seg_bucket);
  if (img->group_items) (void)hipFree(img->group_items);
  delete img;
}

template <typename T>
static bool upload_vec(const std::vector<T>& host, T** dev, int64_t* bytes) {
  if (hipMalloc(dev, host.size() * sizeof(T)) != hipSuccess) return false;
  if (hipMemcpy(*dev, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return false;
  *bytes += (int64_t)(host.size() * sizeof(T));
  return true;
}

int mfma_bwdg_build(const RayenPack* p, MfmaBwdgImage** out, int64_t* bytes) {
  const int n = p->n, k = p->k, np = n_pad_of(n), nkk = np / 32;
  const double* W = p->W.data();
  TileLayout b(n);
  std::vector<BItem> items;
  std::vector<BPack> packs;
  std::vector<int32_t> seg_aux;
  std::vector<int32_t> seg_group, group_items;
  const int n_real = layout_bwdg_tiles(p, b, items, packs, seg_aux, &seg_group, &group_items);
  std::vector<int32_t> seg_bucket(p->segs.size() + 1, 1);       // 1 = linear rows
  for (size_t sgi = 0; sgi < p->segs.size(); ++sgi)
    if (seg_group[sgi] >= 0) seg_bucket[sgi] = 2 + seg_group[sgi];
  if (group_items.empty()) group_items.assign(2, 0);

  MfmaBwdgImage* img = new MfmaBwdgImage();
  img->n_groups = (int)(group_items.size() / 2) - (n_real == 0 ? 1 : 0);
  for (int gi = 0; gi < img->n_groups; ++gi) img->n_group_tiles += group_items[2 * gi + 1] - group_items[2 * gi];
  img->nkk = nkk;
  img->nkg = p->out_identity ? 0 : n_pad_of(k) / 32;
  img->n_items = n_real;
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, p->device) == hipSuccess && prop.multiProcessorCount > 0)
      img->n_simd = prop.multiProcessorCount * 4;
  }
  const std::vector<float> frag = b.fragments_f32();
  std::vector<float> wrow((size_t)(p->n_rows + 2) * np, 0.f);
  for (int r = 0; r < p->n_rows; ++r)
    for (int j = 0; j < n; ++j) wrow[(size_t)r * np + j] = (float)W[(size_t)r * n + j];
  bool ok = true;
  {
    float* d = nullptr;
    ok = ok && upload_vec(frag, &d, &img->bytes);
    img->S = reinterpret_cast<f32x4*>(d);
  }
  ok = ok && upload_vec(wrow, &img->Wrow, &img->bytes) && upload_vec(items, &img->items, &img->bytes) &&
       upload_vec(packs, &img->packs, &img->bytes) && upload_vec(seg_aux, &img->seg_aux, &img->bytes) &&
       upload_vec(seg_bucket, &img->seg_bucket, &img->bytes) && upload_vec(group_items, &img->group_items, &img->bytes);
  if (ok && !p->out_identity) {
    // NA_E' : rows = the n subspace coordinates, K = the k ambient coordinates
    TileLayout bn(k);
    std::vector<std::vector<double>> nt(n, std::vector<double>(k, 0.0));
    for (int i = 0; i < k; ++i)
      for (int e = 0; e < n; ++e) nt[e][i] = p->NA_E[(size_t)i * n + e];
    for (int tp = 0; tp < nkk; ++tp) {
      std::vector<const double*> rows;
      for (int r = 32 * tp; r < 32 * tp + 32 && r < n; ++r) rows.push_back(nt[r].data());
      bn.add_tile(rows, k);
    }
    const std::vector<float> fn = bn.fragments_f32();
    float* d = nullptr;
    ok = upload_vec(fn, &d, &img->bytes);
    img->NT = reinterpret_cast<f32x4*>(d);
  }
  if (!ok) { mfma_bwdg_free(img); return RAYEN_E_ALLOC; }
  *bytes = img->bytes;
  *out = img;
  return RAYEN_OK;
}

int64_t mfma_bwdg_workspace_bytes(const RayenPack* p, const MfmaBwdgImage* img, int64_t B) {
  (void)p;
  return img == nullptr ? 0 : bucket_workspace_bytes_groups(img->n_groups, img->n_group_tiles, B);
}

template <int NKK, int NKG>
static int launch_bwdg(const RayenPack* p, const MfmaBwdgImage* img, const float* v, int64_t B, int64_t ldv,
                       const float* kappa, const int32_t* active, const float* gy, int64_t ldg, float* gv,
                       int64_t ldgv, int old_mode, void* workspace, int64_t workspace_bytes, hipStream_t stream) {
  constexpr int per_wave = NKK == 1 ? 64 : 32;
  const int64_t need = old_mode ? 0 : mfma_bwdg_workspace_bytes(p, img, B);
  const bool bucketed = need > 0 && workspace != nullptr && workspace_bytes >= need;
  const int nb = img->n_groups + 2;
  int32_t* ws = static_cast<int32_t*>(workspace);
  if (bucketed) launch_bucket_sort<float>(kappa, active, B, img->seg_bucket, nb, ws, stream);
  const int64_t n_groups = (B + per_wave - 1) / per_wave + (bucketed ? (64 / per_wave) * nb : 0);
  const int64_t slots = (int64_t)launch_simds(img->n_simd) * kMfmaWavesPerSimd;
  const int64_t rounds = (n_groups + slots - 1) / slots;
  const int64_t waves = (n_groups + rounds - 1) / rounds;
  const int64_t grid = (waves + kMfmaWaves - 1) / kMfmaWaves;
  auto aligned = [](const void* ptr, int64_t ld) { return (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(ptr) & 15) == 0); };
  if (bucketed)
    hipLaunchKernelGGL((mfma_bwdg_kernel<NKK, NKG, true>), dim3((unsigned)grid), dim3(kMfmaWaves * 64), 0, stream, img->S,
                       img->NT, img->items, img->n_items, img->packs, img->seg_aux, img->Wrow, p->n, p->k, v, B, ldv,
                       aligned(v, ldv) ? 1 : 0, kappa, active, gy, ldg, aligned(gy, ldg) ? 1 : 0, gv, ldgv,
                       aligned(gv, ldgv) ? 1 : 0, 0, static_cast<const int32_t*>(ws), nb, img->group_items);
  else
    hipLaunchKernelGGL((mfma_bwdg_kernel<NKK, NKG, false>), dim3((unsigned)grid), dim3(kMfmaWaves * 64), 0, stream, img->S,
                       img->NT, img->items, img->n_items, img->packs, img->seg_aux, img->Wrow, p->n, p->k, v, B, ldv,
                       aligned(v, ldv) ? 1 : 0, kappa, active, gy, ldg, aligned(gy, ldg) ? 1 : 0, gv, ldgv,
                       aligned(gv, ldgv) ? 1 : 0, old_mode, static_cast<const int32_t*>(nullptr), 0,
                       static_cast<const int32_t*>(nullptr));
  return hipGetLastError() == hipSuccess ? RAYEN_OK : RAYEN_E_LAUNCH;
}

int mfma_bwdg_backward(const RayenPack* p, const MfmaBwdgImage* img, const float* v, int64_t B, int64_t ldv,
                       const float* kappa, const int32_t* active, const float* grad_y, int64_t ldg,
                       float* grad_v, int64_t ldgv, int old_mode, void* workspace, int64_t workspace_bytes,
                       hipStream_t stream) {
  if (B == 0) return RAYEN_OK;
#define RAYEN_BWDG_CASE(A, G) \
  if (img->nkk == A && img->nkg == G) \
    return launch_bwdg<A, G>(p, img, v, B, ldv, kappa, active, grad_y, ldg, grad_v, ldgv, old_mode, workspace, \
                             workspace_bytes, stream);
  RAYEN_BWDG_CASE(1, 0)
  RAYEN_BWDG_CASE(1, 1)
  RAYEN_BWDG_CASE(1, 2)
  RAYEN_BWDG_CASE(2, 0)
  RAYEN_BWDG_CASE(2, 2)
#undef RAYEN_BWDG_CASE
  return RAYEN_E_UNSUPPORTED;
}

}  // namespace rayen
