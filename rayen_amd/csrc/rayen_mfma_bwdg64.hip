// fp64 twin of rayen_mfma_bwdg.hip (the reference trains its corridor sets in fp64): matrix-core backward
// for sets with equality constraints and / or packed low-rank quadratics, n <= 32, k <= 64, on
// v_mfma_f64_16x16x4_f64 with the lane layout of rayen_mfma_f64.hip -- lane l holds sample l&15 of a
// 16-sample column block and, with q = l>>4, element 4 s + q of a vector as the B operand of K-step s, which
// is also the row it holds in result register s&3 of row half (s>>2)&1.  Results of one chain are therefore
// valid B operands of the next, which is what both the pull-back t = NA_E' g and the masked two-step
// product u = U_s'(U_s v)/||U_s v|| of the packed quadratics rely on (see rayen_mfma_bwdg.hip).
#include "rayen_bwd_tiles.h"
#include "rayen_internal.h"

namespace rayen {

using f64x4 = double __attribute__((ext_vector_type(4)));
using f64x2 = double __attribute__((ext_vector_type(2)));

struct Mfma64BwdgImage {
  f64x2* S = nullptr;        // item tiles: [tile][step pair][row half][lane] x 2 doubles (K = 32)
  f64x2* NT = nullptr;       // NA_E' (32 rows x k_pad), same order with K = k_pad; null when NA_E = I
  BItem* items = nullptr;
  BPack* packs = nullptr;
  int32_t* seg_aux = nullptr;
  double* Wrow = nullptr;    // [n_rows + 2][32]
  int n_items = 0, nkg = 0, n_simd = 1024;
  int64_t bytes = 0;
};

constexpr int kG64Waves = 8;

__device__ __forceinline__ double gq16(double x) { return __shfl_xor(x, 16); }
__device__ __forceinline__ double gq32(double x) { return __shfl_xor(x, 32); }
__device__ __forceinline__ double gsum(double x) {
  x += gq16(x);
  x += gq32(x);
  return x;
}

template <int NKG>
__global__ __launch_bounds__(kG64Waves * 64, 2) void mfma64_bwdg_kernel(
    const f64x2* __restrict__ Simg, const f64x2* __restrict__ NTimg, const BItem* __restrict__ items, int n_items,
    const BPack* __restrict__ packs, const int32_t* __restrict__ seg_aux, const double* __restrict__ Wrow, int n,
    int k, const double* __restrict__ v, int64_t B, int64_t ldv, const double* __restrict__ kappa,
    const int32_t* __restrict__ active, const double* __restrict__ gy, int64_t ldg, double* __restrict__ gv,
    int64_t ldgv, int old_mode) {
  constexpr int NS = 8, NP = 32;                    // one 32-column block of v: 8 K-steps
  constexpr int NSG = NKG > 0 ? NKG * 8 : 1;        // K-steps of the pull-back
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int j = lane & 15;
  const int q = lane >> 4;
  const int64_t n_groups = (B + 31) / 32;
  const int64_t wave_id = (int64_t)blockIdx.x * kG64Waves + wave;
  const int64_t wave_stride = (int64_t)gridDim.x * kG64Waves;
  const f64x4 zero4 = {0.0, 0.0, 0.0, 0.0};

  for (int64_t grp = wave_id; grp < n_groups; grp += wave_stride) {
    const int64_t s_base = grp * 32;
    double vb[2][NS];
    f64x4 ub[2][2];  // grad kappa: [row half][column block], register g = element 16 rh + 4 g + q
    bool live[2], clipped[2], matched[2], pmatched[2];
    double kap[2], tv[2], sc[2], r_nrm[2], e_beta[2];
    int aseg[2], arow[2];

    // t = NA_E' g (or g itself): [column block][K-step]
    auto pull_back = [&](double (&tb)[2][NS]) {
      if constexpr (NKG == 0) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int64_t s = s_base + 16 * c + j;
          const double* grow = gy + (s < B ? s : 0) * ldg;
#pragma unroll
          for (int st = 0; st < NS; ++st) tb[c][st] = (s < B && 4 * st + q < n) ? grow[4 * st + q] : 0.0;
        }
      } else {
        double gb[2][NSG];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int64_t s = s_base + 16 * c + j;
          const double* grow = gy + (s < B ? s : 0) * ldg;
#pragma unroll
          for (int st = 0; st < NSG; ++st) gb[c][st] = (s < B && 4 * st + q < k) ? grow[4 * st + q] : 0.0;
        }
        f64x4 acc[2][2];
#pragma unroll
        for (int sp = 0; sp < NSG / 2; ++sp) {
          f64x2 a[2];
#pragma unroll
          for (int rh = 0; rh < 2; ++rh) a[rh] = NTimg[((size_t)sp * 2 + rh) * 64 + lane];
#pragma unroll
          for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int rh = 0; rh < 2; ++rh)
#pragma unroll
              for (int c = 0; c < 2; ++c)
                acc[rh][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[rh][e], gb[c][2 * sp + e],
                                                                  (sp == 0 && e == 0) ? zero4 : acc[rh][c], 0, 0, 0);
        }
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int rh = 0; rh < 2; ++rh)
#pragma unroll
            for (int g = 0; g < 4; ++g) tb[c][4 * rh + g] = acc[rh][c][g];
      }
    };

    {
      double tb[2][NS];
      pull_back(tb);
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int64_t s = s_base + 16 * c + j;
        live[c] = s < B;
        const double* row = v + (live[c] ? s : 0) * ldv;
        double dot = 0.0, nrm2 = 0.0;
#pragma unroll
        for (int st = 0; st < NS; ++st) {
          vb[c][st] = (live[c] && 4 * st + q < n) ? row[4 * st + q] : 0.0;
          dot = fma(tb[c][st], vb[c][st], dot);
          nrm2 = fma(vb[c][st], vb[c][st], nrm2);
        }
        tv[c] = gsum(dot);
        kap[c] = live[c] ? kappa[s] : 0.0;
        aseg[c] = live[c] ? active[2 * s] : -1;
        arow[c] = live[c] ? active[2 * s + 1] : 0;
        matched[c] = false;
        pmatched[c] = false;
        r_nrm[c] = 0.0;
        e_beta[c] = 0.0;
        if (old_mode) {
          r_nrm[c] = sqrt(gsum(nrm2));
          e_beta[c] = live[c] ? exp(row[n]) : 0.0;
          clipped[c] = live[c] && aseg[c] >= 0 && r_nrm[c] > 0.0;
          sc[c] = r_nrm[c] > 0.0 ? 1.0 / (r_nrm[c] * e_beta[c] + kap[c]) : 0.0;
        } else {
          clipped[c] = live[c] && kap[c] > 1.0 && aseg[c] >= 0;
          sc[c] = 1.0 / fmax(1.0, kap[c]);
        }
        ub[0][c] = zero4;
        ub[1][c] = zero4;
      }
    }

    if (__ballot(clipped[0] || clipped[1]) != 0 && n_items > 0) {
      const f64x2* wp = Simg + lane;
      f64x2 buf_a[NS / 2][2], buf_b[NS / 2][2];  // a whole tile: [step pair][row half]
      f64x4 wv[2][2];                            // masked, scaled step-1 result of a packed tile
      auto fetch_tile = [&](f64x2 (&buf)[NS / 2][2]) {
#pragma unroll
        for (int sp = 0; sp < NS / 2; ++sp)
#pragma unroll
          for (int rh = 0; rh < 2; ++rh) buf[sp][rh] = wp[(sp * 2 + rh) * 64];
        wp += (NS / 2) * 2 * 64;
        __builtin_amdgcn_sched_barrier(0);
      };
      auto process = [&](const BItem item, const f64x2 (&a)[NS / 2][2]) {
        if (item.type == BI_NOP) return;
        if (item.type == BI_PACK2) {
          // u += U_tile' w : K-step s reads w element 4 s + q = register s&3 of row half s>>2
#pragma unroll
          for (int sp = 0; sp < NS / 2; ++sp)
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
              for (int rh = 0; rh < 2; ++rh)
#pragma unroll
                for (int c = 0; c < 2; ++c)
                  ub[rh][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[sp][rh][e], wv[(2 * sp + e) >> 2][c][(2 * sp + e) & 3],
                                                                   ub[rh][c], 0, 0, 0);
          return;
        }
        f64x4 acc[2][2];
#pragma unroll
        for (int sp = 0; sp < NS / 2; ++sp)
#pragma unroll
          for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int rh = 0; rh < 2; ++rh)
#pragma unroll
              for (int c = 0; c < 2; ++c)
                acc[rh][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[sp][rh][e], vb[c][2 * sp + e],
                                                                  (sp == 0 && e == 0) ? zero4 : acc[rh][c], 0, 0, 0);
        if (item.type == BI_PACK1) {
          // quad m = rows 4m..4m+3 = register (rh = m>>2, g = m&3) of the four lane groups; slot m = 2a + h
          const BPack pk = packs[item.aux_row];
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            bool got = false;
#pragma unroll
            for (int a4 = 0; a4 < 4; ++a4) {
              double qs[2];
              bool mine[2];
#pragma unroll
              for (int h = 0; h < 2; ++h) {
                const int m = 2 * a4 + h;
                const int sid = pk.seg[a4][h];
                mine[h] = clipped[c] && sid >= 0 && sid == aseg[c];
                const double x = acc[m >> 2][c][m & 3];
                qs[h] = gsum(mine[h] ? x * x : 0.0);
              }
              if ((pk.pair_bits >> a4) & 1) { qs[0] += qs[1]; qs[1] = qs[0]; }
#pragma unroll
              for (int h = 0; h < 2; ++h) {
                const int m = 2 * a4 + h;
                const double cw = (mine[h] && qs[h] > 0.0) ? 1.0 / sqrt(qs[h]) : 0.0;
                wv[m >> 2][c][m & 3] = acc[m >> 2][c][m & 3] * cw;
                got |= mine[h];
              }
            }
            pmatched[c] |= got;
          }
          return;
        }
        // dense form of one quadratic / cone (n <= 32: a single tile, first and last at once)
        bool sel[2];
        double total[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          sel[c] = clipped[c] && aseg[c] == item.seg;
          double sum = 0.0;
#pragma unroll
          for (int rh = 0; rh < 2; ++rh)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              sum = fma(acc[rh][c][g], vb[c][4 * rh + g], sum);
              ub[rh][c][g] = sel[c] ? acc[rh][c][g] : ub[rh][c][g];
            }
          total[c] = gsum(sum);  // v'S v
        }
        if (__ballot(sel[0] || sel[1]) != 0) {
          const double* ax = Wrow + (int64_t)item.aux_row * NP + q;
          double cw[2], c0[2], c1[2];
          if (item.type == BI_QUAD) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
              cw[c] = total[c] > 0.0 ? 1.0 / sqrt(total[c]) : 0.0;
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
              const double crs = gsum(cr[c]), brs = gsum(br[c]);
              const double tau = item.f0d, ap = item.f1d;
              const double bp = 2.0 * brs - 2.0 * crs * tau;
              const double den = 2.0 * ap * kap[c] + bp;  // dF/dkappa at the root
              const double inv = den != 0.0 ? -1.0 / den : 0.0;
              cw[c] = 2.0 * inv;
              c0[c] = inv * (-2.0 * crs - 2.0 * tau * kap[c]);
              c1[c] = inv * 2.0 * kap[c];
            }
          }
#pragma unroll
          for (int rh = 0; rh < 2; ++rh)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int st = 4 * rh + g;
              const double x0 = ax[4 * st];
              const double x1 = item.type == BI_SOC ? ax[NP + 4 * st] : 0.0;
#pragma unroll
              for (int c = 0; c < 2; ++c) {
                const double u = fma(cw[c], ub[rh][c][g], fma(c0[c], x0, c1[c] * x1));
                ub[rh][c][g] = sel[c] ? u : ub[rh][c][g];
              }
            }
#pragma unroll
          for (int c = 0; c < 2; ++c) matched[c] |= sel[c];
        }
      };
      fetch_tile(buf_a);
      for (int it = 0; it < n_items; it += 2) {  // n_items is even (padded with a no-op tile)
        fetch_tile(buf_b);
        process(items[it], buf_a);
        fetch_tile(buf_a);
        process(items[it + 1], buf_b);
      }
    }
    // a packed quadratic still needs its phi; what matched nothing at all is a linear row
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      if (!clipped[c] || matched[c]) continue;
      const int rowi = pmatched[c] ? seg_aux[aseg[c]] : arow[c];
      const double* row = Wrow + (int64_t)rowi * NP + q;
#pragma unroll
      for (int rh = 0; rh < 2; ++rh)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const double x = row[4 * (4 * rh + g)];
          ub[rh][c][g] = pmatched[c] ? ub[rh][c][g] + x : x;
        }
    }

    // grad_v = s t - coef grad kappa, with t formed a second time
    {
      double tb[2][NS];
      pull_back(tb);
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        if (!live[c]) continue;
        const int64_t s = s_base + 16 * c + j;
        double* orow = gv + s * ldgv;
        if (!old_mode) {
          const double coef = clipped[c] ? sc[c] * sc[c] * tv[c] : 0.0;
#pragma unroll
          for (int st = 0; st < NS; ++st) {
            const int r = 4 * st + q;
            if (r < n) orow[r] = fma(sc[c], tb[c][st], -coef * ub[st >> 2][c][st & 3]);
          }
        } else {
          // grad_v = s t - s^2 (t.v) (e^beta v / r + grad kappa),  grad_beta = -s^2 (t.v) r e^beta
          const double coef = sc[c] * sc[c] * tv[c];
          const double dir = r_nrm[c] > 0.0 ? e_beta[c] / r_nrm[c] : 0.0;
#pragma unroll
          for (int st = 0; st < NS; ++st) {
            const int r = 4 * st + q;
            if (r < n) orow[r] = fma(sc[c], tb[c][st], -coef * fma(dir, vb[c][st], ub[st >> 2][c][st & 3]));
          }
          if (q == 0) orow[n] = -coef * r_nrm[c] * e_beta[c];
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------

bool mfma64_bwdg_eligible(const RayenPack* p) { return p->n <= 32 && mfma64_eligible(p) && bwdg_tiles_eligible(p); }

void mfma64_bwdg_free(Mfma64BwdgImage* img) {
  if (img == nullptr) return;
  if (img->S) (void)hipFree(img->S);
  if (img->NT) (void)hipFree(img->NT);
  if (img->items) (void)hipFree(img->items);
  if (img->packs) (void)hipFree(img->packs);
  if (img->seg_aux) (void)hipFree(img->seg_aux);
  if (img->Wrow) (void)hipFree(img->Wrow);
  delete img;
}

// [tile][step pair sg][row half rh][lane l][2]: raw[16 rh + (l&15)][4 (2 sg + e) + (l>>4)], e = 0, 1
static std::vector<double> fragments_f64(const TileLayout& b) {
  const int nt = b.n_tiles(), ns = b.n_pad / 4;
  std::vector<double> frag((size_t)nt * (ns / 2) * 2 * 64 * 2, 0.0);
  for (int t = 0; t < nt; ++t)
    for (int sg = 0; sg < ns / 2; ++sg)
      for (int rh = 0; rh < 2; ++rh)
        for (int l = 0; l < 64; ++l)
          for (int e = 0; e < 2; ++e)
            frag[((((size_t)t * (ns / 2) + sg) * 2 + rh) * 64 + l) * 2 + e] =
                b.raw[((size_t)t * 32 + 16 * rh + (l & 15)) * b.n_pad + 4 * (2 * sg + e) + (l >> 4)];
  return frag;
}

template <typename T>
static bool upload64(const std::vector<T>& host, T** dev, int64_t* bytes) {
  if (hipMalloc(dev, host.size() * sizeof(T)) != hipSuccess) return false;
  if (hipMemcpy(*dev, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return false;
  *bytes += (int64_t)(host.size() * sizeof(T));
  return true;
}

int mfma64_bwdg_build(const RayenPack* p, Mfma64BwdgImage** out, int64_t* bytes) {
  const int n = p->n, k = p->k, np = n_pad_of(n);
  TileLayout b(n);
  std::vector<BItem> items;
  std::vector<BPack> packs;
  std::vector<int32_t> seg_aux;
  const int n_real = layout_bwdg_tiles(p, b, items, packs, seg_aux);

  Mfma64BwdgImage* img = new Mfma64BwdgImage();
  img->nkg = p->out_identity ? 0 : n_pad_of(k) / 32;
  img->n_items = n_real;
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, p->device) == hipSuccess && prop.multiProcessorCount > 0)
      img->n_simd = prop.multiProcessorCount * 4;
  }
  const std::vector<double> frag = fragments_f64(b);
  std::vector<double> wrow((size_t)(p->n_rows + 2) * np, 0.0);
  for (int r = 0; r < p->n_rows; ++r)
    for (int c = 0; c < n; ++c) wrow[(size_t)r * np + c] = p->W[(size_t)r * n + c];
  bool ok = true;
  {
    double* d = nullptr;
    ok = ok && upload64(frag, &d, &img->bytes);
    img->S = reinterpret_cast<f64x2*>(d);
  }
  ok = ok && upload64(wrow, &img->Wrow, &img->bytes) && upload64(items, &img->items, &img->bytes) &&
       upload64(packs, &img->packs, &img->bytes) && upload64(seg_aux, &img->seg_aux, &img->bytes);
  if (ok && !p->out_identity) {
    TileLayout bn(k);  // NA_E': rows = the n subspace coordinates (one tile), K = the k ambient coordinates
    std::vector<std::vector<double>> nt(n, std::vector<double>(k, 0.0));
    for (int i = 0; i < k; ++i)
      for (int e = 0; e < n; ++e) nt[e][i] = p->NA_E[(size_t)i * n + e];
    std::vector<const double*> rows;
    for (int r = 0; r < n; ++r) rows.push_back(nt[r].data());
    bn.add_tile(rows, k);
    const std::vector<double> fn = fragments_f64(bn);
    double* d = nullptr;
    ok = upload64(fn, &d, &img->bytes);
    img->NT = reinterpret_cast<f64x2*>(d);
  }
  if (!ok) { mfma64_bwdg_free(img); return RAYEN_E_ALLOC; }
  *bytes = img->bytes;
  *out = img;
  return RAYEN_OK;
}

template <int NKG>
static int launch_bwdg64(const RayenPack* p, const Mfma64BwdgImage* img, const double* v, int64_t B, int64_t ldv,
                         const double* kappa, const int32_t* active, const double* gy, int64_t ldg, double* gv,
                         int64_t ldgv, int old_mode, hipStream_t stream) {
  const int64_t n_groups = (B + 31) / 32;
  const int64_t slots = (int64_t)launch_simds(img->n_simd) * 2;
  const int64_t rounds = (n_groups + slots - 1) / slots;
  const int64_t waves = (n_groups + rounds - 1) / rounds;
  const int64_t grid = (waves + kG64Waves - 1) / kG64Waves;
  hipLaunchKernelGGL((mfma64_bwdg_kernel<NKG>), dim3((unsigned)grid), dim3(kG64Waves * 64), 0, stream, img->S,
                     img->NT, img->items, img->n_items, img->packs, img->seg_aux, img->Wrow, p->n, p->k, v, B, ldv,
                     kappa, active, gy, ldg, gv, ldgv, old_mode);
  return hipGetLastError() == hipSuccess ? RAYEN_OK : RAYEN_E_LAUNCH;
}

int mfma64_bwdg_backward(const RayenPack* p, const Mfma64BwdgImage* img, const double* v, int64_t B, int64_t ldv,
                         const double* kappa, const int32_t* active, const double* grad_y, int64_t ldg,
                         double* grad_v, int64_t ldgv, int old_mode, hipStream_t stream) {
  if (B == 0) return RAYEN_OK;
  switch (img->nkg) {
    case 0: return launch_bwdg64<0>(p, img, v, B, ldv, kappa, active, grad_y, ldg, grad_v, ldgv, old_mode, stream);
    case 1: return launch_bwdg64<1>(p, img, v, B, ldv, kappa, active, grad_y, ldg, grad_v, ldgv, old_mode, stream);
    case 2: return launch_bwdg64<2>(p, img, v, B, ldv, kappa, active, grad_y, ldg, grad_v, ldgv, old_mode, stream);
    default: return RAYEN_E_UNSUPPORTED;
  }
}

}  // namespace rayen
