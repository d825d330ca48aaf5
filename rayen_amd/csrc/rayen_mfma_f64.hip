// fp64 MFMA path: the same tile walk as rayen_mfma.hip on v_mfma_f64_16x16x4_f64.
//
// The reference is trained in fp64 (examples/main.py:288 sets the default dtype to float64 and
// calls it "very important"), so fp64 is a first-class input type here.  One wave owns 32 samples
// (two 16-sample column blocks) and walks the 32-row tiles of W; a tile is two 16-row halves, so a
// K-step (4 columns) issues four independent MFMA chains (row half x column block):
//
//   D[row][sample] of one 16 x 16 block: lane l holds sample l&15 and rows (l>>4) + 4g, g = 0..3
//
// With q = l>>4 the B operand of K-step s is the direction element 4s + q, which is also the row
// this lane holds in result register g = s&3 of row half (s>>2)&1 of tile s>>3: the symmetric-form
// epilogue v'Gv, and the NA_E = I write-out, read their v values from the B-operand registers, as in
// the fp32 kernel.  Reductions over a tile's rows are 8 registers in-lane plus two exchanges across
// the four lane groups.  A operands stream from the L2-resident fragment image through two half-tile
// register buffers (loads issued half a tile = 32 MFMAs ahead).
#include "rayen_internal.h"
#include "rayen_tiles.h"

namespace rayen {

using f64x4 = double __attribute__((ext_vector_type(4)));
using f64x2 = double __attribute__((ext_vector_type(2)));

struct Mfma64Image {
  f64x2* W = nullptr;   // [tile][step pair][row half][lane] x 2 doubles
  MItem* items = nullptr;
  MPack* packs = nullptr;
  double* y0 = nullptr;  // [k_pad]
  int n_items = 0;
  int nkk = 0;
  int identity = 0;
  int n_simd = 1024;
  int64_t bytes = 0;
};

constexpr int k64Waves = 8;

template <typename T>
__device__ __forceinline__ T xq16(T x) { return __shfl_xor(x, 16); }
template <typename T>
__device__ __forceinline__ T xq32(T x) { return __shfl_xor(x, 32); }

template <int NKK, bool TRACK>
__global__ __launch_bounds__(k64Waves * 64, 2) void mfma64_fwd_kernel(
    const f64x2* __restrict__ Wimg, const MItem* __restrict__ items, int n_items,
    const MPack* __restrict__ packs, const double* __restrict__ y0, int identity, int k, int n, const double* __restrict__ v, int64_t B,
    int64_t ldv, double* __restrict__ y, int64_t ldy, double* __restrict__ kappa_out,
    int32_t* __restrict__ active_out, int32_t* __restrict__ nan_flag, int old_mode) {
  constexpr int NS = NKK * 8;   // K-steps (4 columns each) per tile
  __shared__ double aux_lds[k64Waves][2][32][16];  // [wave][column block][aux row][sample]

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int j = lane & 15;   // sample within the column block
  const int q = lane >> 4;   // lane group: row offset inside a group of four rows / column inside a K-step
  const int64_t n_groups = (B + 31) / 32;
  const int64_t wave_id = (int64_t)blockIdx.x * k64Waves + wave;
  const int64_t wave_stride = (int64_t)gridDim.x * k64Waves;
  bool bad = false;

  for (int64_t grp = wave_id; grp < n_groups; grp += wave_stride) {
    const int64_t s_base = grp * 32;

    // ---- B operands: element 4s + q of each of this lane's two samples
    double vb[2][NS];
    bool live[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int64_t s = s_base + 16 * c + j;
      live[c] = s < B;
      const double* row = v + (live[c] ? s : 0) * ldv;
#pragma unroll
      for (int st = 0; st < NS; ++st) vb[c][st] = (live[c] && 4 * st + q < n) ? row[4 * st + q] : 0.0;
    }

    // RAYEN_old head (rayen/constraint_module.py:460-466): y = y0 + N v / (||v|| e^beta + kappa(v))
    double old_den[2] = {0.0, 0.0};
    if (old_mode) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        double nrm2 = 0.0;
#pragma unroll
        for (int st = 0; st < NS; ++st) nrm2 = fma(vb[c][st], vb[c][st], nrm2);
        nrm2 += xq16(nrm2);
        nrm2 += xq32(nrm2);
        const double beta = live[c] ? v[(s_base + 16 * c + j) * ldv + n] : 0.0;
        old_den[c] = sqrt(nrm2) * exp(beta);
      }
    }
    double kap[2], part[2], scale[2];
    int aseg[2], arow[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) { kap[c] = 0.0; part[c] = 0.0; scale[c] = 1.0; aseg[c] = -1; arow[c] = 0; }

    // ---- A operands.  Image order [tile][step pair sg][row half rh][lane]: one 16-byte piece holds
    // the lane's A values of steps 2sg and 2sg+1 for row half rh.  A tile half = NS/4 step pairs.
    const f64x2* wp = Wimg + lane;
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

    auto finish_kappa = [&]() {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        // max over the four lane groups; ties go to the lower group so that all four agree
        int who = q;
#pragma unroll
        for (int step = 0; step < 2; ++step) {
          const double ok = step == 0 ? xq16(kap[c]) : xq32(kap[c]);
          const int owho = step == 0 ? xq16(who) : xq32(who);
          const int oseg = step == 0 ? xq16(aseg[c]) : xq32(aseg[c]);
          const int orow = step == 0 ? xq16(arow[c]) : xq32(arow[c]);
          if (ok > kap[c] || (ok == kap[c] && owho < who)) { kap[c] = ok; who = owho; aseg[c] = oseg; arow[c] = orow; }
        }
        scale[c] = 1.0 / fmax(1.0, kap[c]);
        if (old_mode) scale[c] = old_den[c] > 0.0 ? 1.0 / (old_den[c] + kap[c]) : 0.0;
      }
    };

    f64x4 acc[2][2];  // [row half][column block]
    for (int it = 0; it < n_items; ++it) {
      const MItem item = items[it];
      if (item.type == MI_NOP) {  // pairing filler of the fp32 kernel: skip its (zero) tile
        fetch_half(buf_lo);
        fetch_half(buf_hi);
        continue;
      }
      if (item.type == MI_OUT && (item.flags & MF_FIRST)) finish_kappa();
      const int sbegin = 2 * item.qbegin;  // qbegin counts 8-column groups, a K-step is 4 columns
      // The first MFMA of each of the four chains takes the constant 0 as its C operand (no accumulator
      // initialisation: VALU work and MFMA issue of a SIMD are serial).  Whole 32-column blocks before
      // sbegin were folded into their transposes (wave-uniform): with NKK = 2 that is the buf_lo half.
      const f64x4 zero4 = {0.0, 0.0, 0.0, 0.0};
      const bool skip_lo = (NKK == 2) && sbegin >= 8;
      if constexpr (TRACK) {
        // (with the arg-max bookkeeping the second copy of the buf_hi block costs more registers than the
        // initialisation saves: explicit zeroes, one copy of each block)
#pragma unroll
        for (int rh = 0; rh < 2; ++rh)
#pragma unroll
          for (int c = 0; c < 2; ++c) acc[rh][c] = zero4;
#pragma unroll
        for (int p = 0; p < NS / 4; ++p) {
          if (((2 * p) & ~7) < sbegin) continue;
#pragma unroll
          for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int rh = 0; rh < 2; ++rh)
#pragma unroll
              for (int c = 0; c < 2; ++c)
                acc[rh][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(buf_lo[p][rh][e], vb[c][2 * p + e], acc[rh][c], 0, 0, 0);
        }
        fetch_half(buf_lo);
#pragma unroll
        for (int p = 0; p < NS / 4; ++p) {
          if (((NS / 2 + 2 * p) & ~7) < sbegin) continue;
#pragma unroll
          for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int rh = 0; rh < 2; ++rh)
#pragma unroll
              for (int c = 0; c < 2; ++c)
                acc[rh][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(buf_hi[p][rh][e], vb[c][NS / 2 + 2 * p + e], acc[rh][c], 0, 0, 0);
        }
        fetch_half(buf_hi);
      } else {
      if (!skip_lo) {
#pragma unroll
        for (int p = 0; p < NS / 4; ++p)
#pragma unroll
          for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int rh = 0; rh < 2; ++rh)
#pragma unroll
              for (int c = 0; c < 2; ++c)
                acc[rh][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(buf_lo[p][rh][e], vb[c][2 * p + e],
                                                                  (p == 0 && e == 0) ? zero4 : acc[rh][c], 0, 0, 0);
      }
      fetch_half(buf_lo);
      if (skip_lo) {
#pragma unroll
        for (int p = 0; p < NS / 4; ++p)
#pragma unroll
          for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int rh = 0; rh < 2; ++rh)
#pragma unroll
              for (int c = 0; c < 2; ++c)
                acc[rh][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(buf_hi[p][rh][e], vb[c][NS / 2 + 2 * p + e],
                                                                  (p == 0 && e == 0) ? zero4 : acc[rh][c], 0, 0, 0);
      } else {
#pragma unroll
        for (int p = 0; p < NS / 4; ++p)
#pragma unroll
          for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int rh = 0; rh < 2; ++rh)
#pragma unroll
              for (int c = 0; c < 2; ++c)
                acc[rh][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(buf_hi[p][rh][e], vb[c][NS / 2 + 2 * p + e],
                                                                  acc[rh][c], 0, 0, 0);
      }
      fetch_half(buf_hi);
      }

      // ---- epilogue: this lane holds rows 16 rh + 4 g + q of the tile for its two samples
      if (item.type == MI_LIN) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int rh = 0; rh < 2; ++rh)
#pragma unroll
            for (int g = 0; g < 4; ++g)
              if (acc[rh][c][g] > kap[c]) {
                kap[c] = acc[rh][c][g];
                if (TRACK) { aseg[c] = item.seg; arow[c] = item.row0 + 16 * rh + 4 * g + q; }
              }
      } else if (item.type == MI_AUX) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int rh = 0; rh < 2; ++rh)
#pragma unroll
            for (int g = 0; g < 4; ++g) aux_lds[wave][c][16 * rh + 4 * g + q][j] = acc[rh][c][g];
        __builtin_amdgcn_wave_barrier();
      } else if (item.type == MI_OUT) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          if (!live[c]) continue;
          double* yrow = y + (s_base + 16 * c + j) * ldy;
#pragma unroll
          for (int rh = 0; rh < 2; ++rh)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int r = item.row0 + 16 * rh + 4 * g + q;
              if (r < k) {
                const double o = fma(acc[rh][c][g], scale[c], y0[r]);
                bad |= (o != o);
                yrow[r] = o;
              }
            }
        }
      } else if (item.type == MI_PACK) {
        // eight small factor segments per tile: quad m = rows 4m..4m+3 = register (rh = m>>2, g = m&3)
        // of the four lane groups, so ||U v||^2 of a quad is one square summed across the groups
        const MPack pk = packs[item.aux];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const bool pair = (item.row0 >> a) & 1;
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            double qs[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const int m = 2 * a + h;
              double sq = acc[m >> 2][c][m & 3] * acc[m >> 2][c][m & 3];
              sq += xq16(sq);
              sq += xq32(sq);
              qs[h] = sq;
            }
            if (pair) { qs[0] += qs[1]; qs[1] = qs[0]; }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const int sid = pk.seg[a][h];
              const double kc = aux_lds[wave][c][pk.aux[a][h] & 31][j] + sqrt(qs[h]);
              if (sid >= 0 && kc > kap[c]) { kap[c] = kc; aseg[c] = sid; arow[c] = 0; }
            }
          }
        }
      } else {
        // symmetric form (acc . v), or factor rows (acc . acc); closed on the segment's last tile
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          double sum = (item.flags & MF_FIRST) ? 0.0 : part[c];
          if (item.flags & MF_SYM) {
#pragma unroll
            for (int tp = 0; tp < NKK; ++tp)
              if (item.row0 == tp) {
#pragma unroll
                for (int rh = 0; rh < 2; ++rh)
#pragma unroll
                  for (int g = 0; g < 4; ++g) sum = fma(acc[rh][c][g], vb[c][8 * tp + 4 * rh + g], sum);
              }
          } else {
#pragma unroll
            for (int rh = 0; rh < 2; ++rh)
#pragma unroll
              for (int g = 0; g < 4; ++g) sum = fma(acc[rh][c][g], acc[rh][c][g], sum);
          }
          part[c] = sum;
        }
        if (item.flags & MF_LAST) {
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            double total = part[c] + xq16(part[c]);
            total += xq32(total);
            const double a0 = aux_lds[wave][c][item.aux][j];
            double kc;
            if (item.type != MI_SOC) {
              kc = a0 + sqrt(fmax(total, 0.0));
            } else {
              // a' x^2 + b' x + c' = 0  (rayen/constraint_module.py:392-396, 339-348), a' < 0
              const double br = aux_lds[wave][c][item.aux + 1][j];
              const double cp = total - a0 * a0;
              const double bp = 2.0 * br - 2.0 * a0 * (double)item.f0d;
              const double disc = bp * bp - 4.0 * (double)item.f1d * cp;
              kc = 0.0;
              if (disc >= 0.0) {
                const double root = sqrt(disc);
                const double inv2a = 0.5 / (double)item.f1d;
                kc = fmax((-bp - root) * inv2a, (-bp + root) * inv2a);
              }
            }
            if (kc > kap[c]) { kap[c] = kc; aseg[c] = item.seg; arow[c] = 0; }
          }
        }
      }
    }

    if (identity) {
      finish_kappa();
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        if (!live[c]) continue;
        double* yrow = y + (s_base + 16 * c + j) * ldy;
#pragma unroll
        for (int st = 0; st < NS; ++st) {
          const int r = 4 * st + q;
          if (r < k) {
            const double o = fma(vb[c][st], scale[c], y0[r]);
            bad |= (o != o);
            yrow[r] = o;
          }
        }
      }
    }

    if (q == 0) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        if (!live[c]) continue;
        const int64_t s = s_base + 16 * c + j;
        if (kappa_out) kappa_out[s] = kap[c];
        if (TRACK) { active_out[2 * s] = aseg[c]; active_out[2 * s + 1] = arow[c]; }
      }
    }
  }
  if (nan_flag && bad) atomicOr(nan_flag, 1);
}

// ---------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------

bool mfma64_eligible(const RayenPack* p) {
  if (p->n > 64) return false;  // v as B operands: n/4 doubles per sample and lane, two samples
  for (const RayenSegment& g : p->segs)
    if (g.type == RAYEN_SEG_LMI) return false;
  TileLayout b(p->n);
  if (layout_tiles(p, b, /*allow_pack=*/true) != RAYEN_OK || b.items.empty()) return false;
  const int64_t padded = (int64_t)b.items.size() * 32;
  return b.useful_rows * 2 >= padded && p->n * 2 >= b.n_pad;
}

void mfma64_free(Mfma64Image* img) {
  if (img == nullptr) return;
  if (img->W) (void)hipFree(img->W);
  if (img->items) (void)hipFree(img->items);
  if (img->packs) (void)hipFree(img->packs);
  if (img->y0) (void)hipFree(img->y0);
  delete img;
}

int mfma64_build(const RayenPack* p, Mfma64Image** out, int64_t* bytes) {
  TileLayout b(p->n);
  const int rc = layout_tiles(p, b, /*allow_pack=*/true);
  if (rc != RAYEN_OK) return rc;
  // SOC constants in full precision (MItem carries them as float for the fp32 kernels)
  for (MItem& it : b.items) {
    it.f0d = it.type == MI_SOC ? p->segs[it.seg].f0 : 0.0;
    it.f1d = it.type == MI_SOC ? p->segs[it.seg].f1 : 0.0;
  }
  b.add_tile({}, p->n);  // spare tile: the prefetch runs one tile past the end
  if (b.packs.empty()) b.packs.push_back(MPack());
  const int nt = b.n_tiles(), ns = b.n_pad / 4;
  // [tile][step pair sg][row half rh][lane l][2]: W[16 rh + (l&15)][4 (2 sg + e) + (l>>4)], e = 0, 1
  std::vector<double> frag((size_t)nt * (ns / 2) * 2 * 64 * 2, 0.0);
  for (int t = 0; t < nt; ++t)
    for (int sg = 0; sg < ns / 2; ++sg)
      for (int rh = 0; rh < 2; ++rh)
        for (int l = 0; l < 64; ++l)
          for (int e = 0; e < 2; ++e)
            frag[((((size_t)t * (ns / 2) + sg) * 2 + rh) * 64 + l) * 2 + e] =
                b.raw[((size_t)t * 32 + 16 * rh + (l & 15)) * b.n_pad + 4 * (2 * sg + e) + (l >> 4)];

  Mfma64Image* img = new Mfma64Image();
  img->nkk = b.n_pad / 32;
  img->identity = p->out_identity;
  img->n_items = (int)b.items.size();
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, p->device) == hipSuccess && prop.multiProcessorCount > 0)
      img->n_simd = prop.multiProcessorCount * 4;
  }
  const int k_tiles = (p->k + 31) / 32;
  std::vector<double> y0((size_t)k_tiles * 32 + 32, 0.0);
  for (int i = 0; i < p->k; ++i) y0[i] = p->y0[i];
  const bool ok =
      hipMalloc(&img->W, frag.size() * sizeof(double)) == hipSuccess &&
      hipMemcpy(img->W, frag.data(), frag.size() * sizeof(double), hipMemcpyHostToDevice) == hipSuccess &&
      hipMalloc(&img->y0, y0.size() * sizeof(double)) == hipSuccess &&
      hipMemcpy(img->y0, y0.data(), y0.size() * sizeof(double), hipMemcpyHostToDevice) == hipSuccess &&
      hipMalloc(&img->items, b.items.size() * sizeof(MItem)) == hipSuccess &&
      hipMemcpy(img->items, b.items.data(), b.items.size() * sizeof(MItem), hipMemcpyHostToDevice) == hipSuccess &&
      hipMalloc(&img->packs, b.packs.size() * sizeof(MPack)) == hipSuccess &&
      hipMemcpy(img->packs, b.packs.data(), b.packs.size() * sizeof(MPack), hipMemcpyHostToDevice) == hipSuccess;
  if (!ok) { mfma64_free(img); return RAYEN_E_ALLOC; }
  img->bytes = (int64_t)(frag.size() * sizeof(double) + y0.size() * sizeof(double) + b.items.size() * sizeof(MItem));
  *bytes = img->bytes;
  *out = img;
  return RAYEN_OK;
}

template <int NKK>
static int launch64(const RayenPack* p, const Mfma64Image* img, const double* v, int64_t B, int64_t ldv,
                    double* y, int64_t ldy, double* kappa, int32_t* active, int32_t* nan_flag,
                    int old_mode, hipStream_t stream) {
  // persistent, balanced: 2 waves per SIMD, every wave the same number of 32-sample groups
  const int64_t n_groups = (B + 31) / 32;
  const int64_t slots = (int64_t)launch_simds(img->n_simd) * 2;
  const int64_t rounds = (n_groups + slots - 1) / slots;
  const int64_t waves = (n_groups + rounds - 1) / rounds;
  const int64_t grid = (waves + k64Waves - 1) / k64Waves;
  if (active != nullptr) {
    hipLaunchKernelGGL((mfma64_fwd_kernel<NKK, true>), dim3((unsigned)grid), dim3(k64Waves * 64), 0, stream,
                       img->W, img->items, img->n_items, img->packs, img->y0, img->identity, p->k, p->n, v, B, ldv, y, ldy,
                       kappa, active, nan_flag, old_mode);
  } else {
    hipLaunchKernelGGL((mfma64_fwd_kernel<NKK, false>), dim3((unsigned)grid), dim3(k64Waves * 64), 0, stream,
                       img->W, img->items, img->n_items, img->packs, img->y0, img->identity, p->k, p->n, v, B, ldv, y, ldy,
                       kappa, active, nan_flag, old_mode);
  }
  return hipGetLastError() == hipSuccess ? RAYEN_OK : RAYEN_E_LAUNCH;
}

int mfma64_forward(const RayenPack* p, const Mfma64Image* img, const double* v, int64_t B, int64_t ldv,
                   double* y, int64_t ldy, double* kappa, int32_t* active, int32_t* nan_flag,
                   int old_mode, hipStream_t stream) {
  if (B == 0) return RAYEN_OK;
  if (img->nkk == 1) return launch64<1>(p, img, v, B, ldv, y, ldy, kappa, active, nan_flag, old_mode, stream);
  if (img->nkk == 2) return launch64<2>(p, img, v, B, ldv, y, ldy, kappa, active, nan_flag, old_mode, stream);
  return RAYEN_E_UNSUPPORTED;
}

}  // namespace rayen
