// Wide sets: the epilogue of the projection on PRODUCTS that a library GEMM has formed
// (rayen/constraint_module.py:351-458 + 468-474 from T = v W_ext').
//
// The matrix-core kernels of this library keep a sample tile's direction in registers and walk the rows of W past it:
// n <= 64 (f16 pairs / fp64) or 128 (exact fp32).  Beyond that the only kernel was the lane-per-sample one, which
// re-reads every row of W per lane through the scalar cache -- correct at any n, and 10-100 x off the pace at the sizes of the
// reference's own sweep (examples/scripts/time_analysis.py:57-192: k to 10^4, thousands of rows, B = 2000).  For
// those shapes `T = V W'` is a PLAIN GEMM, and a plain GEMM belongs to the vendor library (hipBLASLt / rocBLAS, called
// by the host side through torch.mm on the caller's stream: fp32 / fp64 MFMA, tuned tilings -- nothing a hand-written
// kernel adds).  What is left of computeKappa is per sample and bandwidth-bound on T, and it is this file:
//
//   one wave per sample reads its row of T once (coalesced), and per segment of the pack reduces
//     LIN       max_i T_i (+ arg-max)                                   CM:353
//     QUAD_SYM  T_aux + sqrt(max(sum_j T_j v_j, 0))                     CM:374    (rows of the segment hold G v)
//     QUAD_FAC  T_aux + sqrt(sum_j T_j^2)                                          (rows hold U v, U'U = G)
//     SOC       larger root of a'x^2 + b'x + c', c' = sum_j T_j^2 - T_aux^2, b' = 2 T_aux+1 - 2 tau T_aux   CM:383-399
//   to kappa = relu(max), then writes y = y0 + (NA_E v) / max(1, kappa) -- NA_E v being the LAST k columns of T when
//   the set has equality constraints (W_ext = [W ; NA_E]), v itself otherwise.
// LMI segments are not taken here (their matrices need the per-sample eigen-solvers of rayen_lmi_*.h).
#include "rayen_internal.h"

#include <vector>

namespace rayen {

struct WSeg {   // 32 bytes
  int32_t type, row0, nrows, aux_row, seg, pad0;
  float f0, f1;
};
struct WSeg64 {
  int32_t type, row0, nrows, aux_row, seg, pad0;
  double f0, f1;
};

struct WideImage {
  void* segs32 = nullptr;   // WSeg[n_seg]
  void* segs64 = nullptr;   // WSeg64[n_seg]
  float* y0_32 = nullptr;
  double* y0_64 = nullptr;
  int n_seg = 0;
  int n_simd = 1024;
  int64_t bytes = 0;
};

namespace {

template <typename T> struct WSegOf;
template <> struct WSegOf<float> { typedef WSeg type; };
template <> struct WSegOf<double> { typedef WSeg64 type; };

template <typename T>
__device__ __forceinline__ T wave_sum(T x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
  return x;
}

template <typename T>
__global__ __launch_bounds__(256) void wide_epilogue_kernel(
    const typename WSegOf<T>::type* __restrict__ segs, int n_seg, const T* __restrict__ y0, int k, int n, int n_rows,
    int identity, const T* __restrict__ Tm, int64_t ldt, const T* __restrict__ v, int64_t B, int64_t ldv,
    T* __restrict__ y, int64_t ldy, T* __restrict__ kappa_out, int32_t* __restrict__ active_out,
    int32_t* __restrict__ nan_flag) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * 4;
  bool bad = false;
  for (int64_t s = wave; s < B; s += n_waves) {
    const T* __restrict__ tr = Tm + s * ldt;
    const T* __restrict__ vr = v + s * ldv;
    T kap = T(0);
    int aseg = -1, arow = 0;
    for (int g = 0; g < n_seg; ++g) {
      const auto sg = segs[g];
      if (sg.type == RAYEN_SEG_LIN) {
        T m = T(0);
        int at = -1;
        for (int j = lane; j < sg.nrows; j += 64) {
          const T t = tr[sg.row0 + j];
          if (t > m || (t != t)) { m = t; at = j; }
        }
        // wave arg-max; ties go to the lower row (what a serial scan keeps)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          const T om = __shfl_xor(m, o);
          const int oat = __shfl_xor(at, o);
          if (om > m || (om == m && oat >= 0 && (at < 0 || oat < at)) || (om != om)) { m = om; at = oat; }
        }
        if (m > kap || (m != m)) { kap = m; aseg = sg.seg; arow = sg.row0 + at; }
      } else if (sg.type == RAYEN_SEG_QUAD_SYM || sg.type == RAYEN_SEG_QUAD_FAC || sg.type == RAYEN_SEG_SOC) {
        T acc = T(0);
        if (sg.type == RAYEN_SEG_QUAD_SYM) {
          for (int j = lane; j < sg.nrows; j += 64) acc = fma(tr[sg.row0 + j], vr[j], acc);
        } else {
          for (int j = lane; j < sg.nrows; j += 64) {
            const T t = tr[sg.row0 + j];
            acc = fma(t, t, acc);
          }
        }
        acc = wave_sum(acc);
        T cand;
        if (sg.type != RAYEN_SEG_SOC) {
          cand = tr[sg.aux_row] + sqrt(fmax(acc, T(0)));
        } else {
          // a' x^2 + b' x + c' = 0 with a' = f1 < 0, tau = f0 (constraint_module.py:392-396); a ray that never meets
          // the cone contributes 0 (documented deviation from CM:342)
          const T cr = tr[sg.aux_row], br = tr[sg.aux_row + 1];
          const T cp = acc - cr * cr;
          const T bp = T(2) * br - T(2) * cr * (T)sg.f0;
          const T disc = bp * bp - T(4) * (T)sg.f1 * cp;
          cand = T(0);
          if (disc >= T(0)) {
            const T root = sqrt(disc), inv2a = T(0.5) / (T)sg.f1;
            cand = fmax((-bp - root) * inv2a, (-bp + root) * inv2a);
          } else if (disc != disc) {
            cand = disc;
          }
        }
        if (cand > kap || (cand != cand)) { kap = cand; aseg = sg.seg; arow = 0; }
      }
    }
    const T scale = T(1) / fmax(T(1), kap);     // (fmax drops a NaN kappa; the outputs below carry it through `kap` itself)
    const T carry = (kap != kap) ? kap : T(0);
    if (lane == 0) {
      if (kappa_out) kappa_out[s] = kap;
      if (active_out) { active_out[2 * s] = aseg; active_out[2 * s + 1] = arow; }
    }
    if (y != nullptr) {
      T* __restrict__ yr = y + s * ldy;
      const T* __restrict__ src = identity ? vr : tr + n_rows;
      for (int i = lane; i < k; i += 64) {
        const T o = fma(src[i], scale, y0[i]) + carry;
        bad |= (o != o);
        yr[i] = o;
      }
    }
  }
  if (nan_flag && bad) atomicOr(nan_flag, 1);
}

// Backward of the wide route.  grad_v = s N'g - [kappa > 1] s^2 (g . N v) grad kappa(v), and grad kappa of the ACTIVE
// constraint is a combination of rows of W (the arg-max row; phi + sum_j (v_j / sqrt(v'Gv)) G_j; phi + sum_j (w_j / |w|) U_j;
// the implicit derivative of the cone's root in c, M'beta and the rows of M) -- so is N'g in the rows of NA_E.  This
// kernel writes those COEFFICIENTS, one row of C per sample (zero outside the active segment), and the host finishes
// with ONE vendor GEMM: grad_v = C W_ext (+ s g when NA_E = I, written to `gs`).
template <typename T>
__global__ __launch_bounds__(256) void wide_bwd_coeff_kernel(
    const typename WSegOf<T>::type* __restrict__ segs, int n_seg, int k, int n, int n_rows, int identity,
    const T* __restrict__ Tm, int64_t ldt, const T* __restrict__ v, int64_t B, int64_t ldv, const T* __restrict__ kappa,
    const int32_t* __restrict__ active, const T* __restrict__ gy, int64_t ldg, T* __restrict__ C, int64_t ldc,
    T* __restrict__ gs) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * 4;
  const int rows_ext = n_rows + (identity ? 0 : k);
  for (int64_t s = wave; s < B; s += n_waves) {
    const T* __restrict__ tr = Tm + s * ldt;
    const T* __restrict__ vr = v + s * ldv;
    const T* __restrict__ gr = gy + s * ldg;
    T* __restrict__ cr_ = C + s * ldc;
    const T kap = kappa[s];
    const T sc = T(1) / fmax(T(1), kap);
    const T* __restrict__ nv = identity ? vr : tr + n_rows;
    T tv = T(0);
    for (int i = lane; i < k; i += 64) tv = fma(gr[i], nv[i], tv);
    tv = wave_sum(tv);
    const bool clipped = kap > T(1);
    const T coef = clipped ? -sc * sc * tv : T(0);
    for (int j = lane; j < n_rows; j += 64) cr_[j] = T(0);
    if (identity) {
      for (int i = lane; i < k; i += 64) gs[s * k + i] = sc * gr[i];
    } else {
      for (int i = lane; i < k; i += 64) cr_[n_rows + i] = sc * gr[i];
    }
    (void)rows_ext;
    const int a = active[2 * s];
    if (!clipped || a < 0 || a >= n_seg) continue;
    const auto sg = segs[a];
    if (sg.type == RAYEN_SEG_LIN) {
      const int row = active[2 * s + 1];
      if (lane == 0 && row >= sg.row0 && row < sg.row0 + sg.nrows) cr_[row] = coef;
    } else if (sg.type == RAYEN_SEG_QUAD_SYM || sg.type == RAYEN_SEG_QUAD_FAC) {
      const bool sym = sg.type == RAYEN_SEG_QUAD_SYM;
      T acc = T(0);
      for (int j = lane; j < sg.nrows; j += 64) {
        const T t = tr[sg.row0 + j];
        acc = sym ? fma(t, vr[j], acc) : fma(t, t, acc);
      }
      acc = wave_sum(acc);
      const T rad = sqrt(fmax(acc, T(0)));
      const T inv = rad > T(0) ? coef / rad : T(0);
      for (int j = lane; j < sg.nrows; j += 64) cr_[sg.row0 + j] = inv * (sym ? vr[j] : tr[sg.row0 + j]);
      if (lane == 0) cr_[sg.aux_row] = coef;
    } else if (sg.type == RAYEN_SEG_SOC) {
      const T cr = tr[sg.aux_row], br = tr[sg.aux_row + 1];
      const T bp = T(2) * br - T(2) * cr * (T)sg.f0;
      const T den = T(2) * (T)sg.f1 * kap + bp;
      const T q = den != T(0) ? coef / den : T(0);
      for (int j = lane; j < sg.nrows; j += 64) cr_[sg.row0 + j] = T(-2) * q * tr[sg.row0 + j];
      if (lane == 0) {
        cr_[sg.aux_row] = q * (T(2) * (T)sg.f0 * kap + T(2) * cr);
        cr_[sg.aux_row + 1] = T(-2) * q * kap;
      }
    }
  }
}

template <typename V>
bool upload_vec(const std::vector<V>& host, void** dev, int64_t* bytes) {
  if (host.empty()) { *dev = nullptr; return true; }
  if (hipMalloc(dev, host.size() * sizeof(V)) != hipSuccess) return false;
  if (hipMemcpy(*dev, host.data(), host.size() * sizeof(V), hipMemcpyHostToDevice) != hipSuccess) return false;
  *bytes += (int64_t)(host.size() * sizeof(V));
  return true;
}

}  // namespace

void wide_free(WideImage* img) {
  if (img == nullptr) return;
  if (img->segs32) (void)hipFree(img->segs32);
  if (img->segs64) (void)hipFree(img->segs64);
  if (img->y0_32) (void)hipFree(img->y0_32);
  if (img->y0_64) (void)hipFree(img->y0_64);
  delete img;
}

// Every pack without an LMI gets the (tiny) tables: segment records and y0 in both precisions.
int wide_build(const RayenPack* p, WideImage** out, int64_t* bytes) {
  *out = nullptr;
  for (const RayenSegment& g : p->segs)
    if (g.type == RAYEN_SEG_LMI) return RAYEN_OK;
  WideImage* img = new WideImage();
  {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      img->n_simd = prop.multiProcessorCount * 4;
  }
  std::vector<WSeg> s32;
  std::vector<WSeg64> s64;
  for (size_t i = 0; i < p->segs.size(); ++i) {
    const RayenSegment& g = p->segs[i];
    s32.push_back(WSeg{g.type, g.row0, g.nrows, g.aux_row, (int32_t)i, 0, (float)g.f0, (float)g.f1});
    s64.push_back(WSeg64{g.type, g.row0, g.nrows, g.aux_row, (int32_t)i, 0, g.f0, g.f1});
  }
  std::vector<float> y32(p->y0.begin(), p->y0.end());
  img->n_seg = (int)s32.size();
  void* d = nullptr;
  bool ok = upload_vec(s32, &img->segs32, &img->bytes) && upload_vec(s64, &img->segs64, &img->bytes);
  ok = ok && upload_vec(y32, &d, &img->bytes);
  img->y0_32 = static_cast<float*>(d);
  d = nullptr;
  ok = ok && upload_vec(p->y0, &d, &img->bytes);
  img->y0_64 = static_cast<double*>(d);
  if (!ok) { wide_free(img); return RAYEN_E_ALLOC; }
  *bytes = img->bytes;
  *out = img;
  return RAYEN_OK;
}

template <typename T>
int wide_epilogue(const RayenPack* p, const WideImage* img, const T* Tm, int64_t ldt, const T* v, int64_t B, int64_t ldv,
                  T* y, int64_t ldy, T* kappa, int32_t* active, int32_t* nan_flag, hipStream_t stream) {
  if (img == nullptr) return RAYEN_E_UNSUPPORTED;
  if (B == 0) return RAYEN_OK;
  const int64_t want = (B + 3) / 4, cap = (int64_t)launch_simds(img->n_simd) * 2;   // a few waves per SIMD, grid-stride beyond
  const unsigned grid = (unsigned)(want < cap ? want : cap);
  typedef typename WSegOf<T>::type S;
  const S* segs = static_cast<const S*>(sizeof(T) == 4 ? img->segs32 : img->segs64);
  const T* y0 = reinterpret_cast<const T*>(sizeof(T) == 4 ? (const void*)img->y0_32 : (const void*)img->y0_64);
  hipLaunchKernelGGL((wide_epilogue_kernel<T>), dim3(grid), dim3(256), 0, stream, segs, img->n_seg, y0, p->k, p->n,
                     p->n_rows, p->out_identity, Tm, ldt, v, B, ldv, y, ldy, kappa, active, nan_flag);
  return hipGetLastError() == hipSuccess ? RAYEN_OK : RAYEN_E_LAUNCH;
}

template <typename T>
int wide_bwd_coefficients(const RayenPack* p, const WideImage* img, const T* Tm, int64_t ldt, const T* v, int64_t B,
                          int64_t ldv, const T* kappa, const int32_t* active, const T* gy, int64_t ldg, T* C, int64_t ldc,
                          T* gs, hipStream_t stream) {
  if (img == nullptr) return RAYEN_E_UNSUPPORTED;
  if (B == 0) return RAYEN_OK;
  const int64_t want = (B + 3) / 4, cap = (int64_t)launch_simds(img->n_simd) * 2;
  const unsigned grid = (unsigned)(want < cap ? want : cap);
  typedef typename WSegOf<T>::type S;
  const S* segs = static_cast<const S*>(sizeof(T) == 4 ? img->segs32 : img->segs64);
  hipLaunchKernelGGL((wide_bwd_coeff_kernel<T>), dim3(grid), dim3(256), 0, stream, segs, img->n_seg, p->k, p->n, p->n_rows,
                     p->out_identity, Tm, ldt, v, B, ldv, kappa, active, gy, ldg, C, ldc, gs);
  return hipGetLastError() == hipSuccess ? RAYEN_OK : RAYEN_E_LAUNCH;
}

template int wide_bwd_coefficients<float>(const RayenPack*, const WideImage*, const float*, int64_t, const float*, int64_t,
                                          int64_t, const float*, const int32_t*, const float*, int64_t, float*, int64_t,
                                          float*, hipStream_t);
template int wide_bwd_coefficients<double>(const RayenPack*, const WideImage*, const double*, int64_t, const double*,
                                           int64_t, int64_t, const double*, const int32_t*, const double*, int64_t,
                                           double*, int64_t, double*, hipStream_t);

template int wide_epilogue<float>(const RayenPack*, const WideImage*, const float*, int64_t, const float*, int64_t,
                                  int64_t, float*, int64_t, float*, int32_t*, int32_t*, hipStream_t);
template int wide_epilogue<double>(const RayenPack*, const WideImage*, const double*, int64_t, const double*, int64_t,
                                   int64_t, double*, int64_t, double*, int32_t*, int32_t*, hipStream_t);

}  // namespace rayen
