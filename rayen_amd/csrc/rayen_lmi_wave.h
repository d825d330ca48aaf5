// One WAVE per sample for sets = [linear rows] + one LMI whose matrix is too large for the four-lanes-per-sample kernel
// (rayen_lmi_quad.h: 32 x 32 in fp32, 24 x 24 in fp64) and for the lane-per-sample kernels (~30 x 30): the reference's
// own LMI sweep goes to 100 x 100 and beyond (examples/scripts/time_analysis.py:159-160; rayen/constraint_module.py:401-449
// handles any r).  The symmetric matrix S(v) = sum_a v_a G_a (G_a = -L' F_a L contracted with NA_E, the packed rows of W)
// lives in the wave's LDS; lambda_max comes from
//   * a Householder tridiagonalisation in which every lane owns rows i = lane, lane + 64, ... (row-wise dot products and
//     rank-2 updates; the leading dimension is odd, so the lanes of a wave hit 64 different banks),
//   * a Sturm-count MULTISECTION: each of the 64 lanes evaluates the count at its own shift, one ballot picks the
//     sub-interval -- 65x per round, six rounds for fp32, ten for fp64.
// The backward repeats the reduction keeping the reflectors, gets the eigenvector of the tridiagonal matrix by inverse
// iteration on (lambda + shift) I - T (positive definite: LDL' without pivoting, as in rayen_lmi_quad.h), maps it back
// through the reflectors and contracts x x' with every generator (what autograd gives for eigvalsh + max,
// rayen/constraint_module.py:424-425).  One 64-thread workgroup per sample; LDS per workgroup = r (r | 1) + O(r + n + k)
// elements: r <= ~190 in fp32, ~135 in fp64.
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <vector>

#include "rayen_internal.h"

namespace rayen {

struct LmiWaveImage {
  void* gt = nullptr;       // [n][Pp]  generators, packed lower triangle, transposed (coalesced over the entries)
  void* dt = nullptr;       // [n][Mp]  linear rows, transposed
  void* nat = nullptr;      // [n][Kp]  NA_E transposed (forward write-out), null when NA_E = I
  void* nrm = nullptr;      // [k][n]   NA_E row-major (backward pull-back), null when NA_E = I
  void* y0 = nullptr;       // [k]
  int32_t* lin_id = nullptr;   // [m][2]  (segment, W row) of every linear row
  int32_t* rho_of = nullptr;   // [n_rows] index among the linear rows of a W row (-1: not a linear row)
  int r = 0, n = 0, k = 0, m = 0, P = 0, Pp = 0, Mp = 0, Kp = 0, identity = 0, lmi_seg = 0;
  int lmi_row0 = 0, n_rows = 0;   // the LMI's first row of W, the rows of W (the products route of rayen_lmi_block.h)
  int64_t bytes = 0;
};

namespace lw {

template <typename T>
__device__ __forceinline__ T wsum(T x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
  return x;
}

template <typename T> struct Eps;
template <> struct Eps<float> {
  static constexpr float tiny = 1e-30f;
  static constexpr int rounds = 6;
  static constexpr float shift = 2e-4f;
};
template <> struct Eps<double> {
  static constexpr double tiny = 1e-290;
  static constexpr int rounds = 10;
  static constexpr double shift = 1e-9;
};

// LDS of one sample (units of T): A[r][LD] | dd[r] | ee[r] | tau[r] | vv[r] | ww[r] | zz[r] | vs[n] | gs[k] | ts[n]
__host__ __device__ inline int ld_of(int r) { return r | 1; }
__host__ __device__ inline size_t lds_elems(int r, int n, int k) {
  return (size_t)r * ld_of(r) + 6 * (size_t)r + 2 * (size_t)n + (size_t)k + 8;
}

// S(v) into the wave's LDS (full symmetric storage)
template <typename T>
__device__ __forceinline__ void form_S(T* A, const int LD, const T* __restrict__ gt, const T* vs, const int n, const int P,
                                       const int Pp, const int lane) {
  for (int idx = lane; idx < P; idx += 64) {
    // (four partial sums: the loads of consecutive generators are independent, one accumulator would serialise them)
    T p0 = T(0), p1 = T(0), p2 = T(0), p3 = T(0);
    const T* col = gt + idx;
    int a = 0;
    for (; a + 3 < n; a += 4) {
      p0 = fma(vs[a + 0], col[(size_t)(a + 0) * Pp], p0);
      p1 = fma(vs[a + 1], col[(size_t)(a + 1) * Pp], p1);
      p2 = fma(vs[a + 2], col[(size_t)(a + 2) * Pp], p2);
      p3 = fma(vs[a + 3], col[(size_t)(a + 3) * Pp], p3);
    }
    for (; a < n; ++a) p0 = fma(vs[a], col[(size_t)a * Pp], p0);
    const T acc = (p0 + p1) + (p2 + p3);
    int i = (int)((sqrtf(8.f * (float)idx + 1.f) - 1.f) * 0.5f);
    while ((i + 1) * (i + 2) / 2 <= idx) ++i;
    while (i * (i + 1) / 2 > idx) --i;
    const int j = idx - i * (i + 1) / 2;
    A[i * LD + j] = acc;
    A[j * LD + i] = acc;
  }
  __syncthreads();
}

// Householder reduction to tridiagonal form: dd (diagonal), ee (signed sub-diagonal, ee[r - 1] = 0); the reflector of
// column c stays in A[c + 1 ..][c] with its scale 2 / v'v in tau[c] (KEEP: the backward maps the eigenvector back)
template <typename T>
__device__ __forceinline__ void tridiagonalise(T* A, const int LD, const int r, T* dd, T* ee, T* tau, T* vv, T* ww,
                                               const int lane) {
  for (int kc = 0; kc + 2 < r; ++kc) {
    const int i0 = kc + 1;
    const T x0 = A[i0 * LD + kc];
    T sigma = T(0);
    for (int i = i0 + lane; i < r; i += 64) {
      const T xi = A[i * LD + kc];
      sigma = fma(xi, xi, sigma);
    }
    sigma = wsum(sigma);
    const T below = sigma - x0 * x0;   // what the reflector has to annihilate
    if (!(below > Eps<T>::tiny * Eps<T>::tiny)) {   // nothing to do: H = I
      if (lane == 0) { dd[kc] = A[kc * LD + kc]; ee[kc] = x0; tau[kc] = T(0); }
      for (int i = i0 + lane; i < r; i += 64) A[i * LD + kc] = T(0);
      __syncthreads();
      continue;
    }
    const T alpha = (x0 >= T(0) ? T(-1) : T(1)) * sqrt(sigma);
    const T taup = T(1) / (sigma - x0 * alpha);
    for (int i = i0 + lane; i < r; i += 64) {
      T vi = A[i * LD + kc];
      if (i == i0) vi -= alpha;
      vv[i] = vi;
      A[i * LD + kc] = vi;
    }
    __syncthreads();
    T pv = T(0);
    for (int i = i0 + lane; i < r; i += 64) {
      const T* row = A + i * LD;
      T q0 = T(0), q1 = T(0), q2 = T(0), q3 = T(0);
      int j = i0;
      for (; j + 3 < r; j += 4) {
        q0 = fma(row[j + 0], vv[j + 0], q0);
        q1 = fma(row[j + 1], vv[j + 1], q1);
        q2 = fma(row[j + 2], vv[j + 2], q2);
        q3 = fma(row[j + 3], vv[j + 3], q3);
      }
      for (; j < r; ++j) q0 = fma(row[j], vv[j], q0);
      const T acc = (q0 + q1) + (q2 + q3);
      const T pi = taup * acc;
      ww[i] = pi;
      pv = fma(pi, vv[i], pv);
    }
    const T K = T(0.5) * taup * wsum(pv);
    for (int i = i0 + lane; i < r; i += 64) ww[i] = fma(-K, vv[i], ww[i]);
    __syncthreads();
    for (int i = i0 + lane; i < r; i += 64) {
      T* row = A + i * LD;
      const T vi = vv[i], wi = ww[i];
      for (int j = i0; j < r; ++j) row[j] = row[j] - (vi * ww[j] + wi * vv[j]);
    }
    if (lane == 0) { dd[kc] = A[kc * LD + kc]; ee[kc] = alpha; tau[kc] = taup; }
    __syncthreads();
  }
  if (lane == 0) {
    if (r >= 2) {
      dd[r - 2] = A[(r - 2) * LD + (r - 2)];
      ee[r - 2] = A[(r - 1) * LD + (r - 2)];
      tau[r - 2] = T(0);
    }
    dd[r - 1] = A[(r - 1) * LD + (r - 1)];
    ee[r - 1] = T(0);
    tau[r - 1] = T(0);
  }
  __syncthreads();
}

// largest eigenvalue of the tridiagonal (dd, ee): Sturm counts at 64 shifts per round
template <typename T>
__device__ __forceinline__ T lambda_max_tridiagonal(const T* dd, const T* ee, const int r, const int lane) {
  T lo = dd[0], hi = dd[0], scale = T(0);
  for (int i = 0; i < r; ++i) {
    const T off = (i > 0 ? fabs(ee[i - 1]) : T(0)) + (i + 1 < r ? fabs(ee[i]) : T(0));
    lo = fmin(lo, dd[i] - off);
    hi = fmax(hi, dd[i] + off);
    scale = fmax(scale, fabs(dd[i]) + off);
  }
  const T pad = scale * (sizeof(T) == 4 ? T(1e-6) : T(1e-14)) + Eps<T>::tiny;
  lo -= pad;
  hi += pad;
  const T floor_q = fmax(scale * (sizeof(T) == 4 ? T(1e-30) : T(1e-200)), Eps<T>::tiny);
  for (int round = 0; round < Eps<T>::rounds; ++round) {
    const T step = (hi - lo) * (T(1) / T(65));
    const T sigma = lo + step * (T)(lane + 1);
    // number of eigenvalues below sigma = negative pivots of T - sigma I
    int below = 0;
    T q = dd[0] - sigma;
    below += q < T(0);
    for (int i = 1; i < r; ++i) {
      if (fabs(q) < floor_q) q = q < T(0) ? -floor_q : floor_q;
      const T e = ee[i - 1];
      q = dd[i] - sigma - e * e / q;
      below += q < T(0);
    }
    const unsigned long long above = __ballot(below >= r);   // sigma beyond the largest eigenvalue
    const int first = above ? __builtin_ctzll(above) : 64;
    const T new_lo = first == 0 ? lo : lo + step * (T)first;
    const T new_hi = first == 64 ? hi : lo + step * (T)(first + 1);
    lo = new_lo;
    hi = new_hi;
  }
  return T(0.5) * (lo + hi);
}

// kappa of the linear rows: (value, linear-row index) of the largest D_i . v (value 0, index -1 when none is positive)
template <typename T>
__device__ __forceinline__ void linear_rows(const T* __restrict__ dt, const T* vs, const int n, const int m, const int Mp,
                                            const int lane, T& best, int& who) {
  best = T(0);
  who = -1;
  for (int i = lane; i < m; i += 64) {
    T acc = T(0);
    const T* col = dt + i;
    for (int a = 0; a < n; ++a) acc = fma(vs[a], col[(size_t)a * Mp], acc);
    if (acc > best) { best = acc; who = i; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const T ob = __shfl_xor(best, o);
    const int ow = __shfl_xor(who, o);
    if (ob > best || (ob == best && ow >= 0 && (who < 0 || ow < who))) { best = ob; who = ow; }
  }
}

template <typename T>
__global__ __launch_bounds__(64) void lmi_wave_kernel(
    const T* __restrict__ gt, const T* __restrict__ dt, const T* __restrict__ nat, const T* __restrict__ y0,
    const int32_t* __restrict__ lin_id, int r, int n, int k, int m, int P, int Pp, int Mp, int Kp, int identity,
    int lmi_seg, const T* __restrict__ v, int64_t B, int64_t ldv, T* __restrict__ y, int64_t ldy,
    T* __restrict__ kappa_out, int32_t* __restrict__ active_out, int32_t* __restrict__ nan_flag) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lw_smem[];
  T* A = reinterpret_cast<T*>(lw_smem);
  const int LD = ld_of(r);
  T* dd = A + (size_t)r * LD;
  T* ee = dd + r;
  T* tau = ee + r;
  T* vv = tau + r;
  T* ww = vv + r;
  T* vs = ww + 2 * r;
  const int lane = threadIdx.x;
  const int64_t b = blockIdx.x;
  if (b >= B) return;
  for (int a = lane; a < n; a += 64) vs[a] = v[b * ldv + a];
  __syncthreads();

  T kap;
  int who;
  linear_rows<T>(dt, vs, n, m, Mp, lane, kap, who);
  int aseg = who >= 0 ? lin_id[2 * who] : -1, arow = who >= 0 ? lin_id[2 * who + 1] : 0;

  form_S<T>(A, LD, gt, vs, n, P, Pp, lane);
  tridiagonalise<T>(A, LD, r, dd, ee, tau, vv, ww, lane);
  const T lam = lambda_max_tridiagonal<T>(dd, ee, r, lane);
  if (lam > kap) { kap = lam; aseg = lmi_seg; arow = 0; }

  const T scale = T(1) / fmax(T(1), kap);
  if (lane == 0) {
    if (kappa_out) kappa_out[b] = kap;
    if (active_out) { active_out[2 * b] = aseg; active_out[2 * b + 1] = arow; }
  }
  bool bad = false;
  T* yrow = y + b * ldy;
  for (int i = lane; i < k; i += 64) {
    T val;
    if (identity) {
      val = fma(vs[i], scale, y0[i]);
    } else {
      T acc = T(0);
      const T* col = nat + i;
      for (int a = 0; a < n; ++a) acc = fma(vs[a], col[(size_t)a * Kp], acc);
      val = fma(acc, scale, y0[i]);
    }
    bad |= (val != val);
    yrow[i] = val;
  }
  if (nan_flag && bad) atomicOr(nan_flag, 1);
}

//   grad_v = s t - [kappa > 1] s^2 (t . v) grad kappa(v),   t = NA_E' g,   s = 1 / max(1, kappa)
template <typename T>
__global__ __launch_bounds__(64) void lmi_wave_bwd_kernel(
    const T* __restrict__ gt, const T* __restrict__ dt, const T* __restrict__ nrm, const int32_t* __restrict__ rho_of,
    int r, int n, int k, int P, int Pp, int Mp, int identity, int lmi_seg, const T* __restrict__ v, int64_t B,
    int64_t ldv, const T* __restrict__ kappa, const int32_t* __restrict__ active, const T* __restrict__ gy, int64_t ldg,
    T* __restrict__ gv, int64_t ldgv) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lw_smem[];
  T* A = reinterpret_cast<T*>(lw_smem);
  const int LD = ld_of(r);
  T* dd = A + (size_t)r * LD;
  T* ee = dd + r;
  T* tau = ee + r;
  T* vv = tau + r;
  T* ww = vv + r;
  T* zz = ww + r;
  T* vs = zz + r;
  T* gs = vs + n;
  T* ts = gs + k;
  const int lane = threadIdx.x;
  const int64_t b = blockIdx.x;
  if (b >= B) return;
  for (int a = lane; a < n; a += 64) vs[a] = v[b * ldv + a];
  for (int i = lane; i < k; i += 64) gs[i] = gy[b * ldg + i];
  __syncthreads();
  T tv = T(0);
  for (int a = lane; a < n; a += 64) {
    T acc;
    if (identity) {
      acc = gs[a];
    } else {
      acc = T(0);
      for (int i = 0; i < k; ++i) acc = fma(nrm[(size_t)i * n + a], gs[i], acc);
    }
    ts[a] = acc;
    tv = fma(acc, vs[a], tv);
  }
  tv = wsum(tv);
  const T kap = kappa[b];
  const int aseg = active[2 * b], arow = active[2 * b + 1];
  const bool clipped = kap > T(1) && aseg >= 0;
  const T sc = T(1) / fmax(T(1), kap);
  const T coef = clipped ? sc * sc * tv : T(0);
  __syncthreads();

  if (clipped && aseg == lmi_seg) {
    form_S<T>(A, LD, gt, vs, n, P, Pp, lane);
    tridiagonalise<T>(A, LD, r, dd, ee, tau, vv, ww, lane);
    // ---- eigenvector of the tridiagonal matrix: inverse iteration on M = (lam + shift) I - T = L D L' (every lane runs
    // the O(r) recurrences redundantly; lane 0 writes)
    T scale = fabs(kap);
    for (int i = 0; i < r; ++i) scale = fmax(scale, fabs(dd[i]));
    const T shift = Eps<T>::shift * fmax(scale, Eps<T>::tiny);
    // ww: D of the factorisation, vv: the sub-diagonal of L (vv[i] couples rows i - 1 and i)
    {
      T dprev = fmax(kap + shift - dd[0], shift * T(1e-3));
      if (lane == 0) { ww[0] = dprev; vv[0] = T(0); }
      for (int i = 1; i < r; ++i) {
        const T li = ee[i - 1] / dprev;              // M's off-diagonal is -ee: l = -ee / D, kept with the sign folded
        const T di = fmax(kap + shift - dd[i] - li * ee[i - 1], shift * T(1e-3));
        if (lane == 0) { vv[i] = -li; ww[i] = di; }
        dprev = di;
      }
    }
    for (int i = lane; i < r; i += 64) zz[i] = T(1) + T(0.01) * (T)i;   // not orthogonal to anything special
    __syncthreads();
    for (int it = 0; it < 3; ++it) {
      if (lane == 0) {
        for (int i = 1; i < r; ++i) zz[i] = fma(-vv[i], zz[i - 1], zz[i]);          // L y = b
        zz[r - 1] = zz[r - 1] / ww[r - 1];
        T nrm2 = zz[r - 1] * zz[r - 1];
        for (int i = r - 2; i >= 0; --i) {                                            // D L' z = y
          zz[i] = fma(-vv[i + 1], zz[i + 1], zz[i] / ww[i]);
          nrm2 = fma(zz[i], zz[i], nrm2);
        }
        const T inv = T(1) / sqrt(fmax(nrm2, Eps<T>::tiny));
        for (int i = 0; i < r; ++i) zz[i] *= inv;
      }
      __syncthreads();
    }
    // ---- x = H_0 H_1 ... H_{r-3} z
    for (int c = r - 3; c >= 0; --c) {
      const T tc = tau[c];
      if (tc == T(0)) continue;                      // (wave-uniform)
      T dot = T(0);
      for (int i = c + 1 + lane; i < r; i += 64) dot = fma(A[i * LD + c], zz[i], dot);
      dot = wsum(dot) * tc;
      for (int i = c + 1 + lane; i < r; i += 64) zz[i] = fma(-dot, A[i * LD + c], zz[i]);
      __syncthreads();
    }
    // ---- x_i x_j (twice off the diagonal) in packed order over the matrix storage, then one contraction per generator
    __syncthreads();
    for (int idx = lane; idx < P; idx += 64) {
      int i = (int)((sqrtf(8.f * (float)idx + 1.f) - 1.f) * 0.5f);
      while ((i + 1) * (i + 2) / 2 <= idx) ++i;
      while (i * (i + 1) / 2 > idx) --i;
      const int j = idx - i * (i + 1) / 2;
      A[idx] = (i == j ? T(1) : T(2)) * zz[i] * zz[j];
    }
    __syncthreads();
    for (int a = 0; a < n; ++a) {
      const T* col = gt + (size_t)a * Pp;
      T part = T(0);
      for (int idx = lane; idx < P; idx += 64) part = fma(col[idx], A[idx], part);
      part = wsum(part);
      if (lane == 0) gv[b * ldgv + a] = fma(sc, ts[a], -coef * part);
    }
  } else {
    const int rho = clipped ? rho_of[arow] : -1;
    for (int a = lane; a < n; a += 64) {
      const T u = rho >= 0 ? dt[(size_t)a * Mp + rho] : T(0);
      gv[b * ldgv + a] = fma(sc, ts[a], -coef * u);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------
constexpr size_t kWaveLdsMax = 150 * 1024;

template <typename T>
bool lmi_wave_eligible_t(const RayenPack* p) {
  int n_lmi = 0, r = 0;
  for (const RayenSegment& g : p->segs) {
    if (g.type == RAYEN_SEG_LMI) { ++n_lmi; r = g.dim; }
    else if (g.type != RAYEN_SEG_LIN) return false;
  }
  if (n_lmi != 1 || r < 2) return false;
  return lds_elems(r, p->n, p->k) * sizeof(T) <= kWaveLdsMax;
}

inline void lmi_wave_free_image(LmiWaveImage* img) {
  if (img == nullptr) return;
  for (void* ptr : {img->gt, img->dt, img->nat, img->nrm, img->y0, (void*)img->lin_id, (void*)img->rho_of})
    if (ptr) (void)hipFree(ptr);
  delete img;
}

template <typename T>
int lmi_wave_build_t(const RayenPack* p, LmiWaveImage** out, int64_t* bytes) {
  LmiWaveImage* img = new LmiWaveImage();
  const int n = p->n, k = p->k;
  const RayenSegment* lmi = nullptr;
  std::vector<int32_t> ids, rho_of((size_t)(p->n_rows > 0 ? p->n_rows : 1), -1);
  std::vector<const double*> lin_rows;
  for (size_t s = 0; s < p->segs.size(); ++s) {
    const RayenSegment& g = p->segs[s];
    if (g.type == RAYEN_SEG_LMI) { lmi = &g; img->lmi_seg = (int)s; }
    if (g.type == RAYEN_SEG_LIN)
      for (int rr = 0; rr < g.nrows; ++rr) {
        rho_of[(size_t)(g.row0 + rr)] = (int32_t)lin_rows.size();
        lin_rows.push_back(p->W.data() + (size_t)(g.row0 + rr) * n);
        ids.push_back((int32_t)s);
        ids.push_back(g.row0 + rr);
      }
  }
  const int r = lmi->dim, m = (int)lin_rows.size(), P = r * (r + 1) / 2;
  img->r = r; img->n = n; img->k = k; img->m = m; img->P = P; img->identity = p->out_identity;
  img->lmi_row0 = lmi->row0; img->n_rows = p->n_rows;
  img->Pp = (P + 63) / 64 * 64;
  img->Mp = m > 0 ? (m + 63) / 64 * 64 : 64;
  img->Kp = (k + 63) / 64 * 64;
  std::vector<T> gt((size_t)n * img->Pp, T(0)), dt((size_t)n * img->Mp, T(0)), y0((size_t)k);
  for (int a = 0; a < n; ++a) {
    for (int idx = 0; idx < P; ++idx) gt[(size_t)a * img->Pp + idx] = (T)p->W[(size_t)(lmi->row0 + idx) * n + a];
    for (int rr = 0; rr < m; ++rr) dt[(size_t)a * img->Mp + rr] = (T)lin_rows[rr][a];
  }
  for (int i = 0; i < k; ++i) y0[i] = (T)p->y0[i];
  if (ids.empty()) ids.assign(2, 0);
  auto up = [&](const void* host, size_t nbytes, void** dev) {
    if (hipMalloc(dev, nbytes) != hipSuccess) return false;
    if (hipMemcpy(*dev, host, nbytes, hipMemcpyHostToDevice) != hipSuccess) return false;
    img->bytes += (int64_t)nbytes;
    return true;
  };
  bool ok = up(gt.data(), gt.size() * sizeof(T), &img->gt) && up(dt.data(), dt.size() * sizeof(T), &img->dt) &&
            up(y0.data(), y0.size() * sizeof(T), &img->y0) &&
            up(ids.data(), ids.size() * sizeof(int32_t), reinterpret_cast<void**>(&img->lin_id)) &&
            up(rho_of.data(), rho_of.size() * sizeof(int32_t), reinterpret_cast<void**>(&img->rho_of));
  if (ok && !p->out_identity) {
    std::vector<T> nat((size_t)n * img->Kp, T(0)), nrm((size_t)k * n);
    for (int i = 0; i < k; ++i)
      for (int a = 0; a < n; ++a) {
        nat[(size_t)a * img->Kp + i] = (T)p->NA_E[(size_t)i * n + a];
        nrm[(size_t)i * n + a] = (T)p->NA_E[(size_t)i * n + a];
      }
    ok = up(nat.data(), nat.size() * sizeof(T), &img->nat) && up(nrm.data(), nrm.size() * sizeof(T), &img->nrm);
  }
  if (!ok) { lmi_wave_free_image(img); return RAYEN_E_ALLOC; }
  // (pack creation is the one place that may touch function attributes: large matrices need more than 64 KiB of LDS)
  const size_t lds = lds_elems(r, n, k) * sizeof(T);
  if (lds > 48 * 1024 && lds <= kWaveLdsMax) {
    const void* fwd = reinterpret_cast<const void*>(&lmi_wave_kernel<T>);
    const void* bwd = reinterpret_cast<const void*>(&lmi_wave_bwd_kernel<T>);
    if (hipFuncSetAttribute(fwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWaveLdsMax) != hipSuccess ||
        hipFuncSetAttribute(bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWaveLdsMax) != hipSuccess) {
      (void)hipGetLastError();
      lmi_wave_free_image(img);
      return RAYEN_E_LAUNCH;
    }
  }
  *bytes = img->bytes;
  *out = img;
  return RAYEN_OK;
}

// (the image is also built for packs only the workgroup-per-sample forward of rayen_lmi_block.h holds)
template <typename T>
bool lmi_wave_serves_t(const LmiWaveImage* img) {
  return img != nullptr && lds_elems(img->r, img->n, img->k) * sizeof(T) <= kWaveLdsMax;
}

template <typename T>
int lmi_wave_forward_t(const RayenPack* p, const LmiWaveImage* img, const T* v, int64_t B, int64_t ldv, T* y, int64_t ldy,
                       T* kappa, int32_t* active, int32_t* nan_flag, hipStream_t stream) {
  (void)p;
  if (!lmi_wave_serves_t<T>(img)) return RAYEN_E_UNSUPPORTED;
  if (B == 0) return RAYEN_OK;
  if (B > 0x7fffffffLL) return RAYEN_E_UNSUPPORTED;
  const size_t lds = lds_elems(img->r, img->n, img->k) * sizeof(T);
  hipLaunchKernelGGL(lmi_wave_kernel<T>, dim3((unsigned)B), dim3(64), lds, stream, static_cast<const T*>(img->gt),
                     static_cast<const T*>(img->dt), static_cast<const T*>(img->nat), static_cast<const T*>(img->y0),
                     img->lin_id, img->r, img->n, img->k, img->m, img->P, img->Pp, img->Mp, img->Kp, img->identity,
                     img->lmi_seg, v, B, ldv, y, ldy, kappa, active, nan_flag);
  return hipGetLastError() == hipSuccess ? RAYEN_OK : RAYEN_E_LAUNCH;
}

template <typename T>
int lmi_wave_backward_t(const RayenPack* p, const LmiWaveImage* img, const T* v, int64_t B, int64_t ldv, const T* kappa,
                        const int32_t* active, const T* gy, int64_t ldg, T* gv, int64_t ldgv, hipStream_t stream) {
  (void)p;
  if (!lmi_wave_serves_t<T>(img)) return RAYEN_E_UNSUPPORTED;
  if (B == 0) return RAYEN_OK;
  if (B > 0x7fffffffLL) return RAYEN_E_UNSUPPORTED;
  const size_t lds = lds_elems(img->r, img->n, img->k) * sizeof(T);
  hipLaunchKernelGGL(lmi_wave_bwd_kernel<T>, dim3((unsigned)B), dim3(64), lds, stream, static_cast<const T*>(img->gt),
                     static_cast<const T*>(img->dt), static_cast<const T*>(img->nrm), img->rho_of, img->r, img->n, img->k,
                     img->P, img->Pp, img->Mp, img->identity, img->lmi_seg, v, B, ldv, kappa, active, gy, ldg, gv, ldgv);
  return hipGetLastError() == hipSuccess ? RAYEN_OK : RAYEN_E_LAUNCH;
}

}  // namespace lw
}  // namespace rayen
