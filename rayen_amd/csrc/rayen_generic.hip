// Generic path: one lane = one sample, any shape that fits LDS, fp32 and fp64.
//
// Replaces the op chain of rayen/constraint_module.py:351-474 (computeKappa +
// forwardForRAYEN + getyFromz) with one kernel.  A workgroup of BLOCK lanes owns
// BLOCK samples.  The block's directions are staged once in LDS, transposed
// ([n][BLOCK+1], conflict-free both for the coalesced fill and for the
// lane-per-sample reads).  Every constant is wave-uniform: the rows of W are
// walked in blocks of eight, stored as Wg[rb][j][8], so the eight multipliers of
// column j arrive with ONE scalar load (s_load_dwordx8) and feed eight v_fmac
// whose other operand is the single ds_read of v_j.  All reductions of
// computeKappa (row max, v'Gv, ||Uv||^2, the SOC quadratic, lambda_max) are
// lane-local: no cross-lane traffic, no atomics, no second pass over HBM.
//
// HBM traffic per sample: n loads + k stores (the algorithmic minimum); the
// constants stream through the scalar cache / L2.
#include "rayen_internal.h"
#include "rayen_tiles.h"

#include <cmath>
#include <cstring>
#include <type_traits>

namespace rayen {

template <typename T> struct Num;
template <> struct Num<float> {
  static constexpr int bisect_iters = 32;
  __device__ static float tiny() { return 1.0e-30f; }
};
template <> struct Num<double> {
  static constexpr int bisect_iters = 60;
  __device__ static double tiny() { return 1.0e-290; }
};

// 1/x for the Sturm recurrence: only the SIGN sequence of q matters there, so the 1-ulp hardware
// reciprocal is enough in fp32 (fp64 keeps the exact division)
__device__ __forceinline__ float sturm_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ double sturm_rcp(double x) { return 1.0 / x; }

template <typename T> __device__ __forceinline__ T fma_(T a, T b, T c);
template <> __device__ __forceinline__ float fma_(float a, float b, float c) { return fmaf(a, b, c); }
template <> __device__ __forceinline__ double fma_(double a, double b, double c) { return fma(a, b, c); }

// acc[r] = sum_j Wg[rb][j][r] * vT[j][lane]
template <typename T, int LD>
__device__ __forceinline__ void dot8(const T* __restrict__ Wg, int rb, int ncols,
                                     const T* vcol, T (&acc)[kRowBlock]) {
#pragma unroll
  for (int r = 0; r < kRowBlock; ++r) acc[r] = T(0);
  const T* __restrict__ w = Wg + (size_t)rb * (size_t)ncols * kRowBlock;
  int j = 0;
  for (; j + 4 <= ncols; j += 4) {
    const T x0 = vcol[(j + 0) * LD];
    const T x1 = vcol[(j + 1) * LD];
    const T x2 = vcol[(j + 2) * LD];
    const T x3 = vcol[(j + 3) * LD];
#pragma unroll
    for (int r = 0; r < kRowBlock; ++r) acc[r] = fma_(w[(j + 0) * kRowBlock + r], x0, acc[r]);
#pragma unroll
    for (int r = 0; r < kRowBlock; ++r) acc[r] = fma_(w[(j + 1) * kRowBlock + r], x1, acc[r]);
#pragma unroll
    for (int r = 0; r < kRowBlock; ++r) acc[r] = fma_(w[(j + 2) * kRowBlock + r], x2, acc[r]);
#pragma unroll
    for (int r = 0; r < kRowBlock; ++r) acc[r] = fma_(w[(j + 3) * kRowBlock + r], x3, acc[r]);
  }
  for (; j < ncols; ++j) {
    const T x0 = vcol[j * LD];
#pragma unroll
    for (int r = 0; r < kRowBlock; ++r) acc[r] = fma_(w[j * kRowBlock + r], x0, acc[r]);
  }
}

// Largest eigenvalue of the symmetric r x r matrix whose packed lower triangle
// sits in this lane's LDS column (entry idx at lm[idx * BLOCK]).  Householder
// tridiagonalisation followed by Sturm-count bisection: backward stable, no
// convergence test, identical control flow on every lane.  This is the
// per-sample replacement of torch.linalg.eigvalsh + max (constraint_module.py:424-425).
template <typename T, int BLOCK>
__device__ T lambda_max_packed(T* lm, int r) {
  const int npk = r * (r + 1) / 2;
  auto A = [&](int i, int j) -> T& { return lm[(size_t)((i * (i + 1)) / 2 + j) * BLOCK]; };  // i >= j
  if (r == 1) return A(0, 0);
  T* hv = lm + (size_t)npk * BLOCK;  // Householder vector
  T* hp = hv + (size_t)r * BLOCK;    // p, then w
  T* dd = hp + (size_t)r * BLOCK;    // diagonal of T
  T* e2 = dd + (size_t)r * BLOCK;    // squared off-diagonal of T

  for (int c = 0; c + 2 < r; ++c) {
    const T x0 = A(c + 1, c);
    T sigma = T(0);
#pragma unroll 4
    for (int i = c + 2; i < r; ++i) { const T a = A(i, c); sigma = fma_(a, a, sigma); }
    const T mu = sqrt(fma_(x0, x0, sigma));
    const bool act = sigma > T(0);
    const T v0 = (x0 <= T(0)) ? (x0 - mu) : (-sigma / (x0 + mu));
    const T beta = act ? (T(2) * v0 * v0 / (sigma + v0 * v0)) : T(0);
    const T inv_v0 = act ? (T(1) / v0) : T(0);
    hv[(size_t)(c + 1) * BLOCK] = T(1);
#pragma unroll 4
    for (int i = c + 2; i < r; ++i) hv[(size_t)i * BLOCK] = A(i, c) * inv_v0;
    dd[(size_t)c * BLOCK] = A(c, c);
    e2[(size_t)c * BLOCK] = act ? (mu * mu) : (x0 * x0);

    T pv = T(0);
    for (int i = c + 1; i < r; ++i) {
      // row i of the symmetric matrix: A(i, c+1..i) then A(i+1..r-1, i); independent loads, 4 in flight
      T acc0 = T(0), acc1 = T(0);
#pragma unroll 4
      for (int j = c + 1; j <= i; ++j) acc0 = fma_(A(i, j), hv[(size_t)j * BLOCK], acc0);
#pragma unroll 4
      for (int j = i + 1; j < r; ++j) acc1 = fma_(A(j, i), hv[(size_t)j * BLOCK], acc1);
      T acc = acc0 + acc1;
      acc *= beta;
      hp[(size_t)i * BLOCK] = acc;
      pv = fma_(acc, hv[(size_t)i * BLOCK], pv);
    }
    const T K = T(0.5) * beta * pv;
#pragma unroll 4
    for (int i = c + 1; i < r; ++i) hp[(size_t)i * BLOCK] -= K * hv[(size_t)i * BLOCK];
    for (int i = c + 1; i < r; ++i) {
      const T vi = hv[(size_t)i * BLOCK], wi = hp[(size_t)i * BLOCK];
#pragma unroll 4
      for (int j = c + 1; j <= i; ++j)
        A(i, j) -= vi * hp[(size_t)j * BLOCK] + wi * hv[(size_t)j * BLOCK];
    }
  }
  dd[(size_t)(r - 2) * BLOCK] = A(r - 2, r - 2);
  dd[(size_t)(r - 1) * BLOCK] = A(r - 1, r - 1);
  { const T e = A(r - 1, r - 2); e2[(size_t)(r - 2) * BLOCK] = e * e; }

  // Gershgorin bracket of lambda_max: max diag <= lambda_max <= max(d_i + |e_{i-1}| + |e_i|)
  T lo = dd[0], hi = dd[0], emax = T(0);
  {
    T eprev = T(0);
    for (int i = 0; i < r; ++i) {
      const T d = dd[(size_t)i * BLOCK];
      const T enext = (i + 1 < r) ? sqrt(e2[(size_t)i * BLOCK]) : T(0);
      lo = (i == 0) ? d : fmax(lo, d);
      hi = (i == 0) ? (d + enext) : fmax(hi, d + eprev + enext);
      emax = fmax(emax, enext);
      eprev = enext;
    }
  }
  const T pivmin = Num<T>::tiny() * fmax(T(1), emax * emax);
  for (int it = 0; it < Num<T>::bisect_iters; ++it) {
    const T mid = T(0.5) * (lo + hi);
    T q = dd[0] - mid;
    int below = q < T(0);
    for (int i = 1; i < r; ++i) {
      if (fabs(q) < pivmin) q = -pivmin;
      q = dd[(size_t)i * BLOCK] - mid - e2[(size_t)(i - 1) * BLOCK] * sturm_rcp(q);
      below += q < T(0);
    }
    if (below == r) hi = mid; else lo = mid;  // all eigenvalues < mid  ->  lambda_max < mid
  }
  return T(0.5) * (lo + hi);
}

// Register-resident variant of lambda_max_packed for a compile-time size R: the packed matrix, the
// Householder vectors and the tridiagonal all live in VGPRs (every index is a constant after full
// unrolling), so the (4/3) R^3 flops run at VALU rate instead of one LDS round trip per operand.
// `a` holds the packed lower triangle, entry (i,j) at i(i+1)/2 + j; it is destroyed.
template <typename T, int R>
__device__ __forceinline__ T lambda_max_regs(T (&a)[R * (R + 1) / 2]) {
  auto IDX = [](int i, int j) { return i * (i + 1) / 2 + j; };  // i >= j
  T dd[R], e2[R];
  if (R == 1) return a[0];
#pragma unroll
  for (int c = 0; c + 2 < R; ++c) {
    const T x0 = a[IDX(c + 1, c)];
    T sigma = T(0);
#pragma unroll
    for (int i = c + 2; i < R; ++i) sigma = fma_(a[IDX(i, c)], a[IDX(i, c)], sigma);
    const T mu = sqrt(fma_(x0, x0, sigma));
    const bool act = sigma > T(0);
    const T v0 = (x0 <= T(0)) ? (x0 - mu) : (-sigma / (x0 + mu));
    const T beta = act ? (T(2) * v0 * v0 / (sigma + v0 * v0)) : T(0);
    const T inv_v0 = act ? (T(1) / v0) : T(0);
    T hv[R], hp[R];
#pragma unroll
    for (int i = 0; i < R; ++i) hv[i] = T(0);
    hv[c + 1] = T(1);
#pragma unroll
    for (int i = c + 2; i < R; ++i) hv[i] = a[IDX(i, c)] * inv_v0;
    dd[c] = a[IDX(c, c)];
    e2[c] = act ? (mu * mu) : (x0 * x0);
    T pv = T(0);
#pragma unroll
    for (int i = c + 1; i < R; ++i) {
      T acc = T(0);
#pragma unroll
      for (int j = c + 1; j < R; ++j) acc = fma_((i >= j) ? a[IDX(i, j)] : a[IDX(j, i)], hv[j], acc);
      acc *= beta;
      hp[i] = acc;
      pv = fma_(acc, hv[i], pv);
    }
    const T K = T(0.5) * beta * pv;
#pragma unroll
    for (int i = c + 1; i < R; ++i) hp[i] -= K * hv[i];
#pragma unroll
    for (int i = c + 1; i < R; ++i)
#pragma unroll
      for (int j = c + 1; j <= i; ++j) a[IDX(i, j)] -= hv[i] * hp[j] + hp[i] * hv[j];
  }
  dd[R - 2] = a[IDX(R - 2, R - 2)];
  dd[R - 1] = a[IDX(R - 1, R - 1)];
  e2[R - 2] = a[IDX(R - 1, R - 2)] * a[IDX(R - 1, R - 2)];
  e2[R - 1] = T(0);

  T lo = dd[0], hi = dd[0] + sqrt(e2[0]), emax = T(0), eprev = T(0);
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const T enext = (i + 1 < R) ? sqrt(e2[i]) : T(0);
    lo = (i == 0) ? dd[i] : fmax(lo, dd[i]);
    hi = (i == 0) ? hi : fmax(hi, dd[i] + eprev + enext);
    emax = fmax(emax, enext);
    eprev = enext;
  }
  const T pivmin = Num<T>::tiny() * fmax(T(1), emax * emax);
  for (int it = 0; it < Num<T>::bisect_iters; ++it) {
    const T mid = T(0.5) * (lo + hi);
    T q = dd[0] - mid;
    int below = q < T(0);
#pragma unroll
    for (int i = 1; i < R; ++i) {
      if (fabs(q) < pivmin) q = -pivmin;
      q = dd[i] - mid - e2[i - 1] * sturm_rcp(q);
      below += q < T(0);
    }
    if (below == R) hi = mid; else lo = mid;
  }
  return T(0.5) * (lo + hi);
}

// VG = true: directions are read straight from global memory by each lane (no LDS tile), for
// subspace dimensions too large to stage; slower (uncoalesced, cache-served) but size-independent.
template <typename T, int BLOCK, int RREG, bool VG>
__global__ __launch_bounds__(BLOCK) void generic_fwd_kernel(
    const T* __restrict__ Wg, const T* __restrict__ Ng, const T* __restrict__ y0,
    const GSeg* __restrict__ segs, int n_gseg, int out_nrb, int k, int n, int lmi_words,
    const T* __restrict__ v, int64_t B, int64_t ldv, T* __restrict__ y, int64_t ldy,
    T* __restrict__ kappa_out, int32_t* __restrict__ active_out, int32_t* __restrict__ nan_flag,
    int old_mode, int64_t ldk) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int LD = VG ? 1 : BLOCK + 1;
  const int n_pad = (n + kRowBlock - 1) / kRowBlock * kRowBlock;
  T* vT = reinterpret_cast<T*>(smem_raw);  // [n_pad][LD]  (absent when VG)
  T* sc = vT + (VG ? 0 : (size_t)n_pad * LD);  // [BLOCK] clip factor
  T* lmi = sc + BLOCK;                     // [lmi_words][BLOCK]

  const int tid = threadIdx.x;
  const int64_t b0 = (int64_t)blockIdx.x * BLOCK;
  const int nb = (int)((B - b0) < (int64_t)BLOCK ? (B - b0) : (int64_t)BLOCK);

  if (!VG) {
    // coalesced fill of the transposed tile (zero for the tail samples and the pad rows)
    for (int idx = tid; idx < BLOCK * n_pad; idx += BLOCK) {
      const int bl = idx / n_pad;
      const int j = idx - bl * n_pad;
      T x = T(0);
      if (bl < nb && j < n) x = v[(b0 + bl) * ldv + j];
      vT[j * LD + bl] = x;
    }
    __syncthreads();
  }

  const T* vcol = VG ? (v + (tid < nb ? (b0 + tid) : b0) * ldv) : (vT + tid);
  T kap = T(0);
  int aseg = -1, arow = 0;
  T acc[kRowBlock];

  for (int s = 0; s < n_gseg; ++s) {
    const GSeg sg = segs[s];
    if (sg.type == RAYEN_SEG_LIN) {
      for (int b = 0; b < sg.nrb; ++b) {
        dot8<T, LD>(Wg, sg.rb0 + b, n, vcol, acc);
#pragma unroll
        for (int r = 0; r < kRowBlock; ++r) {
          if (acc[r] > kap) { kap = acc[r]; aseg = sg.seg; arow = sg.row0 + b * kRowBlock + r; }
        }
      }
    } else if (sg.type == RAYEN_SEG_QUAD_SYM || sg.type == RAYEN_SEG_QUAD_FAC) {
      dot8<T, LD>(Wg, sg.aux_rb, n, vcol, acc);
      const T lin = acc[0];
      T qf = T(0);
      // (a symmetric form is evaluated through its factor when the image holds one: fuzz set 570 -- a dense 44 x 44
      // form on a set with equalities -- is 1.4e-5 off in fp32 as v'(G v) and 7e-6 as ||U v||^2)
      const bool own_factor = sg.type == RAYEN_SEG_QUAD_SYM && sg.fnrb > 0;
      const bool fac = sg.type == RAYEN_SEG_QUAD_FAC || own_factor;
      const int qb0 = own_factor ? sg.frb0 : sg.rb0;
      const int qnb = own_factor ? sg.fnrb : sg.nrb;
      for (int b = 0; b < qnb; ++b) {
        dot8<T, LD>(Wg, qb0 + b, n, vcol, acc);
        if (!fac) {
#pragma unroll
          for (int r = 0; r < kRowBlock; ++r) {
            const int jj = b * kRowBlock + r;  // rows of G beyond n are zero padding
            qf = fma_(acc[r], (!VG || jj < n) ? vcol[jj * LD] : T(0), qf);
          }
        } else {
#pragma unroll
          for (int r = 0; r < kRowBlock; ++r) qf = fma_(acc[r], acc[r], qf);
        }
      }
      const T kq = lin + sqrt(fmax(qf, T(0)));
      if (kq > kap) { kap = kq; aseg = sg.seg; arow = 0; }
    } else if (sg.type == RAYEN_SEG_SOC) {
      dot8<T, LD>(Wg, sg.aux_rb, n, vcol, acc);
      const T cr = acc[0], br = acc[1];
      T mm = T(0);
      for (int b = 0; b < sg.nrb; ++b) {
        dot8<T, LD>(Wg, sg.rb0 + b, n, vcol, acc);
#pragma unroll
        for (int r = 0; r < kRowBlock; ++r) mm = fma_(acc[r], acc[r], mm);
      }
      // a' x^2 + b' x + c' = 0  (constraint_module.py:392-396), a' < 0
      const T tau = (T)sg.f0, ap = (T)sg.f1;
      const T cp = mm - cr * cr;
      const T bp = T(2) * br - T(2) * cr * tau;
      const T disc = bp * bp - T(4) * ap * cp;
      T ks = T(0);
      if (disc >= T(0)) {
        const T root = sqrt(disc);
        const T inv2a = T(0.5) / ap;
        ks = fmax((-bp - root) * inv2a, (-bp + root) * inv2a);
      }
      if (ks > kap) { kap = ks; aseg = sg.seg; arow = 0; }
    } else if (sg.type == RAYEN_SEG_LMI) {
      T lam;
      if constexpr (RREG > 0) {
        // matrix of size dim <= RREG assembled straight into registers; rows/columns beyond dim are
        // decoupled and far below every real eigenvalue
        constexpr int NPK = RREG * (RREG + 1) / 2;
        T a[NPK];
#pragma unroll
        for (int i = 0; i < RREG; ++i)
#pragma unroll
          for (int j = 0; j <= i; ++j) a[i * (i + 1) / 2 + j] = (i == j && i >= sg.dim) ? T(-1e18) : T(0);
#pragma unroll
        for (int b = 0; b < (NPK + kRowBlock - 1) / kRowBlock; ++b) {
          if (b < sg.nrb) {
            dot8<T, LD>(Wg, sg.rb0 + b, n, vcol, acc);
#pragma unroll
            for (int r = 0; r < kRowBlock; ++r) {
              const int idx = b * kRowBlock + r;  // packed index in a dim x dim matrix == the same index here
              if (idx < NPK && idx < sg.nrows) a[idx] = acc[r];
            }
          }
        }
        lam = lambda_max_regs<T, RREG>(a);
      } else {
        T* lm = lmi + tid;
        for (int b = 0; b < sg.nrb; ++b) {
          dot8<T, LD>(Wg, sg.rb0 + b, n, vcol, acc);
#pragma unroll
          for (int r = 0; r < kRowBlock; ++r) {
            const int idx = b * kRowBlock + r;
            if (idx < sg.nrows) lm[(size_t)idx * BLOCK] = acc[r];
          }
        }
        lam = lambda_max_packed<T, BLOCK>(lm, sg.dim);
      }
      if (lam > kap) { kap = lam; aseg = sg.seg; arow = 0; }
    }
  }

  const bool live = tid < nb;
  T scale = T(1) / fmax(T(1), kap);
  if (old_mode) {
    // RAYEN_old head (rayen/constraint_module.py:460-466): step 1/(exp(beta) + kappa(v_bar)) along
    // v_bar = v/||v||, i.e. y = y0 + N v / (||v|| exp(beta) + kappa(v)); beta is column n of the input
    T nrm2 = T(0);
    for (int jj = 0; jj < n; ++jj) nrm2 = fma_(vcol[jj * LD], vcol[jj * LD], nrm2);
    const T beta = live ? v[(b0 + tid) * ldv + n] : T(0);
    const T nrm = sqrt(nrm2);
    scale = nrm > T(0) ? T(1) / (nrm * exp(beta) + kap) : T(0);
  }
  if (live) {
    if (kappa_out) kappa_out[(b0 + tid) * ldk] = kap;       // (ldk = 1 but for the sets of rayen_abi.hip::mixed_forward)
    if (active_out) { active_out[2 * (b0 + tid)] = aseg; active_out[2 * (b0 + tid) + 1] = arow; }
  }
  if (y == nullptr) return;

  bool bad = false;
  if (Ng == nullptr && VG) {
    if (live)
      for (int jj = 0; jj < n; ++jj) {
        const T val = fma_(vcol[jj], scale, y0[jj]);
        bad |= (val != val);
        y[(b0 + tid) * ldy + jj] = val;
      }
  } else if (Ng == nullptr) {
    // NA_E = I: y = y0 + v * scale, written coalesced from the staged tile
    sc[tid] = scale;
    __syncthreads();
    for (int idx = tid; idx < nb * n; idx += BLOCK) {
      const int bl = idx / n;
      const int j = idx - bl * n;
      const T val = fma_(vT[j * LD + bl], sc[bl], y0[j]);
      bad |= (val != val);
      y[(b0 + bl) * ldy + j] = val;
    }
  } else {
    for (int b = 0; b < out_nrb; ++b) {
      dot8<T, LD>(Ng, b, n, vcol, acc);
      if (live) {
#pragma unroll
        for (int r = 0; r < kRowBlock; ++r) {
          const int i = b * kRowBlock + r;
          if (i < k) {
            const T val = fma_(acc[r], scale, y0[i]);
            bad |= (val != val);
            y[(b0 + tid) * ldy + i] = val;
          }
        }
      }
    }
  }
  if (nan_flag && bad) atomicOr(nan_flag, 1);
}

// ---------------------------------------------------------------------------------------------
// host side: image construction and launch
// ---------------------------------------------------------------------------------------------

template <typename T>
static int upload(const std::vector<T>& host, T** dev, int64_t* bytes) {
  *dev = nullptr;
  if (host.empty()) return RAYEN_OK;
  if (hipMalloc(dev, host.size() * sizeof(T)) != hipSuccess) return RAYEN_E_ALLOC;
  if (hipMemcpy(*dev, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess)
    return RAYEN_E_ALLOC;
  *bytes += (int64_t)(host.size() * sizeof(T));
  return RAYEN_OK;
}

// append rows [row0, row0+nrows) of a row-major [*, ncols] matrix as row blocks
template <typename T>
static int append_rowblocks(std::vector<T>& out, const double* M, int row0, int nrows, int ncols) {
  const int nrb = (nrows + kRowBlock - 1) / kRowBlock;
  const size_t base = out.size();
  out.resize(base + (size_t)nrb * ncols * kRowBlock, T(0));
  for (int r = 0; r < nrows; ++r) {
    const int rb = r / kRowBlock, rr = r % kRowBlock;
    for (int j = 0; j < ncols; ++j)
      out[base + ((size_t)rb * ncols + j) * kRowBlock + rr] = (T)M[(size_t)(row0 + r) * ncols + j];
  }
  return nrb;
}

template <typename T>
int generic_build(const RayenPack* p, GenericImage<T>* img) {
  std::vector<T> Wg;
  std::vector<GSeg> gs;
  int rb = 0, lmi_words = 0;
  for (size_t s = 0; s < p->segs.size(); ++s) {
    const RayenSegment& sg = p->segs[s];
    GSeg g;
    std::memset(&g, 0, sizeof(g));
    g.type = sg.type;
    g.seg = (int32_t)s;
    g.row0 = sg.row0;
    g.nrows = sg.nrows;
    g.dim = sg.dim;
    g.f0 = sg.f0;
    g.f1 = sg.f1;
    g.aux_rb = -1;
    if (sg.type == RAYEN_SEG_LMI && img->skip_lmi) {      // (rayen_abi.hip::mixed_forward: the LMI is another kernel's)
      g.type = -1;
      g.rb0 = rb;
      g.nrb = 0;
      gs.push_back(g);
      continue;
    }
    if (sg.type == RAYEN_SEG_QUAD_SYM || sg.type == RAYEN_SEG_QUAD_FAC) {
      g.aux_rb = rb;
      rb += append_rowblocks(Wg, p->W.data(), sg.aux_row, 1, p->n);
    } else if (sg.type == RAYEN_SEG_SOC) {
      g.aux_rb = rb;
      rb += append_rowblocks(Wg, p->W.data(), sg.aux_row, 2, p->n);
    } else if (sg.type == RAYEN_SEG_LMI) {
      const int words = sg.nrows + 4 * sg.dim;
      if (words > lmi_words) lmi_words = words;
    }
    g.rb0 = rb;
    g.nrb = append_rowblocks(Wg, p->W.data(), sg.row0, sg.nrows, p->n);
    rb += g.nrb;
    if (sg.type == RAYEN_SEG_QUAD_SYM && sg.nrows == p->n && p->n <= 128) {
      // the forward's factor of G (eigen-factor, <= n rows; the Jacobi sweeps are O(n^3) on the host: not for the
      // 800 x 800 forms this path also serves); the backward keeps reading G itself
      const std::vector<std::vector<double>> rows = psd_factor_rows(p->W.data() + (size_t)sg.row0 * p->n, p->n);
      if (!rows.empty()) {
        std::vector<double> flat;
        for (const std::vector<double>& u : rows) flat.insert(flat.end(), u.begin(), u.end());
        g.frb0 = rb;
        g.fnrb = append_rowblocks(Wg, flat.data(), 0, (int)rows.size(), p->n);
        rb += g.fnrb;
      }
    }
    gs.push_back(g);
  }
  img->n_rb = rb;
  img->n_gseg = (int)gs.size();
  img->lmi_words = lmi_words;
  int rc = upload(Wg, &img->Wg, &img->bytes);
  if (rc) return rc;
  if (!gs.empty()) {
    if (hipMalloc(&img->segs, gs.size() * sizeof(GSeg)) != hipSuccess) return RAYEN_E_ALLOC;
    if (hipMemcpy(img->segs, gs.data(), gs.size() * sizeof(GSeg), hipMemcpyHostToDevice) != hipSuccess)
      return RAYEN_E_ALLOC;
    img->bytes += (int64_t)(gs.size() * sizeof(GSeg));
  }
  std::vector<T> y0(p->y0.begin(), p->y0.end());
  rc = upload(y0, &img->y0, &img->bytes);
  if (rc) return rc;
  if (!p->out_identity) {
    std::vector<T> Ng;
    img->out_nrb = append_rowblocks(Ng, p->NA_E.data(), 0, p->k, p->n);
    rc = upload(Ng, &img->Ng, &img->bytes);
    if (rc) return rc;
    // NA_E' for the backward: rows = n, columns = k
    std::vector<double> NT((size_t)p->n * p->k);
    for (int i = 0; i < p->k; ++i)
      for (int j = 0; j < p->n; ++j) NT[(size_t)j * p->k + i] = p->NA_E[(size_t)i * p->n + j];
    std::vector<T> NTg;
    append_rowblocks(NTg, NT.data(), 0, p->n, p->k);
    rc = upload(NTg, &img->NTg, &img->bytes);
    if (rc) return rc;
  }
  img->built = true;
  return RAYEN_OK;
}

template <typename T>
void generic_free(GenericImage<T>* img) {
  if (img->Wg) (void)hipFree(img->Wg);
  if (img->Ng) (void)hipFree(img->Ng);
  if (img->NTg) (void)hipFree(img->NTg);
  if (img->y0) (void)hipFree(img->y0);
  if (img->segs) (void)hipFree(img->segs);
  *img = GenericImage<T>();
}

template <typename T>
static size_t generic_lds_bytes(int n, int lmi_words, int block) {
  const int n_pad = (n + kRowBlock - 1) / kRowBlock * kRowBlock;
  return sizeof(T) * ((size_t)n_pad * (block + 1) + block + (size_t)lmi_words * block);
}

constexpr size_t kLdsSoft = 64 * 1024;    // keep >= 2 workgroups per CU when possible
constexpr size_t kLdsHard = 160 * 1024;   // gfx950 LDS per CU

// does the lane-per-sample forward hold the pack's largest LMI (its packed matrix + four vectors per lane)?
template <typename T>
bool generic_holds_lmis(const RayenPack* p) {
  int words = 0;
  for (const RayenSegment& g : p->segs)
    if (g.type == RAYEN_SEG_LMI && g.nrows + 4 * g.dim > words) words = g.nrows + 4 * g.dim;
  return generic_lds_bytes<T>(p->n, words, 64) <= kLdsHard || generic_lds_bytes<T>(0, words, 64) + sizeof(T) * 64 <= kLdsHard;
}
template bool generic_holds_lmis<float>(const RayenPack*);
template bool generic_holds_lmis<double>(const RayenPack*);

template <typename T>
int generic_block_for(const RayenPack* p, const GenericImage<T>& img) {
  for (int block : {256, 128, 64})
    if (generic_lds_bytes<T>(p->n, img.lmi_words, block) <= kLdsSoft) return block;
  if (generic_lds_bytes<T>(p->n, img.lmi_words, 64) <= kLdsHard) return 64;
  return 0;
}

template <typename T, int BLOCK, int RREG, bool VG = false>
static int launch_fwd(const RayenPack* p, const GenericImage<T>& img, const T* v, int64_t B,
                      int64_t ldv, T* y, int64_t ldy, T* kappa, int32_t* active, int32_t* nan_flag,
                      int old_mode, hipStream_t stream, int64_t ldk) {
  const int lmi_words = RREG > 0 ? 0 : img.lmi_words;  // the register path needs no LDS scratch
  const size_t lds = VG ? sizeof(T) * ((size_t)BLOCK + (size_t)lmi_words * BLOCK)
                        : generic_lds_bytes<T>(p->n, lmi_words, BLOCK);
  if (lds > kLdsHard) return RAYEN_E_UNSUPPORTED;
  auto kern = generic_fwd_kernel<T, BLOCK, RREG, VG>;
  if (lds > 48 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return RAYEN_E_LAUNCH;
  }
  const int64_t grid = (B + BLOCK - 1) / BLOCK;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(BLOCK), lds, stream, img.Wg, img.Ng, img.y0,
                     img.segs, img.n_gseg, img.out_nrb, p->k, p->n, lmi_words, v, B, ldv, y, ldy,
                     kappa, active, nan_flag, old_mode, ldk);
  return hipGetLastError() == hipSuccess ? RAYEN_OK : RAYEN_E_LAUNCH;
}

// size class of the register-resident eigen-solve (fp32, one LMI of size <= 24), 0 = use LDS
template <typename T>
static int lmi_reg_class(const RayenPack* p) {
  if (!std::is_same<T, float>::value) return 0;
  int n_lmi = 0, dim = 0;
  for (const RayenSegment& g : p->segs)
    if (g.type == RAYEN_SEG_LMI) { ++n_lmi; dim = g.dim; }
  if (n_lmi != 1 || dim > 24 || dim < 2) return 0;
  return (dim + 3) / 4 * 4;
}

template <typename T>
int generic_forward(const RayenPack* p, const GenericImage<T>& img, const T* v, int64_t B,
                    int64_t ldv, T* y, int64_t ldy, T* kappa, int32_t* active, int32_t* nan_flag,
                    int old_mode, hipStream_t stream, int64_t ldk) {
  if (B == 0) return RAYEN_OK;
  if constexpr (std::is_same<T, float>::value) {
    if (!img.skip_lmi && generic_lds_bytes<T>(p->n, 0, 64) <= kLdsHard) {
      switch (lmi_reg_class<T>(p)) {
        case 4: return launch_fwd<T, 64, 4>(p, img, v, B, ldv, y, ldy, kappa, active, nan_flag, old_mode, stream, ldk);
        case 8: return launch_fwd<T, 64, 8>(p, img, v, B, ldv, y, ldy, kappa, active, nan_flag, old_mode, stream, ldk);
        case 12: return launch_fwd<T, 64, 12>(p, img, v, B, ldv, y, ldy, kappa, active, nan_flag, old_mode, stream, ldk);
        case 16: return launch_fwd<T, 64, 16>(p, img, v, B, ldv, y, ldy, kappa, active, nan_flag, old_mode, stream, ldk);
        case 20: return launch_fwd<T, 64, 20>(p, img, v, B, ldv, y, ldy, kappa, active, nan_flag, old_mode, stream, ldk);
        case 24: return launch_fwd<T, 64, 24>(p, img, v, B, ldv, y, ldy, kappa, active, nan_flag, old_mode, stream, ldk);
        default: break;
      }
    }
  }
  switch (generic_block_for<T>(p, img)) {
    case 256: return launch_fwd<T, 256, 0>(p, img, v, B, ldv, y, ldy, kappa, active, nan_flag, old_mode, stream, ldk);
    case 128: return launch_fwd<T, 128, 0>(p, img, v, B, ldv, y, ldy, kappa, active, nan_flag, old_mode, stream, ldk);
    case 64: return launch_fwd<T, 64, 0>(p, img, v, B, ldv, y, ldy, kappa, active, nan_flag, old_mode, stream, ldk);
    default:  // n too large for an LDS tile: directions straight from global memory
      return launch_fwd<T, 64, 0, true>(p, img, v, B, ldv, y, ldy, kappa, active, nan_flag, old_mode, stream, ldk);
  }
}

// ---------------------------------------------------------------------------------------------
// backward: grad_v = s N'g - [kappa > 1] s^2 (N'g . v) grad kappa(v),  s = 1/max(1, kappa)
//
// Same lane = sample layout.  grad kappa is evaluated on the active constraint only (the one the
// forward recorded), which is what autograd gives for rayen/constraint_module.py:351-474: max ->
// its arg-max, relu, sqrt, the SOC root (implicit differentiation of a'x^2+b'x+c'=0) and
// eigvalsh -> x x' for the top eigenvector.  Each lane evaluates the gradient of its own active
// constraint only.
// ---------------------------------------------------------------------------------------------

// u_j += sum_r Wg[rb][j][r] * w[r]   (W' w for one row block), masked per lane
template <typename T, int LD>
__device__ __forceinline__ void axpy8_t(const T* __restrict__ Wg, int rb, int ncols, T* ucol,
                                        const T (&w)[kRowBlock], bool on) {
  const T* __restrict__ wp = Wg + (size_t)rb * (size_t)ncols * kRowBlock;
  for (int j = 0; j < ncols; ++j) {
    T a = T(0);
#pragma unroll
    for (int r = 0; r < kRowBlock; ++r) a = fma_(wp[j * kRowBlock + r], w[r], a);
    if (on) ucol[j * LD] += a;
  }
}

// Unit eigenvector of lambda_max for the packed symmetric matrix in lm (destroyed): Cholesky of
// C = (lambda + shift) I - A (positive definite by construction) + 3 steps of inverse iteration.
template <typename T, int BLOCK>
__device__ void top_eigenvector_packed(T* lm, int r, T lam, T* x) {
  auto A = [&](int i, int j) -> T& { return lm[(size_t)((i * (i + 1)) / 2 + j) * BLOCK]; };  // i >= j
  T scale = fabs(lam);
  for (int i = 0; i < r; ++i) scale = fmax(scale, fabs(A(i, i)));
  const T shift = (sizeof(T) == 4 ? T(2e-4) : T(1e-9)) * fmax(scale, Num<T>::tiny());
  for (int i = 0; i < r; ++i) {
    for (int j = 0; j <= i; ++j) A(i, j) = -A(i, j);
    A(i, i) += lam + shift;
  }
  // in-place Cholesky C = L L'
  for (int j = 0; j < r; ++j) {
    T d = A(j, j);
    for (int p = 0; p < j; ++p) { const T l = A(j, p); d = fma_(-l, l, d); }
    d = sqrt(fmax(d, shift * T(1e-3)));
    A(j, j) = d;
    const T inv = T(1) / d;
    for (int i = j + 1; i < r; ++i) {
      T a = A(i, j);
      for (int p = 0; p < j; ++p) a = fma_(-A(i, p), A(j, p), a);
      A(i, j) = a * inv;
    }
  }
  for (int i = 0; i < r; ++i) x[(size_t)i * BLOCK] = T(1) + T(0.01) * T(i);  // not orthogonal to anything special
  for (int it = 0; it < 3; ++it) {
    for (int i = 0; i < r; ++i) {  // L z = x
      T a = x[(size_t)i * BLOCK];
      for (int p = 0; p < i; ++p) a = fma_(-A(i, p), x[(size_t)p * BLOCK], a);
      x[(size_t)i * BLOCK] = a / A(i, i);
    }
    T nrm = T(0);
    for (int i = r - 1; i >= 0; --i) {  // L' y = z
      T a = x[(size_t)i * BLOCK];
      for (int p = i + 1; p < r; ++p) a = fma_(-A(p, i), x[(size_t)p * BLOCK], a);
      a /= A(i, i);
      x[(size_t)i * BLOCK] = a;
      nrm = fma_(a, a, nrm);
    }
    const T inv = T(1) / sqrt(fmax(nrm, Num<T>::tiny()));
    for (int i = 0; i < r; ++i) x[(size_t)i * BLOCK] *= inv;
  }
}

template <typename T, int BLOCK>
__global__ __launch_bounds__(BLOCK) void generic_bwd_kernel(
    const T* __restrict__ Wg, const T* __restrict__ NTg, const GSeg* __restrict__ segs, int n_gseg,
    int k, int n, int w_rows, int lmi_words, const T* __restrict__ v, int64_t B, int64_t ldv,
    const T* __restrict__ kappa, const int32_t* __restrict__ active, const T* __restrict__ grad_y,
    int64_t ldg, T* __restrict__ grad_v, int64_t ldgv, int old_mode) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int LD = BLOCK + 1;
  const int n_pad = (n + kRowBlock - 1) / kRowBlock * kRowBlock;
  const int k_pad = (k + kRowBlock - 1) / kRowBlock * kRowBlock;
  T* vT = reinterpret_cast<T*>(smem_raw);  // [n_pad][LD]
  T* tT = vT + (size_t)n_pad * LD;         // [n_pad][LD]  N'g, finally the result
  // One region serves grad_y first ([k_pad][LD], only until t = NA_E' g is formed) and then grad kappa
  // ([n_pad][LD]) followed by U v / M v of the active segment ([w_rows][LD]): after the fill every lane
  // touches its own column only, so the hand-over needs no barrier.
  T* gT = tT + (size_t)n_pad * LD;
  T* uT = gT;
  T* wT = uT + (size_t)n_pad * LD;
  const int shared_rows = (NTg != nullptr && k_pad > n_pad + w_rows) ? k_pad : n_pad + w_rows;
  T* lmi = gT + (size_t)shared_rows * LD;  // [lmi_words][BLOCK]

  const int tid = threadIdx.x;
  const int64_t b0 = (int64_t)blockIdx.x * BLOCK;
  const int nb = (int)((B - b0) < (int64_t)BLOCK ? (B - b0) : (int64_t)BLOCK);

  for (int idx = tid; idx < BLOCK * n_pad; idx += BLOCK) {
    const int bl = idx / n_pad, j = idx - bl * n_pad;
    T x = T(0), g = T(0);
    if (bl < nb && j < n) {
      x = v[(b0 + bl) * ldv + j];
      if (NTg == nullptr) g = grad_y[(b0 + bl) * ldg + j];
    }
    vT[j * LD + bl] = x;
    tT[j * LD + bl] = g;
  }
  if (NTg != nullptr) {
    for (int idx = tid; idx < BLOCK * k_pad; idx += BLOCK) {
      const int bl = idx / k_pad, i = idx - bl * k_pad;
      gT[i * LD + bl] = (bl < nb && i < k) ? grad_y[(b0 + bl) * ldg + i] : T(0);
    }
  }
  __syncthreads();

  const T* vcol = vT + tid;
  T* tcol = tT + tid;
  T* ucol = uT + tid;
  T acc[kRowBlock];
  const bool live = tid < nb;

  if (NTg != nullptr) {  // t = NA_E' g
    const int nrb = n_pad / kRowBlock;
    for (int b = 0; b < nrb; ++b) {
      dot8<T, LD>(NTg, b, k, gT + tid, acc);
#pragma unroll
      for (int r = 0; r < kRowBlock; ++r) tcol[(b * kRowBlock + r) * LD] = acc[r];
    }
  }
  for (int j = 0; j < n_pad; ++j) ucol[j * LD] = T(0);  // (the region held this lane's grad_y column until here)
  T tv = T(0);
  for (int j = 0; j < n; ++j) tv = fma_(tcol[j * LD], vcol[j * LD], tv);

  const T kap = live ? kappa[b0 + tid] : T(0);
  const int aseg = live ? active[2 * (b0 + tid)] : -1;
  const int arow = live ? active[2 * (b0 + tid) + 1] : 0;
  // RAYEN: s = 1/max(1,kappa), so kappa only matters once it clips.  RAYEN_old (old_mode):
  // s = 1/(r e^beta + kappa) with r = ||v||: kappa always matters, and r, beta get gradients too.
  T r_nrm = T(0), e_beta = T(0);
  if (old_mode) {
    T nrm2 = T(0);
    for (int j = 0; j < n; ++j) nrm2 = fma_(vcol[j * LD], vcol[j * LD], nrm2);
    r_nrm = sqrt(nrm2);
    e_beta = live ? exp(v[(b0 + tid) * ldv + n]) : T(0);
  }
  const bool clipped = old_mode ? (live && aseg >= 0 && r_nrm > T(0)) : (live && kap > T(1) && aseg >= 0);
  const T sc = old_mode ? (r_nrm > T(0) ? T(1) / (r_nrm * e_beta + kap) : T(0)) : T(1) / fmax(T(1), kap);

  // Every lane walks ONLY the constraint that set its kappa: its own segment record, its own row
  // blocks (the reads of W become per-lane loads; lanes that share a segment share the addresses).
  // A wave pays for the distinct constraint TYPES among its lanes, not for the whole constraint set.
  if (clipped) {
    const GSeg sg = segs[aseg];
    const bool on = true;
    if (sg.type == RAYEN_SEG_LIN) {
      if (on) {
        const int r = arow - sg.row0;
        const T* __restrict__ wp = Wg + ((size_t)(sg.rb0 + r / kRowBlock) * n) * kRowBlock + (r % kRowBlock);
        for (int j = 0; j < n; ++j) ucol[j * LD] = wp[(size_t)j * kRowBlock];
      }
    } else if (sg.type == RAYEN_SEG_QUAD_SYM) {
      T qf = T(0);
      for (int b = 0; b < sg.nrb; ++b) {
        dot8<T, LD>(Wg, sg.rb0 + b, n, vcol, acc);
#pragma unroll
        for (int r = 0; r < kRowBlock; ++r) {
          const int j = b * kRowBlock + r;
          qf = fma_(acc[r], vcol[j * LD], qf);
          if (on && j < n) ucol[j * LD] = acc[r];
        }
      }
      const T rinv = qf > T(0) ? T(1) / sqrt(qf) : T(0);
      const T* __restrict__ ph = Wg + (size_t)sg.aux_rb * n * kRowBlock;
      if (on)
        for (int j = 0; j < n; ++j) ucol[j * LD] = fma_(ucol[j * LD], rinv, ph[(size_t)j * kRowBlock]);
    } else if (sg.type == RAYEN_SEG_QUAD_FAC || sg.type == RAYEN_SEG_SOC) {
      T ssq = T(0);
      T* wcol = wT + tid;
      for (int b = 0; b < sg.nrb; ++b) {
        dot8<T, LD>(Wg, sg.rb0 + b, n, vcol, acc);
#pragma unroll
        for (int r = 0; r < kRowBlock; ++r) {
          ssq = fma_(acc[r], acc[r], ssq);
          wcol[(b * kRowBlock + r) * LD] = acc[r];
        }
      }
      // u = coef_w * W'w + coef_0 * aux0 + coef_1 * aux1
      T cw, c0, c1 = T(0);
      if (sg.type == RAYEN_SEG_QUAD_FAC) {
        cw = ssq > T(0) ? T(1) / sqrt(ssq) : T(0);
        c0 = T(1);
      } else {
        dot8<T, LD>(Wg, sg.aux_rb, n, vcol, acc);
        const T cr = acc[0], br = acc[1];
        const T tau = (T)sg.f0, ap = (T)sg.f1;
        const T bp = T(2) * br - T(2) * cr * tau;
        const T den = T(2) * ap * kap + bp;  // dF/dkappa at the root
        const T inv = den != T(0) ? T(-1) / den : T(0);
        cw = T(2) * inv;                        // d c'/dv = 2 M'Mv - 2 (c.v) c
        c0 = inv * (T(-2) * cr - T(2) * tau * kap);
        c1 = inv * T(2) * kap;                  // kappa * d b'/dv = kappa (2 b - 2 tau c)
      }
      for (int b = 0; b < sg.nrb; ++b) {
        T w8[kRowBlock];
#pragma unroll
        for (int r = 0; r < kRowBlock; ++r) w8[r] = wcol[(b * kRowBlock + r) * LD] * cw;
        axpy8_t<T, LD>(Wg, sg.rb0 + b, n, ucol, w8, on);
      }
      const T* __restrict__ ax = Wg + (size_t)sg.aux_rb * n * kRowBlock;
      if (on)
        for (int j = 0; j < n; ++j)
          ucol[j * LD] += c0 * ax[(size_t)j * kRowBlock] + c1 * ax[(size_t)j * kRowBlock + 1];
    } else if (sg.type == RAYEN_SEG_LMI) {
      T* lm = lmi + tid;
      const int r = sg.dim;
      for (int b = 0; b < sg.nrb; ++b) {
        dot8<T, LD>(Wg, sg.rb0 + b, n, vcol, acc);
#pragma unroll
        for (int q = 0; q < kRowBlock; ++q) {
          const int idx = b * kRowBlock + q;
          if (idx < sg.nrows) lm[(size_t)idx * BLOCK] = acc[q];
        }
      }
      T* x = lm + (size_t)sg.nrows * BLOCK;
      top_eigenvector_packed<T, BLOCK>(lm, r, kap, x);
      // u_b = sum_{p>=q} (2 - [p==q]) x_p x_q GL[(p,q)][b]: reuse the row-block walk with w = x x'
      for (int b = 0; b < sg.nrb; ++b) {
        T w8[kRowBlock];
#pragma unroll
        for (int q = 0; q < kRowBlock; ++q) {
          const int idx = b * kRowBlock + q;
          T wv = T(0);
          if (idx < sg.nrows) {
            int p = (int)((sqrt((double)(8 * idx + 1)) - 1.0) * 0.5);
            while ((p + 1) * (p + 2) / 2 <= idx) ++p;
            while (p * (p + 1) / 2 > idx) --p;
            const int qq = idx - p * (p + 1) / 2;
            wv = x[(size_t)p * BLOCK] * x[(size_t)qq * BLOCK] * (p == qq ? T(1) : T(2));
          }
          w8[q] = wv;
        }
        axpy8_t<T, LD>(Wg, sg.rb0 + b, n, ucol, w8, on);
      }
    }
  }

  // grad_v = s t - [clipped] s^2 (t.v) u
  if (!old_mode) {
    const T coef = clipped ? sc * sc * tv : T(0);
    for (int j = 0; j < n; ++j) tcol[j * LD] = sc * tcol[j * LD] - coef * ucol[j * LD];
  } else {
    // grad_v = s t - s^2 (t.v) (e^beta v / r + grad kappa),  grad_beta = -s^2 (t.v) r e^beta
    const T coef = sc * sc * tv;
    const T dir = r_nrm > T(0) ? e_beta / r_nrm : T(0);
    for (int j = 0; j < n; ++j)
      tcol[j * LD] = sc * tcol[j * LD] - coef * (dir * vcol[j * LD] + (clipped ? ucol[j * LD] : T(0)));
    if (live) grad_v[(b0 + tid) * ldgv + n] = -coef * r_nrm * e_beta;
  }
  __syncthreads();
  for (int idx = tid; idx < nb * n; idx += BLOCK) {
    const int bl = idx / n, j = idx - bl * n;
    grad_v[(b0 + bl) * ldgv + j] = tT[j * LD + bl];
  }
}

template <typename T>
static void bwd_shape(const RayenPack* p, const GenericImage<T>& img, int* w_rows, int* lmi_words) {
  *w_rows = 0;
  *lmi_words = 0;
  for (const RayenSegment& g : p->segs) {
    if (g.type == RAYEN_SEG_QUAD_FAC || g.type == RAYEN_SEG_SOC) {
      const int rows = (g.nrows + kRowBlock - 1) / kRowBlock * kRowBlock;
      if (rows > *w_rows) *w_rows = rows;
    }
    if (g.type == RAYEN_SEG_LMI && !img.skip_lmi && g.nrows + 2 * g.dim > *lmi_words) *lmi_words = g.nrows + 2 * g.dim;
  }
}

template <typename T>
static size_t bwd_lds_bytes(const RayenPack* p, int w_rows, int lmi_words, int block) {
  const int n_pad = (p->n + kRowBlock - 1) / kRowBlock * kRowBlock;
  const int k_pad = p->out_identity ? 0 : (p->k + kRowBlock - 1) / kRowBlock * kRowBlock;
  const int shared_rows = k_pad > n_pad + w_rows ? k_pad : n_pad + w_rows;  // grad_y, then grad kappa + U v
  return sizeof(T) * ((size_t)(2 * n_pad + shared_rows) * (block + 1) + (size_t)lmi_words * block);
}

template <typename T, int BLOCK>
static int launch_bwd(const RayenPack* p, const GenericImage<T>& img, int w_rows, int lmi_words, const T* v,
                      int64_t B, int64_t ldv, const T* kappa, const int32_t* active, const T* grad_y,
                      int64_t ldg, T* grad_v, int64_t ldgv, int old_mode, hipStream_t stream) {
  const size_t lds = bwd_lds_bytes<T>(p, w_rows, lmi_words, BLOCK);
  auto kern = generic_bwd_kernel<T, BLOCK>;
  if (lds > 48 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)lds) != hipSuccess)
    return RAYEN_E_LAUNCH;
  const int64_t grid = (B + BLOCK - 1) / BLOCK;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(BLOCK), lds, stream, img.Wg, img.NTg, img.segs,
                     img.n_gseg, p->k, p->n, w_rows, lmi_words, v, B, ldv, kappa, active, grad_y, ldg, grad_v,
                     ldgv, old_mode);
  return hipGetLastError() == hipSuccess ? RAYEN_OK : RAYEN_E_LAUNCH;
}

template <typename T>
int generic_backward(const RayenPack* p, const GenericImage<T>& img, const T* v, int64_t B, int64_t ldv,
                     const T* kappa, const int32_t* active, const T* grad_y, int64_t ldg, T* grad_v,
                     int64_t ldgv, int old_mode, hipStream_t stream) {
  if (B == 0) return RAYEN_OK;
  int w_rows, lmi_words;
  bwd_shape<T>(p, img, &w_rows, &lmi_words);
  int block = 0;
  for (int cand : {256, 128, 64})
    if (bwd_lds_bytes<T>(p, w_rows, lmi_words, cand) <= kLdsSoft) { block = cand; break; }
  if (block == 0 && bwd_lds_bytes<T>(p, w_rows, lmi_words, 64) <= kLdsHard) block = 64;
  switch (block) {
    case 256: return launch_bwd<T, 256>(p, img, w_rows, lmi_words, v, B, ldv, kappa, active, grad_y, ldg, grad_v, ldgv, old_mode, stream);
    case 128: return launch_bwd<T, 128>(p, img, w_rows, lmi_words, v, B, ldv, kappa, active, grad_y, ldg, grad_v, ldgv, old_mode, stream);
    case 64: return launch_bwd<T, 64>(p, img, w_rows, lmi_words, v, B, ldv, kappa, active, grad_y, ldg, grad_v, ldgv, old_mode, stream);
    default: return RAYEN_E_UNSUPPORTED;
  }
}

// does the lane-per-sample backward stage this pack at all?  (the predicate generic_backward dispatches on: what
// rayen_pack_info reports as the backward family must be what a call would run)
template <typename T>
bool generic_backward_serves(const RayenPack* p, const GenericImage<T>& img) {
  int w_rows, lmi_words;
  bwd_shape<T>(p, img, &w_rows, &lmi_words);
  return bwd_lds_bytes<T>(p, w_rows, lmi_words, 64) <= kLdsHard;
}
template bool generic_backward_serves<float>(const RayenPack*, const GenericImage<float>&);
template bool generic_backward_serves<double>(const RayenPack*, const GenericImage<double>&);

#define RAYEN_INSTANTIATE(T)                                                                        \
  template int generic_build<T>(const RayenPack*, GenericImage<T>*);                               \
  template void generic_free<T>(GenericImage<T>*);                                                  \
  template int generic_block_for<T>(const RayenPack*, const GenericImage<T>&);                      \
  template int generic_forward<T>(const RayenPack*, const GenericImage<T>&, const T*, int64_t,      \
                                  int64_t, T*, int64_t, T*, int32_t*, int32_t*, int, hipStream_t,   \
                                  int64_t);                                                         \
  template int generic_backward<T>(const RayenPack*, const GenericImage<T>&, const T*, int64_t,     \
                                   int64_t, const T*, const int32_t*, const T*, int64_t, T*,        \
                                   int64_t, int, hipStream_t);
RAYEN_INSTANTIATE(float)
RAYEN_INSTANTIATE(double)

}  // namespace rayen
