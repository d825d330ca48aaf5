// LMI path with FOUR lanes per sample (a DPP quad), fp32 and fp64 -- shared by rayen_lmi_quad32.hip
// and rayen_lmi_quad64.hip (one translation unit per element type so that they compile side by side).
//
// kappa_LMI(v) = relu(lambda_max(sum_a v_a G_a)),  G_a = -L'F_a L folded with NA_E
// (rayen/constraint_module.py:43-52, 401-449: einsum, L' S L, eigvalsh, max, relu).
//
// The lane-per-sample kernels keep a whole r x r matrix per lane: at config 4's batch (16384) that is
// 256 waves on 1024 SIMDs, each running ~30k dependent instructions, and in fp64 the matrix does not
// fit the register file at all.  Here the four lanes of a quad share one sample: lane `sub` owns rows
// i = 4t + sub of the (full, symmetric) matrix, a quarter of the registers and a quarter of the work,
// and every exchange is a DPP quad permutation (no LDS, no barriers):
//   * S = sum_a v_a G_a      each lane forms its rows from an LDS image of the full G_a;
//   * Householder tridiagonalisation, distributed: column norm and u'p by quad sums, A u and the rank-2
//     update with the other lanes' u_j, w_j read through quad_perm broadcasts;
//   * lambda_max of the tridiagonal by MULTI-section on the Sturm count: each lane evaluates two
//     shifts per round (eight section points per quad), so the bracket shrinks 9x per round.
// Same numerics as lambda_max_regs of rayen_generic.hip (Householder formulas, Gershgorin bracket,
// pivmin guard); rows/columns beyond the true size are decoupled and far below every eigenvalue.
// Linear rows (if any) are split over the four lanes as well.  Serves packs = [linear rows] + one LMI.
#pragma once

#include <cstdlib>
#include <type_traits>
#include <vector>

#include "rayen_internal.h"

namespace rayen {

struct LmiQuadImage {
  void* data = nullptr;   // device: [Wf n*R*R | Wlin m*n | N k*n (absent when identity) | y0 k] of T
  int32_t* lin_id = nullptr;  // device: [m][2] (segment, W row) of every linear row
  void* wrow = nullptr;       // device: [n_rows][n] of T, row-major W (backward: the active linear row)
  void* wm = nullptr;         // device: [ks][R4 * R4][64] of T: the generators as v_mfma_*_16x16x4 A operands (forward)
  int ks = 0;                 // K-steps of that image: ceil(n / 4)
  int r = 0, R = 0, n = 0, k = 0, m = 0, identity = 0, lmi_seg = 0;
  int64_t elems = 0;      // number of T elements in `data`
  int64_t bytes = 0;
};

namespace lq {

// developer build (scripts/ubench/lq_stamps.hip): s_memtime at the phase boundaries of the forward kernel, one wave per
// block, into a __device__ array the micro-benchmark reads back.  Expands to nothing in the library.
#ifdef RAYEN_LQ_STAMPS
__device__ unsigned long long lq_stamp_buf[4096 * 8];
#define RAYEN_LQ_STAMP(slot)                                                                         \
  do {                                                                                               \
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 1024)                                                \
      lq_stamp_buf[((blockIdx.x << 2) + (threadIdx.x >> 6)) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define RAYEN_LQ_STAMP(slot) do { } while (0)
#endif

template <int I, int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    sfor<I + 1, N>(f);
  }
}

template <int CTRL>
__device__ __forceinline__ int dpp_(int x) { return __builtin_amdgcn_mov_dpp(x, CTRL, 0xF, 0xF, true); }  // (quad_perm: every lane has a source)
template <int CTRL>
__device__ __forceinline__ float dpp_(float x) { return __int_as_float(dpp_<CTRL>(__float_as_int(x))); }
template <int CTRL>
__device__ __forceinline__ double dpp_(double x) {
  const int lo = dpp_<CTRL>(__double2loint(x)), hi = dpp_<CTRL>(__double2hiint(x));
  return __hiloint2double(hi, lo);
}
// value of lane M of the quad
template <int M, typename U>
__device__ __forceinline__ U qb(U x) { return dpp_<M * 0x55>(x); }
template <typename U>
__device__ __forceinline__ U qsum(U x) {
  x += dpp_<0xB1>(x);  // lanes [1,0,3,2]
  x += dpp_<0x4E>(x);  // lanes [2,3,0,1]
  return x;
}
template <typename U>
__device__ __forceinline__ U qmax(U x) {
  x = fmax(x, dpp_<0xB1>(x));
  return fmax(x, dpp_<0x4E>(x));
}
template <typename U>
__device__ __forceinline__ U qmin(U x) {
  x = fmin(x, dpp_<0xB1>(x));
  return fmin(x, dpp_<0x4E>(x));
}

template <typename T> struct Lim;
template <> struct Lim<float> {
  static constexpr int rounds = 8;   // 9^-8 < 2^-25 of the Gershgorin bracket: the midpoint is within 1e-8 of its width
  __device__ static float tiny() { return 1.0e-30f; }
  __device__ static float sqrt_(float x) { return __builtin_amdgcn_sqrtf(x); }  // 1 ulp, no denormal fix-up sequence
  __device__ static float rcp(float x) { return __builtin_amdgcn_rcpf(x); }   // 1 ulp
  __device__ static int expo(float x) { return __builtin_amdgcn_frexp_expf(x); }
  __device__ static float scale2(float x, int e) { return __builtin_amdgcn_ldexpf(x, e); }
  __device__ static unsigned sign_word(float x) { return (unsigned)__float_as_int(x); }
};
template <> struct Lim<double> {
  static constexpr int rounds = 19;  // 9^-19 < 2^-60
  __device__ static double tiny() { return 1.0e-290; }
  __device__ static double sqrt_(double x) { return sqrt(x); }
  __device__ static double rcp(double x) { return 1.0 / x; }
  __device__ static int expo(double x) { return __builtin_amdgcn_frexp_exp(x); }
  __device__ static double scale2(double x, int e) { return __builtin_amdgcn_ldexp(x, e); }
  __device__ static unsigned sign_word(double x) { return (unsigned)__double2hiint(x); }
};

template <typename T> __device__ __forceinline__ T fma_(T a, T b, T c);
template <> __device__ __forceinline__ float fma_(float a, float b, float c) { return fmaf(a, b, c); }
template <> __device__ __forceinline__ double fma_(double a, double b, double c) { return fma(a, b, c); }

// Two adjacent matrix columns in one register pair: on gfx950 an fp32 pair is ONE v_pk_fma_f32 / v_pk_mul_f32 /
// v_pk_add_f32 (the packed forms are what the 157 TFLOP/s fp32 vector peak is quoted on), so the Householder
// sweep, the formation of S and the Sturm recurrences below issue half the instructions of their scalar forms.
// (fp64 pairs compile to two scalar instructions; same source.)
template <typename T> using V2 = T __attribute__((ext_vector_type(2)));
// (the scalar passes through an opaque statement: taken straight out of the HIGH half of a register pair, hipcc encodes
// the broadcast as op_sel:[0,1,..] on the packed instruction -- the one operand selection that reads 0 in lanes 48-63 now and
// then while an MFMA runs on the SIMD, rayen_amd/_build.py, scripts/ubench/pkfma_hazard.hip.  In a register of its own it is
// broadcast from the low half.)
template <typename T> __device__ __forceinline__ V2<T> splat(T x) {
  asm volatile("" : "+v"(x));
  return V2<T>{x, x};
}
template <typename T> __device__ __forceinline__ V2<T> fma2(V2<T> a, V2<T> b, V2<T> c) {
  return __builtin_elementwise_fma(a, b, c);
}

// One Householder step on column C of the distributed matrix: a[t][jj] = (A[4t + sub][2jj], A[4t + sub][2jj + 1]).
// KEEP (backward): the reflector H_C = I - beta u u', u = (1, hv...) on rows >= C+1, is kept -- hv in the
// column it has just annihilated (rows >= C+2 of column C, never touched again), beta in e2[C + R]'s place
// (the caller passes arrays of 2R) -- and e2[C] holds the SIGNED sub-diagonal entry instead of its square.
template <typename T, int R4, int C, bool KEEP = false>
__device__ __forceinline__ void hh_step(V2<T> (&a)[R4][2 * R4], T (&dd)[4 * R4], T (&e2)[KEEP ? 8 * R4 : 4 * R4],
                                        const int sub) {
  constexpr int R = 4 * R4, I1 = C + 1, H = R / 2;
  constexpr int T0 = I1 / 4;  // first row group with a live row (rows >= I1)
  constexpr int J0 = I1 / 2;  // first column pair with a live column (columns >= I1)
  const T x0 = qb<(I1 & 3)>(a[I1 >> 2][C >> 1][C & 1]);
  T sig = T(0);
  sfor<0, R4>([&](auto it) {
    constexpr int t = decltype(it)::value;
    if constexpr (4 * t + 3 >= C + 2) {
      const T x = a[t][C >> 1][C & 1];
      if constexpr (4 * t >= C + 2) {
        sig = fma_(x, x, sig);
      } else {
        sig += (4 * t + sub >= C + 2) ? x * x : T(0);
      }
    }
  });
  sig = qsum(sig);
  const T mu = Lim<T>::sqrt_(fma_(x0, x0, sig));
  const bool act = sig > T(0);
  const T v0 = (x0 <= T(0)) ? (x0 - mu) : (-sig * Lim<T>::rcp(x0 + mu));
  // KEEP: u = (1, x / v0) with beta = 2 v0^2 / (sig + v0^2) (the stored form the backward reads back);
  // otherwise u = (v0, x) as it stands in the column, beta = 2 / (sig + v0^2): one reciprocal and no scaling pass
  const T bden = Lim<T>::rcp(fma_(v0, v0, sig));
  const T beta = act ? (KEEP ? T(2) * v0 * v0 * bden : T(2) * bden) : T(0);
  const T inv_v0 = (KEEP && act) ? Lim<T>::rcp(v0) : T(0);
  dd[C] = qb<(C & 3)>(a[C >> 2][C >> 1][C & 1]);
  if constexpr (KEEP) {
    e2[C] = act ? mu : x0;   // H x = mu e_1 (mu = ||x|| > 0); untouched column when sigma = 0
    e2[R + C] = beta;
  } else {
    e2[C] = act ? (mu * mu) : (x0 * x0);
  }

  T hv[R4], p[R4];
  sfor<0, R4>([&](auto it) {
    constexpr int t = decltype(it)::value;
    const int i = 4 * t + sub;
    hv[t] = T(0);
    p[t] = T(0);
    if constexpr (4 * t >= C + 2) {          // every row of the group lies below the sub-diagonal
      hv[t] = KEEP ? a[t][C >> 1][C & 1] * inv_v0 : a[t][C >> 1][C & 1];
      if constexpr (KEEP) a[t][C >> 1][C & 1] = hv[t];
    } else if constexpr (t >= T0) {          // the group(s) straddling rows C+1, C+2
      const T top = KEEP ? T(1) : v0;
      hv[t] = (i == I1) ? top : ((i >= C + 2) ? (KEEP ? a[t][C >> 1][C & 1] * inv_v0 : a[t][C >> 1][C & 1]) : T(0));
      if constexpr (KEEP) a[t][C >> 1][C & 1] = (i >= C + 2) ? hv[t] : a[t][C >> 1][C & 1];
    }
  });
  // u over the live columns, every lane's copy (zero in the dead half of a straddling pair)
  V2<T> hb[H];
  sfor<J0, H>([&](auto ij) {
    constexpr int jj = decltype(ij)::value;
    constexpr int j0 = 2 * jj, j1 = 2 * jj + 1;
    hb[jj][0] = (j0 >= I1) ? qb<(j0 & 3)>(hv[j0 >> 2]) : T(0);
    hb[jj][1] = qb<(j1 & 3)>(hv[j1 >> 2]);
  });
  // p = beta A u over the live block.  Column pairs outermost: the R4 - T0 row groups' sums are independent chains, and a
  // packed fma that reads the accumulator the previous instruction wrote costs a wait state on gfx950 (hipcc fills it
  // with s_nop when the source order offers nothing else: one row group after the other was 45 s_nops per step).
  {
    V2<T> accs[R4];
    sfor<T0, R4>([&](auto it) { accs[decltype(it)::value] = splat(T(0)); });
    sfor<J0, H>([&](auto ij) {
      constexpr int jj = decltype(ij)::value;
      sfor<T0, R4>([&](auto it) {
        constexpr int t = decltype(it)::value;
        accs[t] = fma2<T>(a[t][jj], hb[jj], accs[t]);
      });
    });
    sfor<T0, R4>([&](auto it) {
      constexpr int t = decltype(it)::value;
      p[t] = accs[t][0] + accs[t][1];
    });
  }
  T pv = T(0);
  sfor<T0, R4>([&](auto it) {
    constexpr int t = decltype(it)::value;
    if constexpr (4 * t >= I1) p[t] = p[t] * beta;
    else p[t] = (4 * t + sub >= I1) ? p[t] * beta : T(0);
    pv = fma_(p[t], hv[t], pv);
  });
  pv = qsum(pv);
  const T K = T(0.5) * beta * pv;
  sfor<T0, R4>([&](auto it) {
    constexpr int t = decltype(it)::value;
    p[t] = fma_(-K, hv[t], p[t]);  // now w
  });
  // A -= u w' + w u'
  V2<T> wb[H];
  sfor<J0, H>([&](auto ij) {
    constexpr int jj = decltype(ij)::value;
    constexpr int j0 = 2 * jj, j1 = 2 * jj + 1;
    wb[jj][0] = (j0 >= I1) ? qb<(j0 & 3)>(p[j0 >> 2]) : T(0);
    wb[jj][1] = qb<(j1 & 3)>(p[j1 >> 2]);
  });
  sfor<T0, R4>([&](auto it) {
    constexpr int t = decltype(it)::value;
    const V2<T> mh = splat(-hv[t]), mw = splat(-p[t]);
    sfor<J0, H>([&](auto ij) {
      constexpr int jj = decltype(ij)::value;
      a[t][jj] = fma2<T>(mh, wb[jj], a[t][jj]);
      a[t][jj] = fma2<T>(mw, hb[jj], a[t][jj]);
    });
  });
}

// lambda_max of the symmetric R x R matrix spread over a quad (true size r; rows beyond it carry the decoupled
// pad diagonal); every lane returns the same value.
//
// Tridiagonal form by the distributed Householder sweep, then MULTI-section on the Sturm count: eight section
// points per round (lane `sub` takes points 2 sub + 1 and 2 sub + 2 as the two halves of packed registers), the
// bracket shrinks 9x per round.  The count is taken on the characteristic polynomials of the leading blocks,
//     p_0 = 1,  p_1 = d_0 - x,  p_{i+1} = (d_i - x) p_i - e_{i-1}^2 p_{i-1},
// (number of sign changes = number of eigenvalues below x) -- no division: one packed multiply, add and fma per
// step for both points, the signs shifted into a bit mask.  The matrix is first mapped into [-1, 0] (shift by the
// Gershgorin top, divide by its radius) so that |d_i - x| <= 1 and e^2 <= 1: the recurrence cannot overflow, and
// every fourth step both chains are renormalised by the exponent of their larger member against underflow.
template <typename T, int R4>
__device__ __forceinline__ T lambda_max_quad(V2<T> (&a)[R4][2 * R4], const int r, const int sub) {
  constexpr int R = 4 * R4;
  T dd[R], e2[R];
  sfor<0, R - 2>([&](auto ic) {
    constexpr int c = decltype(ic)::value;
    hh_step<T, R4, c>(a, dd, e2, sub);
  });
  dd[R - 2] = qb<((R - 2) & 3)>(a[(R - 2) >> 2][(R - 2) >> 1][0]);
  dd[R - 1] = qb<((R - 1) & 3)>(a[(R - 1) >> 2][(R - 1) >> 1][1]);
  {
    const T e = qb<((R - 1) & 3)>(a[(R - 1) >> 2][(R - 2) >> 1][0]);
    e2[R - 2] = e * e;
  }
  e2[R - 1] = T(0);
  RAYEN_LQ_STAMP(3);

  // Gershgorin bracket of lambda_max over the true rows: max diag <= lambda_max <= max(d_i + |e_{i-1}| + |e_i|)
  T lo = dd[0], hi = dd[0] + Lim<T>::sqrt_(e2[0]), dmin = dd[0], emax = T(0), eprev = T(0);
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const bool real = i < r;                       // (pad rows: diagonal -1e18, couplings exactly 0)
    const T enext = (i + 1 < R && i + 1 < r) ? Lim<T>::sqrt_(e2[i]) : T(0);
    if (i > 0) {
      lo = real ? fmax(lo, dd[i]) : lo;
      hi = real ? fmax(hi, dd[i] + eprev + enext) : hi;
      dmin = real ? fmin(dmin, dd[i]) : dmin;
    }
    emax = fmax(emax, enext);
    eprev = enext;
  }
  const T rad = fmax(fmax(hi - dmin, emax), Lim<T>::tiny());
  const T sc = T(1) / rad;
#pragma unroll
  for (int i = 0; i < R; ++i) {
    dd[i] = (i < r) ? (dd[i] - hi) * sc : T(-4);   // true rows in [-1, 0]; pad rows below everything
    e2[i] = (i + 1 < r) ? e2[i] * sc * sc : T(0);
  }
  T lo_n = (lo - hi) * sc, hi_n = T(0);
  for (int it = 0; it < Lim<T>::rounds; ++it) {
    const T w = (hi_n - lo_n) * T(1.0 / 9.0);
    const V2<T> x = {fma_(w, T(2 * sub + 1), lo_n), fma_(w, T(2 * sub + 2), lo_n)};
    V2<T> p0 = splat(T(1)), p1 = splat(dd[0]) - x;
    unsigned ma = Lim<T>::sign_word(p1[0]) >> 31, mb = Lim<T>::sign_word(p1[1]) >> 31;
#pragma unroll
    for (int i = 1; i < R; ++i) {
      const V2<T> t = splat(e2[i - 1]) * p0;
      const V2<T> pn = fma2<T>(splat(dd[i]) - x, p1, -t);
      p0 = p1;
      p1 = pn;
      ma = __builtin_amdgcn_alignbit(ma, Lim<T>::sign_word(p1[0]), 31);   // (ma << 1) | sign
      mb = __builtin_amdgcn_alignbit(mb, Lim<T>::sign_word(p1[1]), 31);
      if ((i & 3) == 0 && i + 1 < R) {
        const int ea = Lim<T>::expo(fmax(fabs(p0[0]), fabs(p1[0]))), eb = Lim<T>::expo(fmax(fabs(p0[1]), fabs(p1[1])));
        p0[0] = Lim<T>::scale2(p0[0], -ea);
        p1[0] = Lim<T>::scale2(p1[0], -ea);
        p0[1] = Lim<T>::scale2(p0[1], -eb);
        p1[1] = Lim<T>::scale2(p1[1], -eb);
      }
    }
    // sign changes along p_0 (> 0) .. p_R = eigenvalues below x; all R of them  ->  lambda_max < x
    const int na = __builtin_popcount(ma ^ (ma >> 1)), nb = __builtin_popcount(mb ^ (mb >> 1));
    T lo_c = lo_n, hi_c = hi_n;
    if (na == R) hi_c = fmin(hi_c, x[0]); else lo_c = fmax(lo_c, x[0]);
    if (nb == R) hi_c = fmin(hi_c, x[1]); else lo_c = fmax(lo_c, x[1]);
    lo_n = qmax(lo_c);
    hi_n = fmax(qmin(hi_c), lo_n);
  }
  return fma_(T(0.5) * (lo_n + hi_n), rad, hi);
}

// S = sum_a v_a G_a, this lane's rows, from the LDS image (pad rows/columns decoupled, far below the spectrum)
template <typename T, int R4>
__device__ __forceinline__ void form_S(V2<T> (&a)[R4][2 * R4], const T* Wf, const T* vs, const int n, const int r,
                                       const int sub) {
  constexpr int R = 4 * R4, H = R / 2;
#pragma unroll
  for (int t = 0; t < R4; ++t)
#pragma unroll
    for (int jj = 0; jj < H; ++jj) {
      a[t][jj][0] = (4 * t + sub == 2 * jj && 2 * jj >= r) ? T(-1e18) : T(0);
      a[t][jj][1] = (4 * t + sub == 2 * jj + 1 && 2 * jj + 1 >= r) ? T(-1e18) : T(0);
    }
  for (int aa = 0; aa < n; ++aa) {
    const V2<T> va = splat(vs[aa]);
    const V2<T>* wa = reinterpret_cast<const V2<T>*>(Wf + (size_t)aa * R * R + sub * R);
#pragma unroll
    for (int t = 0; t < R4; ++t)
#pragma unroll
      for (int jj = 0; jj < H; ++jj) a[t][jj] = fma2<T>(wa[(4 * t * R) / 2 + jj], va, a[t][jj]);
  }
}

// ---- S on the matrix cores (round 4).  form_S above reads every generator entry from LDS once per LANE: 1 MB per
// 64-sample block at config 4, 8000 clocks of the CU's 128 B/clk -- a quarter of the kernel (s_memtime stamps,
// scripts/ubench/lq_stamps.hip), although it is only 500 packed fmas.  The matrix cores broadcast operands in hardware:
// per wave (16 samples) S^T[entry][sample] = sum_a G_a[entry] v_a[sample] is R4*R4 tiles of v_mfma_*_16x16x4 (exact
// fp32 / fp64: an fma chain over a in ascending order, the same bits as form_S), A operands = the generators from an LDS
// image in operand order (one ds_read_b32 per tile and k-step: 77 KB per block instead of 1 MB; read straight from
// global memory by every wave they took as long as the LDS version of form_S -- four waves x 75 loads per CU through
// the vector-memory path at kernel start), B operands = the wave's 16 rows of v.  The
// tile rows are ORDERED so that lane group q = lane >> 4 receives exactly the entries of matrix rows 4t + q -- the
// quad kernel's distribution -- and one trip through a wave-private LDS patch moves them from lane 16 q + s to lane
// 4 s + q (the DPP quad of sample s).
template <typename T> struct Mma16;
template <> struct Mma16<float> {
  typedef float acc_t __attribute__((ext_vector_type(4)));
  __device__ static __forceinline__ acc_t mma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
  // C/D: col = lane & 15, row = 4 (lane >> 4) + reg   ->   tile row m carries (sub = m >> 2, entry 4 tau + (m & 3))
  __host__ __device__ static constexpr int sub_of_row(int m) { return m >> 2; }
  __host__ __device__ static constexpr int reg_of_row(int m) { return m & 3; }
};
template <> struct Mma16<double> {
  typedef double acc_t __attribute__((ext_vector_type(4)));
  __device__ static __forceinline__ acc_t mma(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
  // C/D: col = lane & 15, row = (lane >> 4) + 4 reg   ->   tile row m carries (sub = m & 3, entry 4 tau + (m >> 2))
  __host__ __device__ static constexpr int sub_of_row(int m) { return m & 3; }
  __host__ __device__ static constexpr int reg_of_row(int m) { return m >> 2; }
};

// tiles per trip through a wave's LDS patch (64 lanes x 4 T each): all R4 * R4 of them when four such patches fit next to
// the image (one write phase, one read phase: every trip costs two LDS round trips of latency), else R4 per trip
template <typename T, int R4> constexpr int quad_stage_tiles() { return (R4 * R4 * 64 * 4 * sizeof(T) * 4 <= 104 * 1024) ? R4 * R4 : R4; }

template <typename T, int R4>
__device__ __forceinline__ void form_S_mfma(V2<T> (&a)[R4][2 * R4], const T* wm, const int ks, const T* vt16,
                                            const int LDV, const int n, const int r, T* stage, const int lane) {
  // wm: the A-operand image in LDS, [k-step][tile][lane]; vt16: the wave's 16 rows of v in LDS (row stride LDV)
  constexpr int R = 4 * R4, NT = R4 * R4, CH = quad_stage_tiles<T, R4>();
  typedef typename Mma16<T>::acc_t acc_t;
  acc_t acc[NT];
  sfor<0, NT>([&](auto it) { acc[decltype(it)::value] = acc_t{T(0), T(0), T(0), T(0)}; });
  const int col = lane & 15, kq = lane >> 4;
  const T* vrow = vt16 + col * LDV;
  const T* wl = wm + lane;
  for (int kk = 0; kk < ks; ++kk) {
    T ac[NT];
    sfor<0, NT>([&](auto it) { ac[decltype(it)::value] = wl[decltype(it)::value * 64]; });
    const int aa = 4 * kk + kq;
    const T bc = aa < n ? vrow[aa] : T(0);
    wl += NT * 64;
    sfor<0, NT>([&](auto it) {
      constexpr int tau = decltype(it)::value;
      acc[tau] = Mma16<T>::mma(ac[tau], bc, acc[tau]);
    });
  }
  // lane 16 q + s  ->  lane 4 s + q, CH tiles per trip.  Slot of (q, s): 16 q + ((s + 2 q) & 15) -- eight consecutive
  // writers and eight consecutive readers each touch eight different 16-byte columns.
  const int wq = lane >> 4, ws = lane & 15, rq = lane & 3, rs = lane >> 2;
  acc_t* wslot = reinterpret_cast<acc_t*>(stage) + (16 * wq + ((ws + 2 * wq) & 15));
  const acc_t* rslot = reinterpret_cast<const acc_t*>(stage) + (16 * rq + ((rs + 2 * rq) & 15));
  const int sub = rq;
  sfor<0, (NT + CH - 1) / CH>([&](auto ic) {
    constexpr int c0 = decltype(ic)::value * CH;
    sfor<0, CH>([&](auto iu) {
      constexpr int tau = c0 + decltype(iu)::value;
      if constexpr (tau < NT) wslot[decltype(iu)::value * 64] = acc[tau];
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    sfor<0, CH>([&](auto iu) {
      constexpr int tau = c0 + decltype(iu)::value;
      if constexpr (tau < NT) {
        const acc_t got = rslot[decltype(iu)::value * 64];
        sfor<0, 4>([&](auto ii) {
          constexpr int idx = 4 * tau + decltype(ii)::value, t = idx / R, c = idx % R;
          // (pad rows / columns: decoupled and far below the spectrum, as form_S leaves them; only an entry that can be
          // on some lane's diagonal needs the select)
          if constexpr (c - 4 * t >= 0 && c - 4 * t < 4)
            a[t][c >> 1][c & 1] = (4 * t + sub == c && c >= r) ? T(-1e18) : got[decltype(ii)::value];
          else
            a[t][c >> 1][c & 1] = got[decltype(ii)::value];
        });
      }
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  });
}

// the constant image into LDS: 16-byte pieces, all of a thread's loads in flight before its first store
template <typename T>
__device__ __forceinline__ void fill_image(T* img, const T* image, const int64_t elems, const int tid) {
  typedef int v4i __attribute__((ext_vector_type(4)));
  constexpr int PER = 16 / sizeof(T);
  const int64_t n16 = elems / PER;
  const v4i* src = reinterpret_cast<const v4i*>(image);
  v4i* dst = reinterpret_cast<v4i*>(img);
  for (int64_t base = 0; base < n16; base += 256 * 8) {
    v4i tmp[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t i = base + tid + 256 * u;
      if (i < n16) tmp[u] = src[i];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t i = base + tid + 256 * u;
      if (i < n16) dst[i] = tmp[u];
    }
  }
  for (int64_t i = n16 * PER + tid; i < elems; i += 256) img[i] = image[i];
}

// MF: S on the matrix cores (form_S_mfma) -- `image` is then [Wm | Wlin | N | y0] with the generators in A-operand
// order (LmiQuadImage::wm, `ks` K-steps) instead of [Wf | ...].
template <typename T, int R4, bool MF = false>
__global__ __launch_bounds__(256) void lmi_quad_kernel(
    const T* __restrict__ image, const int32_t* __restrict__ lin_id, int r, int n, int k, int m, int identity,
    int lmi_seg, int64_t elems, const T* __restrict__ v, int64_t B, int64_t ldv, T* __restrict__ y, int64_t ldy,
    T* __restrict__ kappa_out, int32_t* __restrict__ active_out, int32_t* __restrict__ nan_flag, int ks = 0) {
  constexpr int R = 4 * R4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* img = reinterpret_cast<T*>(smem_raw);               // the constant image (Wf or Wm | Wlin | N | y0)
  const T* Wf = img;                                      // [n][R][R]; with MF: Wm [ks][R4 R4][64], the A operands
  const T* Wlin = Wf + (MF ? (size_t)ks * R4 * R4 * 64 : (size_t)n * R * R);      // [m][n]
  const T* Nmat = Wlin + (size_t)m * n;                   // [k][n] (absent when identity)
  const T* y0 = Nmat + (identity ? 0 : (size_t)k * n);    // [k]
  T* vt = img + ((elems + 3) & ~int64_t(3));              // [64][n + 1]
  const int LDV = n + 1;

  const int tid = threadIdx.x;
  const int sl = tid >> 2, sub = tid & 3;
  const int64_t b0 = (int64_t)blockIdx.x * 64;
  const int nb = (int)((B - b0) < 64 ? (B - b0) : 64);
  RAYEN_LQ_STAMP(0);
  // rows of v and the image in ONE round trip to memory: every load is issued before the first LDS store waits for it
  // (fill_image first and the rows of v behind its stores were two dependent round trips: 2 of the kernel's 15 us)
  if constexpr (MF) {
    // (thread tid moves columns (tid & 3) + 4 u of sample tid >> 2: n <= 64 is at most 16 of them, no division)
    T vreg[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int j = sub + 4 * u;
      vreg[u] = (j < n && sl < nb) ? v[(b0 + sl) * ldv + j] : T(0);
    }
    fill_image<T>(img, image, elems, tid);
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int j = sub + 4 * u;
      if (j < n) vt[sl * LDV + j] = vreg[u];
    }
  } else {
    fill_image<T>(img, image, elems, tid);
    for (int idx = tid; idx < 64 * n; idx += 256) {
      const int bl = idx / n, j = idx - bl * n;
      vt[bl * LDV + j] = bl < nb ? v[(b0 + bl) * ldv + j] : T(0);
    }
  }
  __syncthreads();
  RAYEN_LQ_STAMP(1);
  const T* vs = vt + sl * LDV;
  const bool live = sl < nb;

  V2<T> a[R4][R / 2];
  if constexpr (MF) {
    // every wave forms the S of its own 16 samples on the matrix cores; the LDS image starts with the A operands
    T* stage = vt + (((size_t)64 * LDV + 3) & ~size_t(3)) + (size_t)(tid >> 6) * (quad_stage_tiles<T, R4>() * 64 * 4);
    form_S_mfma<T, R4>(a, img, ks, vt + 16 * (tid >> 6) * LDV, LDV, n, r, stage, tid & 63);
  } else {
    form_S<T, R4>(a, Wf, vs, n, r, sub);
  }
  RAYEN_LQ_STAMP(2);

  // ---- linear rows, split over the quad
  T kap = T(0);
  int aseg = -1, arow = 0;
  for (int rho = sub; rho < m; rho += 4) {
    T acc = T(0);
    for (int aa = 0; aa < n; ++aa) acc = fma_(Wlin[rho * n + aa], vs[aa], acc);
    if (acc > kap) { kap = acc; aseg = lin_id[2 * rho]; arow = lin_id[2 * rho + 1]; }
  }
  if (m > 0) {
    // quad arg-max; ties go to the lower lane so that all four agree
    int who = sub;
    {
      const T ok = dpp_<0xB1>(kap);
      const int ow = dpp_<0xB1>(who), os = dpp_<0xB1>(aseg), orow = dpp_<0xB1>(arow);
      if (ok > kap || (ok == kap && ow < who)) { kap = ok; who = ow; aseg = os; arow = orow; }
    }
    {
      const T ok = dpp_<0x4E>(kap);
      const int ow = dpp_<0x4E>(who), os = dpp_<0x4E>(aseg), orow = dpp_<0x4E>(arow);
      if (ok > kap || (ok == kap && ow < who)) { kap = ok; who = ow; aseg = os; arow = orow; }
    }
  }

  const T lam = lambda_max_quad<T, R4>(a, r, sub);
  RAYEN_LQ_STAMP(4);
  if (lam > kap) { kap = lam; aseg = lmi_seg; arow = 0; }

  const T scale = T(1) / fmax(T(1), kap);
  if (live && sub == 0) {
    if (kappa_out) kappa_out[b0 + sl] = kap;
    if (active_out) { active_out[2 * (b0 + sl)] = aseg; active_out[2 * (b0 + sl) + 1] = arow; }
  }
  bool bad = false;
  if (live) {
    T* yrow = y + (b0 + sl) * ldy;
    for (int i = sub; i < k; i += 4) {
      T val;
      if (identity) {
        val = fma_(vs[i], scale, y0[i]);
      } else {
        T acc = T(0);
        for (int aa = 0; aa < n; ++aa) acc = fma_(Nmat[i * n + aa], vs[aa], acc);
        val = fma_(acc, scale, y0[i]);
      }
      bad |= (val != val);
      yrow[i] = val;
    }
  }
  if (nan_flag && bad) atomicOr(nan_flag, 1);
  RAYEN_LQ_STAMP(5);
}

// ---------------------------------------------------------------------------------------------
// Backward with the same four-lanes-per-sample layout:
//   grad_v = s t - [kappa > 1] s^2 (t . v) grad kappa(v),   t = NA_E' g,   s = 1 / max(1, kappa)
// When the LMI is the active constraint, grad kappa_b = x' G_b x for the unit eigenvector x of lambda_max
// (what autograd gives for eigvalsh + max, rayen/constraint_module.py:424-425).  The quad repeats the
// forward's S and Householder sweep but keeps the reflectors (in the columns they annihilate), gets the
// eigenvector z of the tridiagonal matrix by inverse iteration on (lambda + shift) I - T -- positive definite,
// so a plain LDL' without pivoting, O(R) per solve, done redundantly by the four lanes --, maps it back
// through the reflectors, x = H_0 H_1 ... z (quad sums for the dot products), and contracts x with the
// generators straight from the LDS image.
// ---------------------------------------------------------------------------------------------
template <typename T, int R4>
__device__ __forceinline__ void top_eigenvector_quad(V2<T> (&a)[R4][2 * R4], const T lam, T (&xr)[R4], const int sub) {
  constexpr int R = 4 * R4;
  T dd[R], eb[2 * R];  // eb[0..R): signed sub-diagonal, eb[R..2R): reflector scales
  sfor<0, R - 2>([&](auto ic) {
    constexpr int c = decltype(ic)::value;
    hh_step<T, R4, c, true>(a, dd, eb, sub);
  });
  dd[R - 2] = qb<((R - 2) & 3)>(a[(R - 2) >> 2][(R - 2) >> 1][0]);
  dd[R - 1] = qb<((R - 1) & 3)>(a[(R - 1) >> 2][(R - 1) >> 1][1]);
  eb[R - 2] = qb<((R - 1) & 3)>(a[(R - 1) >> 2][(R - 2) >> 1][0]);
  eb[R - 1] = T(0);

  // M = (lam + shift) I - T, tridiagonal and positive definite; padding rows (d = -1e18) are decoupled
  T scale = fabs(lam);
#pragma unroll
  for (int i = 0; i < R; ++i) scale = fmax(scale, dd[i] > T(-1e17) ? fabs(dd[i]) : T(0));
  const T shift = (sizeof(T) == 4 ? T(2e-4) : T(1e-9)) * fmax(scale, Lim<T>::tiny());
  T Dg[R], l[R];  // M = L D L', unit lower bidiagonal L with sub-diagonal l[i] (row i), i >= 1
  Dg[0] = fmax(lam + shift - dd[0], shift * T(1e-3));
  l[0] = T(0);
#pragma unroll
  for (int i = 1; i < R; ++i) {
    l[i] = -eb[i - 1] / Dg[i - 1];
    Dg[i] = fmax(lam + shift - dd[i] + l[i] * eb[i - 1], shift * T(1e-3));
  }
  T z[R];
#pragma unroll
  for (int i = 0; i < R; ++i) z[i] = T(1) + T(0.01) * T(i);  // not orthogonal to anything special
#pragma unroll
  for (int it = 0; it < 3; ++it) {
#pragma unroll
    for (int i = 1; i < R; ++i) z[i] = fma_(-l[i], z[i - 1], z[i]);   // L y = b
    T nrm = T(0);
    z[R - 1] = z[R - 1] / Dg[R - 1];
    nrm = z[R - 1] * z[R - 1];
#pragma unroll
    for (int i = R - 2; i >= 0; --i) {                                 // D L' z = y
      z[i] = fma_(-l[i + 1], z[i + 1], z[i] / Dg[i]);
      nrm = fma_(z[i], z[i], nrm);
    }
    const T inv = T(1) / sqrt(fmax(nrm, Lim<T>::tiny()));
#pragma unroll
    for (int i = 0; i < R; ++i) z[i] *= inv;
  }
  // x = H_0 H_1 ... H_{R-3} z, this lane's rows
  sfor<0, R4>([&](auto it) {
    constexpr int t = decltype(it)::value;
    xr[t] = T(0);
    sfor<0, 4>([&](auto is) {
      constexpr int q = decltype(is)::value;
      if (sub == q) xr[t] = z[4 * t + q];
    });
  });
  sfor<0, R - 2>([&](auto ic) {
    constexpr int c = R - 3 - decltype(ic)::value;
    constexpr int I1 = c + 1, T0 = I1 / 4;
    T dot = T(0);
    sfor<T0, R4>([&](auto it) {
      constexpr int t = decltype(it)::value;
      const int i = 4 * t + sub;
      const T u = (i == I1) ? T(1) : ((i >= c + 2) ? a[t][c >> 1][c & 1] : T(0));
      dot = fma_(u, xr[t], dot);
    });
    dot = qsum(dot) * eb[R + c];
    sfor<T0, R4>([&](auto it) {
      constexpr int t = decltype(it)::value;
      const int i = 4 * t + sub;
      const T u = (i == I1) ? T(1) : ((i >= c + 2) ? a[t][c >> 1][c & 1] : T(0));
      xr[t] = fma_(-dot, u, xr[t]);
    });
  });
}

template <typename T, int R4>
__global__ __launch_bounds__(256) void lmi_quad_bwd_kernel(
    const T* __restrict__ image, const T* __restrict__ wrow, int r, int n, int k, int m, int identity, int lmi_seg,
    int64_t elems, const T* __restrict__ v, int64_t B, int64_t ldv, const T* __restrict__ kappa,
    const int32_t* __restrict__ active, const T* __restrict__ gy, int64_t ldg, T* __restrict__ gv, int64_t ldgv) {
  constexpr int R = 4 * R4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* img = reinterpret_cast<T*>(smem_raw);
  const T* Wf = img;                                      // [n][R][R]
  const T* Nmat = Wf + (size_t)n * R * R + (size_t)m * n; // [k][n] (absent when identity)
  T* vt = img + ((elems + 3) & ~int64_t(3));              // [64][n + 1]
  const int LDV = n + 1, LDG = k + 1;
  T* tt = vt + 64 * LDV;                                  // [64][n + 1]  t = NA_E' g
  T* ut = tt + 64 * LDV;                                  // [64][n + 1]  grad kappa
  T* gt = ut + 64 * LDV;                                  // [64][k + 1]  grad_y

  const int tid = threadIdx.x;
  const int sl = tid >> 2, sub = tid & 3;
  const int64_t b0 = (int64_t)blockIdx.x * 64;
  const int nb = (int)((B - b0) < 64 ? (B - b0) : 64);
  RAYEN_LQ_STAMP(0);
  fill_image<T>(img, image, elems, tid);
  for (int idx = tid; idx < 64 * n; idx += 256) {
    const int bl = idx / n, j = idx - bl * n;
    vt[bl * LDV + j] = bl < nb ? v[(b0 + bl) * ldv + j] : T(0);
    ut[bl * LDV + j] = T(0);
  }
  for (int idx = tid; idx < 64 * k; idx += 256) {
    const int bl = idx / k, j = idx - bl * k;
    gt[bl * LDG + j] = bl < nb ? gy[(b0 + bl) * ldg + j] : T(0);
  }
  __syncthreads();
  const T* vs = vt + sl * LDV;
  const T* gs = gt + sl * LDG;
  const bool live = sl < nb;

  for (int a = sub; a < n; a += 4) {
    T acc = T(0);
    if (identity) {
      acc = gs[a];
    } else {
      for (int i = 0; i < k; ++i) acc = fma_(Nmat[i * n + a], gs[i], acc);
    }
    tt[sl * LDV + a] = acc;
  }
  __syncthreads();
  const T* ts = tt + sl * LDV;
  T tv = T(0);
  for (int a = 0; a < n; ++a) tv = fma_(ts[a], vs[a], tv);

  const T kap = live ? kappa[b0 + sl] : T(0);
  const int aseg = live ? active[2 * (b0 + sl)] : -1;
  const int arow = live ? active[2 * (b0 + sl) + 1] : 0;
  const bool clipped = live && kap > T(1) && aseg >= 0;
  const T sc = T(1) / fmax(T(1), kap);

  if (clipped && aseg == lmi_seg) {  // the same for the four lanes of a quad
    V2<T> a[R4][R / 2];
    form_S<T, R4>(a, Wf, vs, n, r, sub);
    T xr[R4];
    top_eigenvector_quad<T, R4>(a, kap, xr, sub);
    T xf[R];
    sfor<0, R>([&](auto ij) {
      constexpr int j = decltype(ij)::value;
      xf[j] = qb<(j & 3)>(xr[j >> 2]);
    });
    for (int bb = 0; bb < n; ++bb) {  // grad kappa_b = x' G_b x, this lane's rows first
      const T* wb = Wf + (size_t)bb * R * R + sub * R;
      T part = T(0);
#pragma unroll
      for (int t = 0; t < R4; ++t) {
        T rowdot = T(0);
#pragma unroll
        for (int j = 0; j < R; ++j) rowdot = fma_(wb[4 * t * R + j], xf[j], rowdot);
        part = fma_(xr[t], rowdot, part);
      }
      part = qsum(part);
      if (sub == 0) ut[sl * LDV + bb] = part;
    }
  } else if (clipped) {              // a linear row
    for (int bb = sub; bb < n; bb += 4) ut[sl * LDV + bb] = wrow[(size_t)arow * n + bb];
  }
  __syncthreads();
  if (live) {
    const T coef = clipped ? sc * sc * tv : T(0);
    for (int bb = sub; bb < n; bb += 4) gv[(b0 + sl) * ldgv + bb] = fma_(sc, ts[bb], -coef * ut[sl * LDV + bb]);
  }
}

// ---------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------

template <typename T>
inline int quad_size_class(int r) {  // padded size R (multiple of 4) the kernels are instantiated for
  for (int R : {8, 16, 20, 24, 32})
    if (r <= R) return (R == 32 && sizeof(T) == 8) ? 0 : R;  // fp64: a quarter of a 32 x 32 matrix spills
  return 0;
}

template <typename T>
size_t quad_lds_bytes(const RayenPack* p, int R, int m) {
  const size_t elems = (size_t)p->n * R * R + (size_t)m * p->n + (p->out_identity ? 0 : (size_t)p->k * p->n) + p->k;
  return sizeof(T) * (((elems + 3) & ~size_t(3)) + 64 * (size_t)(p->n + 1));
}

template <typename T>
bool lmi_quad_eligible_t(const RayenPack* p) {
  int n_lmi = 0, r = 0, m = 0;
  for (const RayenSegment& g : p->segs) {
    if (g.type == RAYEN_SEG_LMI) { ++n_lmi; r = g.dim; }
    else if (g.type == RAYEN_SEG_LIN) m += g.nrows;
    else return false;
  }
  if (n_lmi != 1 || r < 2) return false;
  const int R = quad_size_class<T>(r);
  return R != 0 && p->n <= 64 && p->k <= 256 && quad_lds_bytes<T>(p, R, m) <= 64 * 1024;
}

template <typename T>
int lmi_quad_build_t(const RayenPack* p, LmiQuadImage** out, int64_t* bytes) {
  LmiQuadImage* img = new LmiQuadImage();
  const int n = p->n, k = p->k;
  const RayenSegment* lmi = nullptr;
  std::vector<int32_t> ids;
  std::vector<const double*> lin_rows;
  for (size_t s = 0; s < p->segs.size(); ++s) {
    const RayenSegment& g = p->segs[s];
    if (g.type == RAYEN_SEG_LMI) { lmi = &g; img->lmi_seg = (int)s; }
    if (g.type == RAYEN_SEG_LIN)
      for (int rr = 0; rr < g.nrows; ++rr) {
        lin_rows.push_back(p->W.data() + (size_t)(g.row0 + rr) * n);
        ids.push_back((int32_t)s);
        ids.push_back(g.row0 + rr);
      }
  }
  const int r = lmi->dim, R = quad_size_class<T>(r), m = (int)lin_rows.size();
  img->r = r; img->R = R; img->n = n; img->k = k; img->m = m; img->identity = p->out_identity;
  std::vector<T> host((size_t)n * R * R + (size_t)m * n + (p->out_identity ? 0 : (size_t)k * n) + k, T(0));
  for (int a = 0; a < n; ++a)
    for (int i = 0; i < r; ++i)
      for (int j = 0; j < r; ++j) {
        const int hi = i > j ? i : j, lo = i > j ? j : i;
        host[((size_t)a * R + i) * R + j] = (T)p->W[(size_t)(lmi->row0 + hi * (hi + 1) / 2 + lo) * n + a];
      }
  size_t off = (size_t)n * R * R;
  for (int rr = 0; rr < m; ++rr)
    for (int a = 0; a < n; ++a) host[off + (size_t)rr * n + a] = (T)lin_rows[rr][a];
  off += (size_t)m * n;
  if (!p->out_identity) {
    for (size_t i = 0; i < (size_t)k * n; ++i) host[off + i] = (T)p->NA_E[i];
    off += (size_t)k * n;
  }
  for (int i = 0; i < k; ++i) host[off + i] = (T)p->y0[i];
  if (ids.empty()) ids.assign(2, 0);
  img->elems = (int64_t)host.size();
  // the generators as 16x16x4 A operands: [k-step][tile][lane]; lane (m = lane & 15, kq = lane >> 4) of tile tau holds
  // G_a[4 t + sub][c] with a = 4 kk + kq and (sub, entry 4 tau + reg = t R + c) from the instruction's C/D row map
  const int R4 = R / 4, NT = R4 * R4, ks = (n + 3) / 4;
  std::vector<T> wm((size_t)ks * NT * 64, T(0));
  for (int kk = 0; kk < ks; ++kk)
    for (int tau = 0; tau < NT; ++tau)
      for (int l = 0; l < 64; ++l) {
        const int mrow = l & 15, a = 4 * kk + (l >> 4);
        const int sub = Mma16<T>::sub_of_row(mrow), idx = 4 * tau + Mma16<T>::reg_of_row(mrow);
        const int row = 4 * (idx / R) + sub, c = idx % R;
        if (a < n && row < r && c < r) wm[((size_t)kk * NT + tau) * 64 + l] = host[((size_t)a * R + row) * R + c];
      }
  img->ks = ks;
  wm.insert(wm.end(), host.begin() + (size_t)n * R * R, host.end());     // ... | Wlin | N | y0 behind the operands
  std::vector<T> wr((size_t)(p->n_rows > 0 ? p->n_rows : 1) * n, T(0));
  for (size_t i = 0; i < (size_t)p->n_rows * n; ++i) wr[i] = (T)p->W[i];
  const bool ok = hipMalloc(&img->data, host.size() * sizeof(T)) == hipSuccess &&
                  hipMemcpy(img->data, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice) == hipSuccess &&
                  hipMalloc(&img->wrow, wr.size() * sizeof(T)) == hipSuccess &&
                  hipMemcpy(img->wrow, wr.data(), wr.size() * sizeof(T), hipMemcpyHostToDevice) == hipSuccess &&
                  hipMalloc(&img->wm, wm.size() * sizeof(T)) == hipSuccess &&
                  hipMemcpy(img->wm, wm.data(), wm.size() * sizeof(T), hipMemcpyHostToDevice) == hipSuccess &&
                  hipMalloc(&img->lin_id, ids.size() * sizeof(int32_t)) == hipSuccess &&
                  hipMemcpy(img->lin_id, ids.data(), ids.size() * sizeof(int32_t), hipMemcpyHostToDevice) == hipSuccess;
  if (!ok) {
    if (img->data) (void)hipFree(img->data);
    if (img->wrow) (void)hipFree(img->wrow);
    if (img->wm) (void)hipFree(img->wm);
    if (img->lin_id) (void)hipFree(img->lin_id);
    delete img;
    return RAYEN_E_ALLOC;
  }
  img->bytes = (int64_t)((host.size() + wr.size() + wm.size()) * sizeof(T) + ids.size() * sizeof(int32_t));
  *bytes = img->bytes;
  *out = img;
  return RAYEN_OK;
}

// LDS of the matrix-core instance: [Wm | Wlin | N | y0] + the rows of v + four wave-private transposition patches
template <typename T>
size_t quad_mf_lds_bytes(const RayenPack* p, int R, int m) {
  const size_t ks = (size_t)(p->n + 3) / 4;
  const size_t elems = ks * (R / 4) * (R / 4) * 64 + (size_t)m * p->n + (p->out_identity ? 0 : (size_t)p->k * p->n) + p->k;
  const int R4 = R / 4;
  const size_t tiles = ((size_t)R4 * R4 * 64 * 4 * sizeof(T) * 4 <= 104 * 1024) ? (size_t)R4 * R4 : (size_t)R4;   // quad_stage_tiles
  return sizeof(T) * (((elems + 3) & ~size_t(3)) + ((64 * (size_t)(p->n + 1) + 3) & ~size_t(3)) + 4 * tiles * 64 * 4);
}

inline bool quad_mfma_enabled() {     // RAYEN_LMI_QUAD_MFMA=0: the round-3 kernel (S from the LDS image), for A/B runs
  static const bool on = [] { const char* e = getenv("RAYEN_LMI_QUAD_MFMA"); return !(e && e[0] == '0'); }();
  return on;
}

template <typename T, int R4>
int launch_quad(const RayenPack* p, const LmiQuadImage* img, const T* v, int64_t B, int64_t ldv, T* y, int64_t ldy,
                T* kappa, int32_t* active, int32_t* nan_flag, hipStream_t stream) {
  const int64_t grid = (B + 63) / 64;
  if (img->wm != nullptr && quad_mfma_enabled() && quad_mf_lds_bytes<T>(p, img->R, img->m) <= 150 * 1024) {
    const size_t lds = quad_mf_lds_bytes<T>(p, img->R, img->m);
    auto kern = lmi_quad_kernel<T, R4, true>;
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) !=
            hipSuccess)
      return RAYEN_E_LAUNCH;
    const int64_t gen = (int64_t)img->n * img->R * img->R;
    const int64_t ops = (int64_t)img->ks * R4 * R4 * 64;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, stream, static_cast<const T*>(img->wm), img->lin_id,
                       img->r, img->n, img->k, img->m, img->identity, img->lmi_seg, img->elems - gen + ops, v, B, ldv, y,
                       ldy, kappa, active, nan_flag, img->ks);
    return hipGetLastError() == hipSuccess ? RAYEN_OK : RAYEN_E_LAUNCH;
  }
  const size_t lds = quad_lds_bytes<T>(p, img->R, img->m);
  auto kern = lmi_quad_kernel<T, R4, false>;
  if (lds > 48 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) !=
          hipSuccess)
    return RAYEN_E_LAUNCH;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, stream, static_cast<const T*>(img->data), img->lin_id,
                     img->r, img->n, img->k, img->m, img->identity, img->lmi_seg, img->elems, v, B, ldv, y, ldy, kappa,
                     active, nan_flag, 0);
  return hipGetLastError() == hipSuccess ? RAYEN_OK : RAYEN_E_LAUNCH;
}

template <typename T>
int lmi_quad_forward_t(const RayenPack* p, const LmiQuadImage* img, const T* v, int64_t B, int64_t ldv, T* y,
                       int64_t ldy, T* kappa, int32_t* active, int32_t* nan_flag, hipStream_t stream) {
  if (B == 0) return RAYEN_OK;
  switch (img->R) {
    case 8: return launch_quad<T, 2>(p, img, v, B, ldv, y, ldy, kappa, active, nan_flag, stream);
    case 16: return launch_quad<T, 4>(p, img, v, B, ldv, y, ldy, kappa, active, nan_flag, stream);
    case 20: return launch_quad<T, 5>(p, img, v, B, ldv, y, ldy, kappa, active, nan_flag, stream);
    case 24: return launch_quad<T, 6>(p, img, v, B, ldv, y, ldy, kappa, active, nan_flag, stream);
    case 32:
      if constexpr (sizeof(T) == 4) return launch_quad<T, 8>(p, img, v, B, ldv, y, ldy, kappa, active, nan_flag, stream);
      return RAYEN_E_UNSUPPORTED;
    default: return RAYEN_E_UNSUPPORTED;
  }
}

template <typename T>
size_t quad_bwd_lds_bytes(const RayenPack* p, int R, int m) {
  const size_t elems = (size_t)p->n * R * R + (size_t)m * p->n + (p->out_identity ? 0 : (size_t)p->k * p->n) + p->k;
  return sizeof(T) * (((elems + 3) & ~size_t(3)) + 64 * (size_t)(3 * (p->n + 1) + p->k + 1));
}

template <typename T, int R4>
int launch_quad_bwd(const RayenPack* p, const LmiQuadImage* img, const T* v, int64_t B, int64_t ldv, const T* kappa,
                    const int32_t* active, const T* gy, int64_t ldg, T* gv, int64_t ldgv, hipStream_t stream) {
  const size_t lds = quad_bwd_lds_bytes<T>(p, img->R, img->m);
  if (lds > 128 * 1024) return RAYEN_E_UNSUPPORTED;
  auto kern = lmi_quad_bwd_kernel<T, R4>;
  if (lds > 48 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) !=
          hipSuccess)
    return RAYEN_E_LAUNCH;
  const int64_t grid = (B + 63) / 64;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, stream, static_cast<const T*>(img->data),
                     static_cast<const T*>(img->wrow), img->r, img->n, img->k, img->m, img->identity, img->lmi_seg,
                     img->elems, v, B, ldv, kappa, active, gy, ldg, gv, ldgv);
  return hipGetLastError() == hipSuccess ? RAYEN_OK : RAYEN_E_LAUNCH;
}

// fp64 keeps a quarter of the matrix, the reflectors and the tridiagonal solve in registers only up to 16 x 16
template <typename T>
bool lmi_quad_bwd_serves(const RayenPack* p, const LmiQuadImage* img) {
  if (sizeof(T) == 8 && img->R > 24) return false;
  return quad_bwd_lds_bytes<T>(p, img->R, img->m) <= 128 * 1024;
}

template <typename T>
int lmi_quad_backward_t(const RayenPack* p, const LmiQuadImage* img, const T* v, int64_t B, int64_t ldv,
                        const T* kappa, const int32_t* active, const T* gy, int64_t ldg, T* gv, int64_t ldgv,
                        hipStream_t stream) {
  if (B == 0) return RAYEN_OK;
  switch (img->R) {
    case 8: return launch_quad_bwd<T, 2>(p, img, v, B, ldv, kappa, active, gy, ldg, gv, ldgv, stream);
    case 16: return launch_quad_bwd<T, 4>(p, img, v, B, ldv, kappa, active, gy, ldg, gv, ldgv, stream);
    case 20: return launch_quad_bwd<T, 5>(p, img, v, B, ldv, kappa, active, gy, ldg, gv, ldgv, stream);
    case 24: return launch_quad_bwd<T, 6>(p, img, v, B, ldv, kappa, active, gy, ldg, gv, ldgv, stream);
    case 32:
      if constexpr (sizeof(T) == 4) return launch_quad_bwd<T, 8>(p, img, v, B, ldv, kappa, active, gy, ldg, gv, ldgv, stream);
      return RAYEN_E_UNSUPPORTED;
    default: return RAYEN_E_UNSUPPORTED;
  }
}

}  // namespace lq
}  // namespace rayen
