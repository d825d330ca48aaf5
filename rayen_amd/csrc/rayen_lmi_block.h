// One WORKGROUP (512 or 1024 threads: four or two per row of the matrix) per sample for sets = [linear rows] + one LMI: the matrices between what the
// four-lanes-per-sample kernel holds (rayen_lmi_quad.h: 32 x 32) and the end of the reference's own sweep
// (examples/scripts/time_analysis.py:157-160: 300 x 300; rayen/constraint_module.py:401-449 handles any r).
//
// rayen_lmi_wave.h gives such a matrix to ONE wave: the full r x r storage lives in that wave's LDS (r <= ~190 in fp32), one
// wave is all a compute unit holds of them, and a single wave reads LDS at a fraction of the unit's bandwidth (r = 180,
// B = 2 000: 33 ms = 0.5 TFLOP/s).  Here the symmetric matrix S(v) = sum_a v_a G_a is stored ONCE -- its lower triangle, packed
// row-major, entry (i, j <= i) at i (i + 1) / 2 + j: r <= 281 in fp32, 197 in fp64 -- and eight waves work on it:
//   * Householder tridiagonalisation, thread t owns row i0 + t of the live block: p = A v with the column index running for a
//     whole wave at once -- below the diagonal a lane reads its own row (the row starts i (i + 1) / 2 are distinct modulo 32
//     for 32 consecutive i: no bank conflict), above it the mirrored entry of row j (consecutive words), v_j is a broadcast;
//     the rank-2 update touches the lower triangle only; four workgroup barriers per column;
//   * lambda_max of the tridiagonal by MULTI-section on the Sturm count, one shift per thread: 513 x per round (four rounds in
//     fp32, seven in fp64).
// Same formulas as rayen_lmi_wave.h (unnormalised reflector v = x - alpha e_1, tau = 1 / (sigma - x0 alpha); pivots of
// T - sigma I with the same floor), so the two kernels agree to rounding.  Forward only: the backward of these sets stays with
// rayen_lmi_wave.h where it fits.  Uses that kernel's device image (LmiWaveImage).
#pragma once

#include <cstdlib>

#include "rayen_lmi_wave.h"

namespace rayen {
namespace lb {

#ifdef RAYEN_LB_PROFILE
// developer build (scripts/ubench/tu_variant.sh rayen_lmi_block prof -DRAYEN_LB_PROFILE): cycles of workgroup 0's thread 0
// between the barriers of the reduction -- [0] column + sigma, [1] matvec, [2] w, [3] update, [4] S(v), [5] Sturm, [6] all
__device__ unsigned long long g_lb_prof[8];
#define LB_TICK(slot) do { if (threadIdx.x == 0 && blockIdx.x == 0) { const unsigned long long now_ = clock64(); \
    g_lb_prof[slot] += now_ - lb_t_; lb_t_ = now_; } } while (0)
#define LB_TICK_DECL unsigned long long lb_t_ = clock64()
#else
#define LB_TICK(slot) do { } while (0)
#define LB_TICK_DECL do { } while (0)
#endif

constexpr int kWaves = 16;                   // slots of the reduction scratch (the largest workgroup's waves)
#ifndef RAYEN_LB_SMALL_WPE
#define RAYEN_LB_SMALL_WPE 4
#endif
constexpr int kSmallWavesPerEu = RAYEN_LB_SMALL_WPE;    // waves per SIMD the 128- / 256-thread instances are compiled for
constexpr size_t kLdsMax = 160 * 1024;

// LDS of a workgroup (units of T): A[P] | dd[r] | ee[r] | vv[r] | ww[r] | red[3][2 kWaves] | vs[n]
__host__ __device__ inline size_t lds_elems(int r, int n) {
  return (size_t)r * (r + 1) / 2 + 4 * (size_t)r + 6 * kWaves + (size_t)n + 8;
}

// rounds of multi-section with one shift per thread: (threads + 1)^rounds >= 2^bits
constexpr int sturm_rounds(int threads, int bits) {
  int rounds = 0;
  double span = 1.0, want = 1.0;
  for (int b = 0; b < bits; ++b) want *= 2.0;
  while (span < want) { span *= (double)(threads + 1); ++rounds; }
  return rounds;
}

// e^2 / q of the Sturm recurrence: the count needs the SIGN of the next pivot, and the bracket is padded by 1e-6 of the
// matrix scale -- the hardware reciprocal (1 ulp) does in fp32 (two instructions in the dependent chain instead of ten)
__device__ __forceinline__ float quot(float e2, float q) { return e2 * __builtin_amdgcn_rcpf(q); }
__device__ __forceinline__ double quot(double e2, double q) { return e2 / q; }

// ... with the first hc columns in registers (head_phase): the packed triangle of the other r - hc rows / columns
__host__ __device__ inline size_t lds_elems_head(int r, int n, int hc, bool bwd) {
  const size_t rp = (size_t)(r - hc);
  return rp * (rp + 1) / 2 + (bwd ? 5 : 4) * (size_t)r + 6 * kWaves + (bwd ? 2 : 1) * (size_t)n + 8;
}

// sum over the workgroup, returned to every thread; ONE barrier.  `red` (kWaves slots) must not be written again before the
// next barrier: the callers alternate between three slot sets.
template <typename T, int NW>
__device__ __forceinline__ T bsum(T x, T* red, const int tid) {
  x = lw::wsum(x);
  if ((tid & 63) == 0) red[tid >> 6] = x;
  __syncthreads();
  T s = red[0];
#pragma unroll
  for (int w = 1; w < NW; ++w) s += red[w];
  return s;
}

// S(v) = sum_a v_a G_a, lower triangle packed (no barrier: the caller synchronises)
template <typename T, int NTH>
__device__ __forceinline__ void form_S(T* A, const T* __restrict__ gt, const T* vs, int n, int P, int Pp, const int tid) {
  // sixteen generators' words in flight per thread (four were latency-bound: 0.57 of 3.56 M cycles per sample at r = 250, k = 100)
  for (int idx = tid; idx < P; idx += NTH) {
    T p0 = T(0), p1 = T(0), p2 = T(0), p3 = T(0);
    const T* col = gt + idx;
    int a = 0;
    for (; a + 15 < n; a += 16) {
      T g[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) g[u] = col[(size_t)(a + u) * Pp];
#pragma unroll
      for (int u = 0; u < 16; u += 4) {
        p0 = fma(vs[a + u + 0], g[u + 0], p0);
        p1 = fma(vs[a + u + 1], g[u + 1], p1);
        p2 = fma(vs[a + u + 2], g[u + 2], p2);
        p3 = fma(vs[a + u + 3], g[u + 3], p3);
      }
    }
    for (; a + 3 < n; a += 4) {
      p0 = fma(vs[a + 0], col[(size_t)(a + 0) * Pp], p0);
      p1 = fma(vs[a + 1], col[(size_t)(a + 1) * Pp], p1);
      p2 = fma(vs[a + 2], col[(size_t)(a + 2) * Pp], p2);
      p3 = fma(vs[a + 3], col[(size_t)(a + 3) * Pp], p3);
    }
    for (; a < n; ++a) p0 = fma(vs[a], col[(size_t)a * Pp], p0);
    A[idx] = (p0 + p1) + (p2 + p3);
  }
}

// Lane layout of a wave in one column step of the reduction: W = 64 / SPLIT consecutive rows, row `lane % W`; the SPLIT lanes
// `lane / W` of a row take its columns i0 + part, + SPLIT, ...  (with two lanes per row a half-wave reads 32 consecutive
// words, or 32 row starts, at once).  SPLIT is chosen PER COLUMN: as the live block shrinks, more lanes share a row
// (tridiagonalise), so a lane's share of the row -- the length of the dependent chain between two barriers -- stays short.
//
// q_i = sum_{j = i0}^{r-1} A(i, j) x_j for row ic of the live block (the part of it this lane's columns hold).  The column
// index runs for the whole wave at once, in three stretches that need no per-lane address arithmetic beyond an add:
//   j below the wave's first row: every lane reads its own row, A[T(ic) + j]: one pointer, immediate offsets;
//   j among the wave's rows: own row or the mirrored entry A[T(j) + ic], by comparison;
//   j beyond the wave's last row: every lane reads the mirrored entry; T(j + SPLIT) - T(j) = SPLIT j + SPLIT (SPLIT + 1) / 2
//   grows by SPLIT^2 per step, so the offsets are running sums.
// (The first version computed T(j) and the comparison for every element: ~12 vector instructions and a 32-bit multiply per
// element; profiles/r05_lmi_block_phases.txt.)
template <typename T, int SPLIT, bool SUM = true>
__device__ __forceinline__ T matvec_row(const T* A, const T* vv, int r, int i0, int wave, int ic, int part) {
  constexpr int W = 64 / SPLIT, S2 = SPLIT * SPLIT * (int)sizeof(T), SZ = (int)sizeof(T);
  const char* Ab = reinterpret_cast<const char*>(A);
  auto at = [&](int byte_off) { return *reinterpret_cast<const T*>(Ab + byte_off); };
  T q0 = T(0), q1 = T(0), q2 = T(0), q3 = T(0);
  const int Tc = ic * (ic + 1) / 2;
  const int nU = (r - i0) / SPLIT;                   // steps every part has a column for
  const int nA = (W * wave) / SPLIT;                 // steps with j below the wave's first row, whatever the part
  int nB = (W * wave + W - 1) / SPLIT + 1;           // first step with j beyond the wave's last row, whatever the part
  nB = nB < nU ? nB : nU;
  const T* pv = vv + i0 + part;
  const T* pa = A + Tc + i0 + part;
  constexpr int UC = sizeof(T) == 4 ? 8 : 4;        // (elements in flight per lane; the fp64 instances have no registers for 8)
  int t = 0;
  for (; t + UC <= nA; t += UC) {
#pragma unroll
    for (int u = 0; u < UC; u += 4) {
      q0 = fma(pa[SPLIT * (u + 0)], pv[SPLIT * (u + 0)], q0);
      q1 = fma(pa[SPLIT * (u + 1)], pv[SPLIT * (u + 1)], q1);
      q2 = fma(pa[SPLIT * (u + 2)], pv[SPLIT * (u + 2)], q2);
      q3 = fma(pa[SPLIT * (u + 3)], pv[SPLIT * (u + 3)], q3);
    }
    pa += SPLIT * UC;
    pv += SPLIT * UC;
  }
  for (; t < nA; ++t) {
    q0 = fma(pa[0], pv[0], q0);
    pa += SPLIT;
    pv += SPLIT;
  }
  int j = i0 + part + SPLIT * nA;
  int offM = (j * (j + 1) / 2 + ic) * SZ;                             // (bytes) the mirrored entry (j, ic)
  int dj = (SPLIT * j + SPLIT * (SPLIT + 1) / 2) * SZ;
#pragma unroll 4
  for (; t < nB; ++t) {
    const T a = at(j <= ic ? (Tc + j) * SZ : offM);
    q0 = fma(a, *pv, q0);
    offM += dj;
    dj += S2;
    j += SPLIT;
    pv += SPLIT;
  }
  for (; t + UC <= nU; t += UC) {
    int o[UC];
    o[0] = offM;
#pragma unroll
    for (int u = 1; u < UC; ++u) o[u] = o[u - 1] + dj + (u - 1) * S2;
    T a[UC];
#pragma unroll
    for (int u = 0; u < UC; ++u) a[u] = at(o[u]);
#pragma unroll
    for (int u = 0; u < UC; u += 4) {
      q0 = fma(a[u + 0], pv[(u + 0) * SPLIT], q0);
      q1 = fma(a[u + 1], pv[(u + 1) * SPLIT], q1);
      q2 = fma(a[u + 2], pv[(u + 2) * SPLIT], q2);
      q3 = fma(a[u + 3], pv[(u + 3) * SPLIT], q3);
    }
    offM = o[UC - 1] + dj + (UC - 1) * S2;
    dj += UC * S2;
    pv += UC * SPLIT;
    j += UC * SPLIT;
  }
  for (; t < nU; ++t) {
    q1 = fma(at(offM), *pv, q1);
    offM += dj;
    dj += S2;
    pv += SPLIT;
    j += SPLIT;
  }
  if (j < r) q2 = fma(at(j <= ic ? (Tc + j) * SZ : offM), *pv, q2);     // (the parts that have one more column)
  T q = (q0 + q1) + (q2 + q3);
  if constexpr (SUM) {
    if constexpr (SPLIT >= 8) q += __shfl_xor(q, 8);
    if constexpr (SPLIT >= 4) q += __shfl_xor(q, 16);
    if constexpr (SPLIT >= 2) q += __shfl_xor(q, 32);
  }
  return q;
}

// A(i, j) -= v_i w_j + w_i v_j for j = i0 .. i, this lane's columns of row i (row = A + T(i)).
// FLY: `ww` holds p = tau A v instead of w = p - K v (K needs a sum over the workgroup: reduce_column writes p BEFORE that
// sum's barrier and saves the barrier between w and the update) and `vv` the raw column (v_{i0} = vi0 is not in it):
// w_j = p_j - K v_j is formed here, one more multiply-add per element.
template <typename T, int SPLIT, bool FLY = false>
__device__ __forceinline__ void update_row(T* row, const T* vv, const T* ww, int i0, int wave, int i, int part, T vi, T w,
                                           T K = T(0), T vi0 = T(0)) {
  constexpr int W = 64 / SPLIT, NB = W / SPLIT + 1;
  const int nA = (W * wave) / SPLIT;
  T* pr = row + i0 + part;
  const T* pw = ww + i0 + part;
  const T* pu = vv + i0 + part;
  bool first = FLY && part == 0;                // (this lane's first element is column i0)
  auto neww = [&](T a, T b, T c) { return FLY ? a - (vi * fma(-K, c, b) + w * c) : a - (vi * b + w * c); };
  // (every load of a group before its first store: the compiler cannot know that the row does not overlap vv / ww, and one
  // LDS round trip per element is what it would schedule otherwise)
  constexpr int UU = sizeof(T) == 4 ? 8 : 4;
  int t = 0;
  for (; t + UU <= nA; t += UU) {
    T a[UU], b[UU], c[UU];
#pragma unroll
    for (int u = 0; u < UU; ++u) { a[u] = pr[SPLIT * u]; b[u] = pw[SPLIT * u]; c[u] = pu[SPLIT * u]; }
    if (first) { c[0] = vi0; first = false; }
#pragma unroll
    for (int u = 0; u < UU; ++u) pr[SPLIT * u] = neww(a[u], b[u], c[u]);
    pr += SPLIT * UU;
    pw += SPLIT * UU;
    pu += SPLIT * UU;
  }
  {                                             // the rest of the stretch below the wave's rows (fewer than UU steps) ...
    T a[UU - 1], b[UU - 1], c[UU - 1];
#pragma unroll
    for (int u = 0; u < UU - 1; ++u)
      if (t + u < nA) { a[u] = pr[SPLIT * u]; b[u] = pw[SPLIT * u]; c[u] = pu[SPLIT * u]; }
    if (first && t < nA) { c[0] = vi0; first = false; }
#pragma unroll
    for (int u = 0; u < UU - 1; ++u)
      if (t + u < nA) pr[SPLIT * u] = neww(a[u], b[u], c[u]);
    const int left = nA - t;
    pr += SPLIT * left;
    pw += SPLIT * left;
    pu += SPLIT * left;
  }
  const int j0 = i0 + part + SPLIT * nA;        // ... and the columns among the wave's rows, up to the diagonal
#pragma unroll
  for (int g = 0; g < NB; g += UU) {
    T a[UU], b[UU], c[UU];
#pragma unroll
    for (int u = 0; u < UU; ++u)
      if (g + u < NB && j0 + SPLIT * (g + u) <= i) { a[u] = pr[SPLIT * (g + u)]; b[u] = pw[SPLIT * (g + u)]; c[u] = pu[SPLIT * (g + u)]; }
    if (g == 0 && first) { c[0] = vi0; first = false; }
#pragma unroll
    for (int u = 0; u < UU; ++u)
      if (g + u < NB && j0 + SPLIT * (g + u) <= i) pr[SPLIT * (g + u)] = neww(a[u], b[u], c[u]);
  }
}

// One column of the Householder reduction with SPLIT lanes per row (three workgroup barriers).
template <typename T, int SPLIT, int NTH, bool KEEP>
__device__ __forceinline__ void reduce_column(T* A, int r, int kc, T* dd, T* ee, T* tt, T* vv, T* ww, T* red, const int tid) {
  constexpr int NW = NTH / 64, W = 64 / SPLIT;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (loop bounds in scalar registers)
  const int rw = lane % W, part = lane / W;
  LB_TICK_DECL;
  const int i0 = kc + 1, rlo = i0 + W * wave, i = rlo + rw;
  const bool has = i < r;
  const int Ti = i * (i + 1) / 2;
  T* rd = red + (kc & 1) * 2 * kWaves;
  const T x = has ? A[Ti + kc] : T(0);
  if (has && part == 0) vv[i] = x;
  const T sigma = bsum<T, NW>(part == 0 ? x * x : T(0), rd, tid);   // barrier 1 (vv = the raw column is visible too)
  LB_TICK(0);
  const T x0 = vv[i0];
  const T below = sigma - x0 * x0;                               // what the reflector has to annihilate
  if (!(below > lw::Eps<T>::tiny * lw::Eps<T>::tiny)) {          // nothing to do: H = I  (the same for every thread)
    if (tid == 0) { dd[kc] = A[kc * (kc + 1) / 2 + kc]; ee[kc] = x0; if constexpr (KEEP) tt[kc] = T(0); }
    __syncthreads();
    return;
  }
  const T alpha = (x0 >= T(0) ? T(-1) : T(1)) * sqrt(sigma);
  const T taup = T(1) / (sigma - x0 * alpha);
  // p = tau A v with v = x - alpha e_i0 (the raw column x is in vv; the correction is one extra term)
  T p = T(0), vi = T(0);
  if (rlo < r) {                            // (a wave without a live row has nothing to sum)
    const T q = matvec_row<T, SPLIT>(A, vv, r, i0, wave, has ? i : r - 1, part);   // (idle lanes read a valid row)
    if (has) {
      p = taup * (q - alpha * A[Ti + i0]);
      vi = i == i0 ? x - alpha : x;
    }
  }
  if (has && part == 0) ww[i] = p;                                // (p, not w: update_row<FLY> forms w_j = p_j - K v_j)
  const T pv = bsum<T, NW>(part == 0 ? p * vi : T(0), rd + kWaves, tid);                // barrier 2 (ww is visible too)
  LB_TICK(1);
  const T K = T(0.5) * taup * pv;
  const T w = fma(-K, vi, p);
  LB_TICK(2);
  // A -= v w' + w v' on the lower triangle of the live block (vv keeps the raw column: v_{i0} = x0 - alpha goes by value)
  if (has) update_row<T, SPLIT, true>(A + Ti, vv, ww, i0, wave, i, part, vi, w, K, x0 - alpha);
  if (tid == 0) { dd[kc] = A[kc * (kc + 1) / 2 + kc]; ee[kc] = alpha; if constexpr (KEEP) tt[kc] = taup; }
  __syncthreads();                                               // barrier 3
  LB_TICK(3);
}

// Householder reduction of the packed matrix to tridiagonal form: dd (diagonal), ee (signed sub-diagonal).  The raw column
// x of step kc stays in A(i > kc, kc); KEEP also leaves tau_kc = 1 / (sigma - x0 alpha) in tt[kc] (0: H = I), so that
// H_kc = I - tau v v' with v = x - alpha e_{kc+1}, alpha = ee[kc], can be applied again (the backward maps the eigenvector back).
// Wave w works on rows i0 + W w .. of the live block, SPLIT = 64 / W lanes per row: as many as the workgroup has for the rows
// that are left (two at 257 .. 512 rows of 1024 threads, eight at 128 and fewer).
template <typename T, int NTH, bool KEEP>
__device__ __forceinline__ void tridiagonalise(T* A, int r, T* dd, T* ee, T* tt, T* vv, T* ww, T* red, const int tid) {
  for (int kc = 0; kc + 2 < r; ++kc) {
    const int m = r - kc - 1;                       // rows of the live block
#if defined(RAYEN_LB_MAX_SPLIT) && RAYEN_LB_MAX_SPLIT == 2
    reduce_column<T, 2, NTH, KEEP>(A, r, kc, dd, ee, tt, vv, ww, red, tid);
#elif defined(RAYEN_LB_MAX_SPLIT) && RAYEN_LB_MAX_SPLIT == 4
    if (4 * m <= NTH) reduce_column<T, 4, NTH, KEEP>(A, r, kc, dd, ee, tt, vv, ww, red, tid);
    else reduce_column<T, 2, NTH, KEEP>(A, r, kc, dd, ee, tt, vv, ww, red, tid);
#else
    // (eight lanes per row pay in fp64: 2.70 against 2.07 ms at r = 100 -- profiles/bench/r05_lmi_block_ab.txt)
    if (sizeof(T) == 4 && 8 * m <= NTH) reduce_column<T, sizeof(T) == 4 ? 8 : 4, NTH, KEEP>(A, r, kc, dd, ee, tt, vv, ww, red, tid);
    else if (4 * m <= NTH) reduce_column<T, 4, NTH, KEEP>(A, r, kc, dd, ee, tt, vv, ww, red, tid);
    else if (2 * m <= NTH || NTH >= 512) reduce_column<T, 2, NTH, KEEP>(A, r, kc, dd, ee, tt, vv, ww, red, tid);
    else if constexpr (NTH < 512) reduce_column<T, 1, NTH, KEEP>(A, r, kc, dd, ee, tt, vv, ww, red, tid);   // (a lane per row)
#endif
  }
  if (tid == 0) {
    if (r >= 2) {
      dd[r - 2] = A[(r - 2) * (r - 1) / 2 + (r - 2)];
      ee[r - 2] = A[(r - 1) * r / 2 + (r - 2)];
    }
    dd[r - 1] = A[(r - 1) * r / 2 + (r - 1)];
    ee[r - 1] = T(0);
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// Matrices whose packed triangle does not fit the LDS (r = 282 .. 304 in fp32, 198 .. 212 in fp64): the first HC COLUMNS stay in
// registers.  Entry (i, c < HC) lives in slot c / 2 of the lane with part c % 2 of row i; the LDS holds the packed triangle of
// rows and columns HC .. r - 1.  The first HC columns are reduced with fixed owners -- the rows HC + 32 w + (lane % 32) in wave
// w, the rows 0 .. HC - 1 in the LAST wave (idle otherwise: HC <= 32, at most nine waves of rows in the LDS) --, after which
// what is left IS the LDS matrix and tridiagonalise() runs on it with every pointer shifted by HC.
//   q_i = sum_j A(i, j) x_j splits into: this lane's slots (columns i0 .. min(i, HC - 1));  for a row >= HC the LDS columns
//   (matvec_row from column 0 of the LDS matrix);  for a row < HC the entries (j > i, i) that sit in OTHER rows' slots -- every
//   lane multiplies its slots by its own x, the 32 lanes of a part add up per slot (shuffles), one lane per wave leaves the
//   sum in ww[column][wave] before the barrier of sigma, the row's lane adds the waves' sums in a fixed order.
// ---------------------------------------------------------------------------------------------
template <typename T>
struct HeadCols { static constexpr int value = sizeof(T) == 4 ? 24 : 16; };

template <typename T, int HS>
__device__ __forceinline__ T slot_of(const T (&head)[HS], int s) {
  T x = T(0);
#pragma unroll
  for (int u = 0; u < HS; ++u) x = u == s ? head[u] : x;
  return x;
}

// S(v): columns < HC into the slots, the rest into the LDS triangle (no barrier)
template <typename T, int NTH, int HC>
__device__ __forceinline__ void form_S_head(T* A, T (&head)[HC / 2], const T* __restrict__ gt, const T* vs, int n, int r, int Pp,
                                            const int tid) {
  constexpr int HS = HC / 2, NW = NTH / 64;
  const int lane = tid & 63, wave = tid >> 6, rw = lane & 31, part = lane >> 5;
  const int i = wave == NW - 1 ? rw : HC + 32 * wave + rw;
  const bool has = wave == NW - 1 ? rw < HC : i < r;
#pragma unroll
  for (int s = 0; s < HS; ++s) head[s] = T(0);
  if (has) {
    const T* row = gt + (size_t)i * (i + 1) / 2 + part;
    for (int a = 0; a < n; ++a) {
      const T va = vs[a];
      const T* g = row + (size_t)a * Pp;
#pragma unroll
      for (int s = 0; s < HS; ++s)
        if (2 * s + part <= i) head[s] = fma(va, g[2 * s], head[s]);
    }
  }
  const int rp = r - HC, Pl = rp * (rp + 1) / 2;
  for (int idx = tid; idx < Pl; idx += NTH) {
    int ip = (int)((sqrtf(8.f * (float)idx + 1.f) - 1.f) * 0.5f);
    while ((ip + 1) * (ip + 2) / 2 <= idx) ++ip;
    while (ip * (ip + 1) / 2 > idx) --ip;
    const int jp = idx - ip * (ip + 1) / 2;
    const T* col = gt + (size_t)(ip + HC) * (ip + HC + 1) / 2 + jp + HC;
    T p0 = T(0), p1 = T(0), p2 = T(0), p3 = T(0);
    int a = 0;
    for (; a + 15 < n; a += 16) {
      T g[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) g[u] = col[(size_t)(a + u) * Pp];
#pragma unroll
      for (int u = 0; u < 16; u += 4) {
        p0 = fma(vs[a + u + 0], g[u + 0], p0);
        p1 = fma(vs[a + u + 1], g[u + 1], p1);
        p2 = fma(vs[a + u + 2], g[u + 2], p2);
        p3 = fma(vs[a + u + 3], g[u + 3], p3);
      }
    }
    for (; a < n; ++a) p0 = fma(vs[a], col[(size_t)a * Pp], p0);
    A[idx] = (p0 + p1) + (p2 + p3);
  }
}

// the first HC columns of the reduction (A: the LDS triangle of rows / columns >= HC); four barriers per column
template <typename T, int NTH, int HC, bool KEEP>
__device__ __forceinline__ void head_phase(T* A, int r, T (&head)[HC / 2], T* dd, T* ee, T* tt, T* vv, T* ww, T* red, const int tid) {
  constexpr int HS = HC / 2, NW = NTH / 64;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), rw = lane & 31, part = lane >> 5;
  const bool head_wave = wave == NW - 1;
  const int i = head_wave ? rw : HC + 32 * wave + rw;
  const bool has = head_wave ? rw < HC : i < r;
  const int rp = r - HC;
  const int nbw = (rp + 31) / 32;                   // waves that hold rows of the LDS matrix
  const int ncontrib = nbw + 1, contributor = head_wave ? nbw : wave;
  const int ip = i - HC, Tp = ip * (ip + 1) / 2;    // (rows of the LDS matrix)
  for (int kc = 0; kc < HC; ++kc) {
    const int i0 = kc + 1, pk = kc & 1, sk = kc >> 1;
    const bool live = has && i >= i0;
    T* rd = red + (kc & 1) * 2 * kWaves;
    const T xown = live && part == pk ? slot_of<T, HS>(head, sk) : T(0);
    const T xi = xown + __shfl_xor(xown, 32);       // (both lanes of the row)
    if (live && part == pk) vv[i] = xi;
    if (has && i == kc && part == pk) dd[kc] = slot_of<T, HS>(head, sk);          // A(kc, kc) is final
    // what the rows above HC need from the other rows' slots: sum_{j > c} A(j, c) x_j for the columns c = i0 .. HC - 1
    if (head_wave || wave < nbw) {
#pragma unroll
      for (int s = 0; s < HS; ++s) {
        if (2 * s + 1 < i0) continue;               // (both parts' columns are behind the reduction)
        const int c = 2 * s + part;
        T val = live && i > c && c >= i0 ? head[s] * xi : T(0);
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) val += __shfl_xor(val, o);
        if (rw == 0 && c >= i0) ww[c * ncontrib + contributor] = val;
      }
    }
    const T sigma = bsum<T, NW>(part == pk ? xown * xown : T(0), rd, tid);        // barrier 1
    const T x0 = vv[i0];
    const T below = sigma - x0 * x0;
    if (!(below > lw::Eps<T>::tiny * lw::Eps<T>::tiny)) {          // H = I
      if (tid == 0) { ee[kc] = x0; if constexpr (KEEP) tt[kc] = T(0); }
      __syncthreads();
      continue;
    }
    const T alpha = (x0 >= T(0) ? T(-1) : T(1)) * sqrt(sigma);
    const T taup = T(1) / (sigma - x0 * alpha);
    T q = T(0);
    if (live) {
#pragma unroll
      for (int s = 0; s < HS; ++s) {
        const int c = 2 * s + part;
        if (c >= i0 && c <= i) q = fma(head[s], vv[c], q);
      }
      if (i0 < HC) {
        if (part == (i0 & 1)) q = fma(-alpha, slot_of<T, HS>(head, i0 >> 1), q);      // - alpha A(i, i0)
      } else if (part == 0) {
        q = fma(-alpha, A[Tp], q);                                                   // i0 = HC: column 0 of the LDS matrix
      }
      if (head_wave && part == 0) {
        T cs = T(0);
        for (int w = 0; w < ncontrib; ++w) cs += ww[i * ncontrib + w];
        q += cs;
      }
    }
    if (!head_wave && wave < nbw) q += matvec_row<T, 2, false>(A, vv + HC, rp, 0, wave, has ? ip : rp - 1, part);
    q += __shfl_xor(q, 32);
    const T p = live ? taup * q : T(0);
    const T vi = live ? (i == i0 ? xi - alpha : xi) : T(0);
    const T pv = bsum<T, NW>(part == 0 ? p * vi : T(0), rd + kWaves, tid);        // barrier 2
    const T K = T(0.5) * taup * pv;
    const T w = fma(-K, vi, p);
    if (live && part == 0) {
      ww[i] = w;
      if (i == i0) vv[i0] = vi;
    }
    __syncthreads();                                               // barrier 3
    if (live) {
#pragma unroll
      for (int s = 0; s < HS; ++s) {
        const int c = 2 * s + part;
        if (c >= i0 && c <= i) head[s] = head[s] - (vi * ww[c] + w * vv[c]);
      }
      if (!head_wave) update_row<T, 2>(A + Tp, vv + HC, ww + HC, 0, wave, ip, part, vi, w);
    }
    if (tid == 0) { ee[kc] = alpha; if constexpr (KEEP) tt[kc] = taup; }
    __syncthreads();                                               // barrier 4
  }
}


// S(v) out of a row of products (packed lower triangle, the order of the LMI's rows of W): into the LDS triangle and, with
// register columns, into the slots (the same split as form_S_head)
template <typename T, int NTH, int HC>
__device__ __forceinline__ void copy_S(T* A, T (&head)[HC > 0 ? HC / 2 : 1], const T* __restrict__ srow, int r, int P, const int tid) {
  if constexpr (HC == 0) {
    for (int idx = tid; idx < P; idx += NTH) A[idx] = srow[idx];
  } else {
    constexpr int HS = HC / 2, NW = NTH / 64;
    const int lane = tid & 63, wave = tid >> 6, rw = lane & 31, part = lane >> 5;
    const int i = wave == NW - 1 ? rw : HC + 32 * wave + rw;
    const bool has = wave == NW - 1 ? rw < HC : i < r;
#pragma unroll
    for (int s = 0; s < HS; ++s) head[s] = has && 2 * s + part <= i ? srow[(size_t)i * (i + 1) / 2 + 2 * s + part] : T(0);
    const int rp = r - HC, Pl = rp * (rp + 1) / 2;
    for (int idx = tid; idx < Pl; idx += NTH) {
      int ip = (int)((sqrtf(8.f * (float)idx + 1.f) - 1.f) * 0.5f);
      while ((ip + 1) * (ip + 2) / 2 <= idx) ++ip;
      while (ip * (ip + 1) / 2 > idx) --ip;
      const int jp = idx - ip * (ip + 1) / 2;
      A[idx] = srow[(size_t)(ip + HC) * (ip + HC + 1) / 2 + jp + HC];
    }
  }
}

// (four waves per SIMD = two 512-thread workgroups per compute unit: one register more and it is one)
template <typename T, int NTH, int HC>
__global__ __launch_bounds__(NTH) __attribute__((amdgpu_waves_per_eu(NTH <= 256 ? kSmallWavesPerEu : 4))) void lmi_block_kernel(
    const T* __restrict__ gt, const T* __restrict__ dt, const T* __restrict__ nat, const T* __restrict__ y0,
    const int32_t* __restrict__ lin_id, int r, int n, int k, int m, int P, int Pp, int Mp, int Kp, int identity,
    int lmi_seg, const T* __restrict__ v, int64_t B, int64_t ldv, T* y, int64_t ldy,
    T* kappa_out, int32_t* active_out, int32_t* __restrict__ nan_flag, const T* kappa_in, int64_t ldk_in,
    const T* __restrict__ prods, int64_t ldt, int lmi_row0, int n_rows, int old_mode) {
  // old_mode: the RAYEN_old head (rayen/constraint_module.py:460-466) -- the same kappa, the step 1 / (||v|| e^beta + kappa)
  // with beta in column n of the input (not with prods).
  // prods != nullptr (sets with many generators, rayen_abi.hip::project_from_products): row b of T = v W_ext' from a library
  // GEMM holds what this kernel otherwise forms per sample -- D v at the linear rows' W rows, S(v) packed at the LMI's rows,
  // NA_E v behind the rows of W -- and v is read only for the output of sets without equalities (no copy of it in LDS)
  extern __shared__ __attribute__((aligned(16))) unsigned char lb_smem[];
  T* A = reinterpret_cast<T*>(lb_smem);
  T* dd = A + (HC > 0 ? (r - HC) * (r - HC + 1) / 2 : P);
  T* ee = dd + r;
  T* vv = ee + r;
  T* ww = vv + r;
  T* red = ww + r;            // [3][2 * kWaves]
  T* vs = red + 6 * kWaves;
  constexpr int kThreads = NTH, NW = NTH / 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  bool bad = false;
  T head[HC > 0 ? HC / 2 : 1];

  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    __syncthreads();          // (the previous sample's last readers of vs / dd / ee)
    const T* prow = prods != nullptr ? prods + b * ldt : nullptr;
    if (prow == nullptr)
      for (int a = tid; a < n; a += kThreads) vs[a] = v[b * ldv + a];
    // sets with quadratics / cones next to the LMI: the lane kernel has left the maximum over everything else (and its row, in
    // active_out) where this sample's outputs go -- kappa_in may be kappa_out or COLUMN 0 OF y.  Every thread takes its copy
    // HERE, barriers ahead of the first store to this sample's y / kappa_out / active_out (a read next to those stores, with
    // no barrier in between, let a late wave see thread 0's y[b][0] as the other constraints' kappa).
    T other = T(0);
    int other_seg = -1, other_row = 0;
    if (kappa_in != nullptr) {
      other = kappa_in[b * ldk_in];
      if (active_out) { other_seg = active_out[2 * b]; other_row = active_out[2 * b + 1]; }
    }
    __syncthreads();

    // ---- linear rows: (value, index among the linear rows) of the largest D_i . v; the lowest index wins a tie
    T kap = T(0);
    int who = -1;
    for (int i = tid; i < m; i += kThreads) {
      T acc = T(0);
      if (prow != nullptr) {
        acc = prow[lin_id[2 * i + 1]];
      } else {
        const T* col = dt + i;
        for (int a = 0; a < n; ++a) acc = fma(vs[a], col[(size_t)a * Mp], acc);
      }
      if (acc > kap) { kap = acc; who = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const T ob = __shfl_xor(kap, o);
      const int ow = __shfl_xor(who, o);
      if (ob > kap || (ob == kap && ow >= 0 && (who < 0 || ow < who))) { kap = ob; who = ow; }
    }
    int* redi = reinterpret_cast<int*>(red + 5 * kWaves);     // (slot set 2, second half: the indices)
    if (lane == 0) { red[2 * 2 * kWaves + wave] = kap; redi[wave] = who; }

#ifdef RAYEN_LB_PROFILE
    unsigned long long lb_t_ = clock64();
    const unsigned long long lb_start = lb_t_;
#endif
    if (prow != nullptr) copy_S<T, NTH, HC>(A, head, prow + lmi_row0, r, P, tid);
    else if constexpr (HC > 0) form_S_head<T, NTH, HC>(A, head, gt, vs, n, r, Pp, tid);
    else form_S<T, NTH>(A, gt, vs, n, P, Pp, tid);
    __syncthreads();
    LB_TICK(4);
    {
      kap = red[2 * 2 * kWaves];
      who = redi[0];
#pragma unroll
      for (int w = 1; w < NW; ++w) {
        const T ob = red[2 * 2 * kWaves + w];
        const int ow = redi[w];
        if (ob > kap || (ob == kap && ow >= 0 && (who < 0 || ow < who))) { kap = ob; who = ow; }
      }
    }
    int aseg = who >= 0 ? lin_id[2 * who] : -1, arow = who >= 0 ? lin_id[2 * who + 1] : 0;

    if constexpr (HC > 0) head_phase<T, NTH, HC, false>(A, r, head, dd, ee, nullptr, vv, ww, red, tid);
    tridiagonalise<T, NTH, false>(A, r - HC, dd + HC, ee + HC, nullptr, vv + HC, ww + HC, red, tid);
#ifdef RAYEN_LB_PROFILE
    lb_t_ = clock64();
#endif
    for (int i = 1 + tid; i < r; i += kThreads) ww[i] = ee[i - 1] * ee[i - 1];     // (visible after the bracket's barrier)

    // ---- lambda_max of the tridiagonal: Gershgorin bracket, then Sturm counts at 512 shifts per round
    T lo, hi, scale;
    {
      T l = dd[0], h = dd[0], s = T(0);
      for (int i = tid; i < r; i += kThreads) {
        const T off = (i > 0 ? fabs(ee[i - 1]) : T(0)) + (i + 1 < r ? fabs(ee[i]) : T(0));
        l = fmin(l, dd[i] - off);
        h = fmax(h, dd[i] + off);
        s = fmax(s, fabs(dd[i]) + off);
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        l = fmin(l, __shfl_xor(l, o));
        h = fmax(h, __shfl_xor(h, o));
        s = fmax(s, __shfl_xor(s, o));
      }
      if (lane == 0) { red[wave] = l; red[kWaves + wave] = h; red[2 * kWaves + wave] = s; }
      __syncthreads();
      lo = red[0]; hi = red[kWaves]; scale = red[2 * kWaves];
#pragma unroll
      for (int w = 1; w < NW; ++w) {
        lo = fmin(lo, red[w]);
        hi = fmax(hi, red[kWaves + w]);
        scale = fmax(scale, red[2 * kWaves + w]);
      }
    }
    const T pad = scale * (sizeof(T) == 4 ? T(1e-6) : T(1e-14)) + lw::Eps<T>::tiny;
    lo -= pad;
    hi += pad;
    const T floor_q = fmax(scale * (sizeof(T) == 4 ? T(1e-30) : T(1e-200)), lw::Eps<T>::tiny);
    int* firsts = reinterpret_cast<int*>(red + 3 * kWaves);      // [2][kWaves]: the rounds alternate
    // (NTH + 1)^rounds >= 2^27 in fp32, 2^60 in fp64
    constexpr int kRounds = sturm_rounds(NTH, sizeof(T) == 4 ? 27 : 60);
    for (int round = 0; round < kRounds; ++round) {
      const T step = (hi - lo) * (T(1) / T(kThreads + 1));
      const T sig = lo + step * (T)(tid + 1);
      int cnt = 0;                                               // eigenvalues below sig = negative pivots of T - sig I
      T q = dd[0] - sig;
      cnt += q < T(0);
      // (eight rows' d and e^2 fetched ahead of the eight dependent steps that use them; ww holds e_{i-1}^2)
      int i = 1;
      for (; i + 8 <= r; i += 8) {
        T d8[8], e8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { d8[u] = dd[i + u]; e8[u] = ww[i + u]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (fabs(q) < floor_q) q = q < T(0) ? -floor_q : floor_q;
          q = d8[u] - sig - quot(e8[u], q);
          cnt += q < T(0);
        }
      }
      for (; i < r; ++i) {
        if (fabs(q) < floor_q) q = q < T(0) ? -floor_q : floor_q;
        q = dd[i] - sig - quot(ww[i], q);
        cnt += q < T(0);
      }
      const unsigned long long above = __ballot(cnt >= r);       // sig beyond the largest eigenvalue
      int* slot = firsts + (round & 1) * kWaves;
      if (lane == 0) slot[wave] = above ? 64 * wave + __builtin_ctzll(above) : kThreads;
      __syncthreads();
      int first = slot[0];
#pragma unroll
      for (int w = 1; w < NW; ++w) first = slot[w] < first ? slot[w] : first;
      const T new_lo = first == 0 ? lo : lo + step * (T)first;
      const T new_hi = first == kThreads ? hi : lo + step * (T)(first + 1);
      lo = new_lo;
      hi = new_hi;
    }
    const T lam = T(0.5) * (lo + hi);
    LB_TICK(5);
#ifdef RAYEN_LB_PROFILE
    if (threadIdx.x == 0 && blockIdx.x == 0) { g_lb_prof[6] += clock64() - lb_start; g_lb_prof[7] += 1; }
#endif
    if (lam > kap) { kap = lam; aseg = lmi_seg; arow = 0; }
    if (kappa_in != nullptr && other >= kap) {        // (the copies taken at the top of this sample)
      kap = other;
      if (active_out) { aseg = other_seg; arow = other_row; }
    }

    T scl = T(1) / fmax(T(1), kap);
    if (old_mode) {
      T part = T(0);
      for (int a = tid; a < n; a += kThreads) part = fma(vs[a], vs[a], part);
      const T nrm = sqrt(bsum<T, NW>(part, red, tid));           // (slot set 0: its last readers are barriers behind)
      scl = nrm > T(0) ? T(1) / (nrm * exp(v[b * ldv + n]) + kap) : T(0);
    }
    if (tid == 0) {
      if (kappa_out) kappa_out[b] = kap;
      if (active_out) { active_out[2 * b] = aseg; active_out[2 * b + 1] = arow; }
    }
    T* yrow = y + b * ldy;
    for (int i = tid; i < k; i += kThreads) {
      T val;
      if (prow != nullptr) {
        val = fma(identity ? v[b * ldv + i] : prow[n_rows + i], scl, y0[i]);
      } else if (identity) {
        val = fma(vs[i], scl, y0[i]);
      } else {
        T acc = T(0);
        const T* col = nat + i;
        for (int a = 0; a < n; ++a) acc = fma(vs[a], col[(size_t)a * Kp], acc);
        val = fma(acc, scl, y0[i]);
      }
      bad |= (val != val);
      yrow[i] = val;
    }
  }
  if (nan_flag && bad) atomicOr(nan_flag, 1);
}

// LDS of the backward's workgroup: A[P] | dd[r] (later the eigenvector) | ee[r] | vv[r] | ww[r] | tt[r] | red | vs[n] | ts[n]
__host__ __device__ inline size_t lds_bwd_elems(int r, int n) {
  return (size_t)r * (r + 1) / 2 + 5 * (size_t)r + 6 * kWaves + 2 * (size_t)n + 8;
}

// grad_v for one sample per workgroup.  y = y0 + NA_E v / max(1, kappa):
//   grad_v = NA_E' g / max(1, kappa) - [kappa > 1] (g' NA_E v) / kappa^2 * d kappa / d v,
// and d kappa / d v_a = x' G_a x for the unit eigenvector x of lambda_max(S(v)) when the LMI is the active row (a row of D
// otherwise).  The reduction is repeated keeping the reflectors; x = H_0 ... H_{r-3} z, z the eigenvector of the tridiagonal
// matrix (inverse iteration at the forward's kappa, the same factorisation with the same guards as rayen_lmi_wave.h:323-354);
// then x x' (off-diagonal entries twice) replaces the matrix in packed order and every generator is ONE dot product with it,
// a wave each -- the same n P words of G the forward reads.
// The unit eigenvector of lambda_max in zz = dd (visible to the whole workgroup on return), after tridiagonalise<KEEP> (and
// head_phase): inverse iteration on the tridiagonal matrix at the forward's kappa (the factorisation and guards of
// rayen_lmi_wave.h:323-354), then x = H_0 ... H_{r-3} z -- the reflectors whose columns are in the LDS by ONE wave with the
// vector in its registers (row lane + 64 q: no barrier per reflector), those of the slot columns by the whole workgroup.
template <typename T, int NTH, int HC>
__device__ __forceinline__ void top_eigenvector(T* A, int r, T kap, T (&head)[HC > 0 ? HC / 2 : 1], T* dd, T* ee, T* tt, T* vv,
                                                T* ww, T* red, const int tid) {
  constexpr int NW = NTH / 64, kMaxChunks = 5;
  const int lane = tid & 63, wave = tid >> 6;
  T* zz = dd;
  // ---- z: inverse iteration on M = (kappa + shift) I - T = L D L'.  ww: D, vv: the sub-diagonal of L.
  if (wave == 0) {
    T scale = fabs(kap);
    for (int i = lane; i < r; i += 64) scale = fmax(scale, fabs(dd[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) scale = fmax(scale, __shfl_xor(scale, o));
    const T shift = lw::Eps<T>::shift * fmax(scale, lw::Eps<T>::tiny);
    if (lane == 0) {
      T dprev = fmax(kap + shift - dd[0], shift * T(1e-3));
      ww[0] = dprev;
      vv[0] = T(0);
      for (int i = 1; i < r; ++i) {
        const T li = ee[i - 1] / dprev;              // M's off-diagonal is -ee: l = -ee / D, kept with the sign folded
        const T di = fmax(kap + shift - dd[i] - li * ee[i - 1], shift * T(1e-3));
        vv[i] = -li;
        ww[i] = T(1) / di;                           // (the solves multiply)
        dprev = di;
      }
      ww[0] = T(1) / ww[0];
      for (int i = 0; i < r; ++i) zz[i] = T(1) + T(0.01) * (T)i;   // not orthogonal to anything special
      for (int it = 0; it < 3; ++it) {
        T prev = zz[0];
        for (int i = 1; i < r; ++i) { prev = fma(-vv[i], prev, zz[i]); zz[i] = prev; }     // L y = b
        prev = prev * ww[r - 1];
        zz[r - 1] = prev;
        T nrm2 = prev * prev;
        for (int i = r - 2; i >= 0; --i) {                                                   // D L' z = y
          prev = fma(-vv[i + 1], prev, zz[i] * ww[i]);
          zz[i] = prev;
          nrm2 = fma(prev, prev, nrm2);
        }
        const T inv = T(1) / sqrt(fmax(nrm2, lw::Eps<T>::tiny));
        for (int i = 0; i < r; ++i) zz[i] *= inv;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // ---- x = H_0 H_1 ... H_{r-3} z, the vector in this wave's registers (row lane + 64 q)
    T xq[kMaxChunks];
#pragma unroll
    for (int q = 0; q < kMaxChunks; ++q) xq[q] = lane + 64 * q < r ? zz[lane + 64 * q] : T(0);
    for (int c = r - 3; c >= HC; --c) {                // (the reflectors whose columns are in the LDS)
      const T tc = tt[c];
      if (tc == T(0)) continue;                      // (wave-uniform)
      const T alpha = ee[c];
      T hv[kMaxChunks];
      T dot = T(0);
#pragma unroll
      for (int q = 0; q < kMaxChunks; ++q) {
        const int i = lane + 64 * q;
        T h = T(0);
        if (i > c && i < r) {
          h = A[(i - HC) * (i - HC + 1) / 2 + (c - HC)];
          if (i == c + 1) h -= alpha;
        }
        hv[q] = h;
        dot = fma(h, xq[q], dot);
      }
      dot = lw::wsum(dot) * tc;
#pragma unroll
      for (int q = 0; q < kMaxChunks; ++q) xq[q] = fma(-dot, hv[q], xq[q]);
    }
#pragma unroll
    for (int q = 0; q < kMaxChunks; ++q)
      if (lane + 64 * q < r) zz[lane + 64 * q] = xq[q];
  }
  __syncthreads();
  if constexpr (HC > 0) {
    // ---- the reflectors of the columns in the slots, HC - 1 .. 0: their entries sit with the rows' lanes (head_phase)
    const int rw = lane & 31, part = lane >> 5;
    const bool head_wave = wave == NW - 1;
    const int i = head_wave ? rw : HC + 32 * wave + rw;
    const bool has = head_wave ? rw < HC : i < r;
    for (int c = HC - 1; c >= 0; --c) {
      const T tc = tt[c];
      if (tc == T(0)) continue;                      // (the same for every thread)
      T h = T(0);
      if (has && i > c && part == (c & 1)) {
        h = slot_of<T, HC / 2>(head, c >> 1);
        if (i == c + 1) h -= ee[c];
      }
      const T dot = bsum<T, NW>(h != T(0) ? h * zz[i] : T(0), red + (c & 1) * 2 * kWaves, tid) * tc;
      if (h != T(0)) zz[i] = fma(-dot, h, zz[i]);
      __syncthreads();
    }
  }
}

template <typename T, int NTH, int HC>
__global__ __launch_bounds__(NTH) __attribute__((amdgpu_waves_per_eu(NTH <= 256 ? kSmallWavesPerEu : 4))) void lmi_block_bwd_kernel(
    const T* __restrict__ gt, const T* __restrict__ dt, const T* __restrict__ nrm, const int32_t* __restrict__ rho_of, int r,
    int n, int k, int P, int Pp, int Mp, int identity, int lmi_seg, const T* __restrict__ v, int64_t B, int64_t ldv,
    const T* __restrict__ kappa, const int32_t* __restrict__ active, const T* __restrict__ gy, int64_t ldg,
    T* __restrict__ gv, int64_t ldgv, int only_lmi, const T* __restrict__ prods, int64_t ldt, T* __restrict__ coeff,
    int64_t ldc, T* __restrict__ gs_out, int lmi_row0, int n_rows, int old_mode) {
  // old_mode (RAYEN_old, not with prods): s = 1 / (||v|| e^beta + kappa), so kappa always matters and ||v||, beta get gradients:
  // grad_v = s t - s^2 (t.v) (e^beta v / ||v|| + grad kappa), grad_beta = -s^2 (t.v) ||v|| e^beta in column n (rayen_generic.hip)
  // prods != nullptr (rayen_abi.hip: rayen_ray_project_bwd_coefficients_*): S(v) and NA_E v come out of row b of T = v W_ext',
  // and instead of grad_v this kernel leaves the row of coefficients C with grad_v = s g [sets without equalities: gs_out]
  // + C W_ext -- s g at the rows of NA_E, -s^2 (g'N v) at the active linear row or times (2 - [i = j]) x_i x_j at the LMI's
  // rows -- for the library GEMM that follows (the k generators are contracted there, not one wave each here)
  extern __shared__ __attribute__((aligned(16))) unsigned char lb_smem[];
  T* A = reinterpret_cast<T*>(lb_smem);
  T* dd = A + (HC > 0 ? (r - HC) * (r - HC + 1) / 2 : P);
  T* ee = dd + r;
  T* vv = ee + r;
  T* ww = vv + r;
  T* tt = ww + r;
  T* red = tt + r;            // [3][2 * kWaves]
  T* vs = red + 6 * kWaves;
  T* ts = vs + n;
  T* zz = dd;                 // (the diagonal is dead once (kappa + shift) I - T is factorised)
  constexpr int NW = NTH / 64;
  constexpr int kMaxChunks = 5;             // rows of the eigenvector a lane of the mapping-back wave holds: r <= 320
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  T head[HC > 0 ? HC / 2 : 1];

  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    // (sets with quadratics / cones: the lane kernel has written every sample's gradient but for the LMI's term)
    if (only_lmi && !((old_mode || kappa[b] > T(1)) && active[2 * b] == lmi_seg)) continue;   // (RAYEN_old: kappa always matters)
    __syncthreads();          // (the previous sample's last readers)
    const T* grow = gy + b * ldg;
    if (prods != nullptr) {
      const T* prow = prods + b * ldt;
      T* crow = coeff + b * ldc;
      T part_tv = T(0);
      if (identity) for (int a = tid; a < n; a += NTH) part_tv = fma(grow[a], v[b * ldv + a], part_tv);
      else for (int i = tid; i < k; i += NTH) part_tv = fma(grow[i], prow[n_rows + i], part_tv);
      const T tvp = bsum<T, NW>(part_tv, red + 2 * 2 * kWaves, tid);
      const T kapp = kappa[b];
      const int asegp = active[2 * b], arowp = active[2 * b + 1];
      const bool clippedp = kapp > T(1) && asegp >= 0;
      const T scp = T(1) / fmax(T(1), kapp);
      const T coefp = clippedp ? scp * scp * tvp : T(0);
      const bool lmi_active = clippedp && asegp == lmi_seg;
      // everything but the LMI's rows (those below, once)
      for (int j = tid; j < n_rows; j += NTH)
        if (j < lmi_row0 || j >= lmi_row0 + P) crow[j] = clippedp && !lmi_active && j == arowp ? -coefp : T(0);
      if (identity) for (int i = tid; i < k; i += NTH) gs_out[b * (int64_t)k + i] = scp * grow[i];
      else for (int i = tid; i < k; i += NTH) crow[n_rows + i] = scp * grow[i];
      if (!lmi_active) {
        for (int idx = tid; idx < P; idx += NTH) crow[lmi_row0 + idx] = T(0);
        continue;
      }
      copy_S<T, NTH, HC>(A, head, prow + lmi_row0, r, P, tid);
      __syncthreads();
      if constexpr (HC > 0) head_phase<T, NTH, HC, true>(A, r, head, dd, ee, tt, vv, ww, red, tid);
      tridiagonalise<T, NTH, true>(A, r - HC, dd + HC, ee + HC, tt + HC, vv + HC, ww + HC, red, tid);
      top_eigenvector<T, NTH, HC>(A, r, kapp, head, dd, ee, tt, vv, ww, red, tid);
      for (int idx = tid; idx < P; idx += NTH) {
        int i = (int)((sqrtf(8.f * (float)idx + 1.f) - 1.f) * 0.5f);
        while ((i + 1) * (i + 2) / 2 <= idx) ++i;
        while (i * (i + 1) / 2 > idx) --i;
        const int j = idx - i * (i + 1) / 2;
        crow[lmi_row0 + idx] = -coefp * (i == j ? T(1) : T(2)) * zz[i] * zz[j];
      }
      continue;
    }
    for (int a = tid; a < n; a += NTH) vs[a] = v[b * ldv + a];
    T tv = T(0);
    for (int a = tid; a < n; a += NTH) {
      T acc;
      if (identity) {
        acc = grow[a];
      } else {
        T a0 = T(0), a1 = T(0), a2 = T(0), a3 = T(0);
        const T* col = nrm + a;
        int i = 0;
        for (; i + 3 < k; i += 4) {
          a0 = fma(col[(size_t)(i + 0) * n], grow[i + 0], a0);
          a1 = fma(col[(size_t)(i + 1) * n], grow[i + 1], a1);
          a2 = fma(col[(size_t)(i + 2) * n], grow[i + 2], a2);
          a3 = fma(col[(size_t)(i + 3) * n], grow[i + 3], a3);
        }
        for (; i < k; ++i) a0 = fma(col[(size_t)i * n], grow[i], a0);
        acc = (a0 + a1) + (a2 + a3);
      }
      ts[a] = acc;
      tv = fma(acc, v[b * ldv + a], tv);
    }
    tv = bsum<T, NW>(tv, red + 2 * 2 * kWaves, tid);         // (vs and ts are visible after this barrier)
    const T kap = kappa[b];
    const int aseg = active[2 * b], arow = active[2 * b + 1];
    T r_nrm = T(0), e_beta = T(0);
    if (old_mode) {
      T part = T(0);
      for (int a = tid; a < n; a += NTH) part = fma(vs[a], vs[a], part);
      r_nrm = sqrt(bsum<T, NW>(part, red + 5 * kWaves, tid));
      e_beta = exp(v[b * ldv + n]);
    }
    const bool clipped = old_mode ? (aseg >= 0 && r_nrm > T(0)) : (kap > T(1) && aseg >= 0);
    const T sc = old_mode ? (r_nrm > T(0) ? T(1) / (r_nrm * e_beta + kap) : T(0)) : T(1) / fmax(T(1), kap);
    const T coef = (clipped || old_mode) ? sc * sc * tv : T(0);        // (times grad kappa only where clipped: cu below)
    const T cu = clipped ? coef : T(0);
    const T cdir = old_mode && r_nrm > T(0) ? coef * e_beta / r_nrm : T(0);   // (times v: the old head's own term)
    if (old_mode && tid == 0) gv[b * ldgv + n] = -coef * r_nrm * e_beta;

    if (!(clipped && aseg == lmi_seg)) {                     // (the same for the whole workgroup)
      const int rho = clipped ? rho_of[arow] : -1;
      for (int a = tid; a < n; a += NTH) {
        const T u = rho >= 0 ? dt[(size_t)a * Mp + rho] : T(0);
        gv[b * ldgv + a] = fma(sc, ts[a], -(cu * u + cdir * vs[a]));
      }
      continue;
    }

    if constexpr (HC > 0) form_S_head<T, NTH, HC>(A, head, gt, vs, n, r, Pp, tid);
    else form_S<T, NTH>(A, gt, vs, n, P, Pp, tid);
    __syncthreads();
    if constexpr (HC > 0) head_phase<T, NTH, HC, true>(A, r, head, dd, ee, tt, vv, ww, red, tid);
    tridiagonalise<T, NTH, true>(A, r - HC, dd + HC, ee + HC, tt + HC, vv + HC, ww + HC, red, tid);

    top_eigenvector<T, NTH, HC>(A, r, kap, head, dd, ee, tt, vv, ww, red, tid);
    if constexpr (HC > 0) {
      // ---- d kappa / d v_a = x' G_a x, a wave per generator: lane l takes the columns l, l + 64, ... and walks down the rows
      // (the packed outer product of the plain kernel would not fit the LDS either)
      for (int a = wave; a < n; a += NW) {
        const T* ga = gt + (size_t)a * Pp;
        T acc = T(0);
        for (int j0 = 0; j0 < r; j0 += 64) {
          const int j = j0 + lane;
          const T xj = j < r ? zz[j] : T(0);
          T s0 = T(0), s1 = T(0), s2 = T(0), s3 = T(0);
          int i = j0;
          int Ti = i * (i + 1) / 2;
          for (; i + 3 < r; i += 4) {
            const int T1 = Ti + i + 1, T2 = T1 + i + 2, T3 = T2 + i + 3;
            const T g0 = j <= i ? ga[Ti + j] : T(0);
            const T g1 = j <= i + 1 ? ga[T1 + j] : T(0);
            const T g2 = j <= i + 2 ? ga[T2 + j] : T(0);
            const T g3 = j <= i + 3 ? ga[T3 + j] : T(0);
            s0 = fma(g0, (j == i ? T(1) : T(2)) * zz[i], s0);
            s1 = fma(g1, (j == i + 1 ? T(1) : T(2)) * zz[i + 1], s1);
            s2 = fma(g2, (j == i + 2 ? T(1) : T(2)) * zz[i + 2], s2);
            s3 = fma(g3, (j == i + 3 ? T(1) : T(2)) * zz[i + 3], s3);
            Ti = T3 + i + 4;
          }
          for (; i < r; ++i) {
            const T g0 = j <= i ? ga[Ti + j] : T(0);
            s0 = fma(g0, (j == i ? T(1) : T(2)) * zz[i], s0);
            Ti += i + 1;
          }
          acc = fma((s0 + s1) + (s2 + s3), xj, acc);
        }
        const T part_a = lw::wsum(acc);
        if (lane == 0) gv[b * ldgv + a] = fma(sc, ts[a], -(cu * part_a + cdir * vs[a]));
      }
      continue;
    }
    // ---- x_i x_j (twice off the diagonal) in packed order over the matrix storage
    for (int idx = tid; idx < P; idx += NTH) {
      int i = (int)((sqrtf(8.f * (float)idx + 1.f) - 1.f) * 0.5f);
      while ((i + 1) * (i + 2) / 2 <= idx) ++i;
      while (i * (i + 1) / 2 > idx) --i;
      const int j = idx - i * (i + 1) / 2;
      A[idx] = (i == j ? T(1) : T(2)) * zz[i] * zz[j];
    }
    __syncthreads();
    // ---- one contraction per generator, a wave each
    for (int a = wave; a < n; a += NW) {
      const T* col = gt + (size_t)a * Pp;
      T p0 = T(0), p1 = T(0), p2 = T(0), p3 = T(0);
      int idx = lane;
      for (; idx + 192 < P; idx += 256) {
        p0 = fma(col[idx], A[idx], p0);
        p1 = fma(col[idx + 64], A[idx + 64], p1);
        p2 = fma(col[idx + 128], A[idx + 128], p2);
        p3 = fma(col[idx + 192], A[idx + 192], p3);
      }
      for (; idx < P; idx += 64) p0 = fma(col[idx], A[idx], p0);
      const T part = lw::wsum((p0 + p1) + (p2 + p3));
      if (lane == 0) gv[b * ldgv + a] = fma(sc, ts[a], -(cu * part + cdir * vs[a]));
    }
  }
}

// ---------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------
// developer: workgroups per compute unit of the persistent grids, as a multiple of what fits at once (RAYEN_LB_GRID_MULT)
inline int grid_mult() {
  static const int v = [] {
    const char* env = std::getenv("RAYEN_LB_GRID_MULT");
    const int x = env != nullptr ? std::atoi(env) : 1;
    return x < 1 ? 1 : (x > 8 ? 8 : x);
  }();
  return v;
}

// The launch shape of a matrix: threads, columns kept in registers (0: the whole packed triangle in LDS), LDS bytes; nth = 0:
// no instance holds it.  This kernel waits -- on its barriers and on dependent LDS round trips -- and what fills the gaps
// is ANOTHER workgroup on the same compute unit: the smaller the workgroup, the more of them (16 waves per unit at these
// register counts), as long as every wave still has rows.  Measured, B = 2 000, forward ms with 128 / 256 / 512 threads
// (profiles/bench/r05_lmi_block_ab.txt; the wave-per-sample kernel takes 0.145 at r = 40 and 0.30 at r = 64):
//   r = 40 0.13 / 0.21 / 0.41, 64 0.26 / 0.38 / 0.68, 70 0.33 / 0.45 / 0.76, 100 1.18 / 0.82 / 1.26, 128 2.21 / 1.47 / 1.99,
//   150 - / 2.85 / 2.71, 180 - / 5.7 / 3.8, 196 - / 7.0 / 4.5;   512 against 1024 threads: 196 4.5 against 7.0 (two workgroups of
//   80.8 KB), 220 9.3 against 8.4 (7.3 with 24 columns in registers: two workgroups again), 250 11.7 against 10.7.
// Hence: 128 threads to r = 80, 256 to r = 140, 512 while TWO workgroups fit the LDS (the register columns stretch that to
// r = 220 in fp32), 1024 beyond.  RAYEN_LB_NTH / RAYEN_LB_512_UPTO (developer) override.
struct Plan { int nth = 0, hc = 0; size_t lds = 0; };

template <typename T>
Plan plan_for(int r, int n, bool bwd) {
  static const int upto = [] {
    const char* env = std::getenv("RAYEN_LB_512_UPTO");
    const int x = env != nullptr ? std::atoi(env) : 257;
    return x < 2 ? 2 : (x > 257 ? 257 : x);
  }();
  constexpr int HC = HeadCols<T>::value;
  const size_t plain = (bwd ? lds_bwd_elems(r, n) : lds_elems(r, n)) * sizeof(T);
  const size_t head = r > HC + 2 ? lds_elems_head(r, n, HC, bwd) * sizeof(T) : kLdsMax + 1;
  Plan p;
  if (r < 2 || (bwd && r > 320)) return p;
  static const int forced = [] {           // developer: RAYEN_LB_NTH=128 / 256 / 512 / 1024 wherever the rows fit
    const char* env = std::getenv("RAYEN_LB_NTH");
    return env != nullptr ? std::atoi(env) : 0;
  }();
  if ((forced == 128 || forced == 256) && r - 1 <= forced && plain <= kLdsMax) { p.nth = forced; p.hc = 0; p.lds = plain; return p; }
  // (every branch checks its LDS: vs[n] (+ ts[n] backward) of the fused route grows with n -- fp64 backward at r = 140 runs out
  // near n = 4 900 -- and such a shape must fall through to UNSUPPORTED, not fail at the launch)
  if (forced == 0 && r <= 80 && plain <= kLdsMax) { p.nth = 128; p.hc = 0; p.lds = plain; }
  else if (forced == 0 && r <= 140 && plain <= kLdsMax) { p.nth = 256; p.hc = 0; p.lds = plain; }
  else if (r <= upto && plain <= kLdsMax && (r <= 128 || 2 * plain <= kLdsMax)) { p.nth = 512; p.hc = 0; p.lds = plain; }
  else if (r <= upto && r - HC <= 224 && 2 * head <= kLdsMax) { p.nth = 512; p.hc = HC; p.lds = head; }   // (7 waves of rows + 1)
  else if (plain <= kLdsMax) { p.nth = 1024; p.hc = 0; p.lds = plain; }
  else if (head <= kLdsMax) { p.nth = 1024; p.hc = HC; p.lds = head; }
  return p;
}

// (eligibility of a shape, rayen_lmi_block.hip)
template <typename T>
int head_cols_fwd(int r, int n) { const Plan p = plan_for<T>(r, n, false); return p.nth == 0 ? -1 : p.hc; }
template <typename T>
int head_cols_bwd(int r, int n) { const Plan p = plan_for<T>(r, n, true); return p.nth == 0 ? -1 : p.hc; }

template <typename T>
bool lmi_block_serves_t(const LmiWaveImage* img) {
  return img != nullptr && plan_for<T>(img->r, img->n, false).nth != 0;
}

template <typename T>
bool lmi_block_bwd_serves_t(const LmiWaveImage* img) {
  return img != nullptr && plan_for<T>(img->r, img->n, true).nth != 0;
}

template <typename T, typename F>
void with_bwd_instance(const Plan& p, F f) {
  constexpr int HC = HeadCols<T>::value;
  if (p.hc > 0 && p.nth == 512) f(lmi_block_bwd_kernel<T, 512, HC>);
  else if (p.hc > 0) f(lmi_block_bwd_kernel<T, 1024, HC>);
  else if (p.nth == 128) f(lmi_block_bwd_kernel<T, 128, 0>);
  else if (p.nth == 256) f(lmi_block_bwd_kernel<T, 256, 0>);
  else if (p.nth == 512) f(lmi_block_bwd_kernel<T, 512, 0>);
  else f(lmi_block_bwd_kernel<T, 1024, 0>);
}

template <typename T, typename F>
void with_instance(const Plan& p, F f) {
  constexpr int HC = HeadCols<T>::value;
  if (p.hc > 0 && p.nth == 512) f(lmi_block_kernel<T, 512, HC>);
  else if (p.hc > 0) f(lmi_block_kernel<T, 1024, HC>);
  else if (p.nth == 128) f(lmi_block_kernel<T, 128, 0>);
  else if (p.nth == 256) f(lmi_block_kernel<T, 256, 0>);
  else if (p.nth == 512) f(lmi_block_kernel<T, 512, 0>);
  else f(lmi_block_kernel<T, 1024, 0>);
}

// called by rayen_pack_create (the only place that may touch function attributes)
template <typename T>
int lmi_block_prepare_t(const LmiWaveImage* img) {
  if (img == nullptr) return RAYEN_OK;
  bool ok = true;
  auto raise = [&](auto kern) {
    ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsMax) == hipSuccess;
  };
  if (lmi_block_serves_t<T>(img)) with_instance<T>(plan_for<T>(img->r, img->n, false), raise);
  if (lmi_block_bwd_serves_t<T>(img)) with_bwd_instance<T>(plan_for<T>(img->r, img->n, true), raise);
  // (the shapes of the products route: no copy of v in LDS)
  if (plan_for<T>(img->r, 0, false).nth != 0) with_instance<T>(plan_for<T>(img->r, 0, false), raise);
  if (plan_for<T>(img->r, 0, true).nth != 0) with_bwd_instance<T>(plan_for<T>(img->r, 0, true), raise);
  if (!ok) { (void)hipGetLastError(); return RAYEN_E_LAUNCH; }
  return RAYEN_OK;
}

template <typename T>
int lmi_block_forward_t(const RayenPack* p, const LmiWaveImage* img, const T* v, int64_t B, int64_t ldv, T* y, int64_t ldy,
                        T* kappa, int32_t* active, int32_t* nan_flag, hipStream_t stream, const T* kappa_in = nullptr,
                        int64_t ldk_in = 1, const T* prods = nullptr, int64_t ldt = 0, int old_mode = 0) {
  if (old_mode && prods != nullptr) return RAYEN_E_UNSUPPORTED;
  // (with products the kernel keeps no copy of v in LDS: the shape is planned for n = 0)
  const Plan plan = img != nullptr ? plan_for<T>(img->r, prods != nullptr ? 0 : img->n, false) : Plan();
  if (plan.nth == 0) return RAYEN_E_UNSUPPORTED;
  if (B == 0) return RAYEN_OK;
  const size_t lds = plan.lds;
  const int nth = plan.nth;
  int cus = 256;
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, p->device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
  }
  with_instance<T>(plan, [&](auto kern) {
    // persistent: as many workgroups as the chip holds at once (LDS, registers, threads)
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, nth, lds) != hipSuccess || per_cu < 1) {
      (void)hipGetLastError();
      per_cu = 1;
    }
    per_cu *= grid_mult();
    const int64_t grid = B < (int64_t)cus * per_cu ? B : (int64_t)cus * per_cu;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(nth), lds, stream, static_cast<const T*>(img->gt),
                       static_cast<const T*>(img->dt), static_cast<const T*>(img->nat), static_cast<const T*>(img->y0),
                       img->lin_id, img->r, img->n, img->k, img->m, img->P, img->Pp, img->Mp, img->Kp, img->identity,
                       img->lmi_seg, v, B, ldv, y, ldy, kappa, active, nan_flag, kappa_in, ldk_in, prods, ldt, img->lmi_row0,
                       img->n_rows, old_mode);
  });
  return hipGetLastError() == hipSuccess ? RAYEN_OK : RAYEN_E_LAUNCH;
}

template <typename T>
int lmi_block_backward_t(const RayenPack* p, const LmiWaveImage* img, const T* v, int64_t B, int64_t ldv, const T* kappa,
                         const int32_t* active, const T* gy, int64_t ldg, T* gv, int64_t ldgv, hipStream_t stream,
                         int only_lmi = 0, const T* prods = nullptr, int64_t ldt = 0, T* coeff = nullptr, int64_t ldc = 0,
                         T* gs_out = nullptr, int old_mode = 0) {
  if (old_mode && prods != nullptr) return RAYEN_E_UNSUPPORTED;
  const Plan plan = img != nullptr ? plan_for<T>(img->r, prods != nullptr ? 0 : img->n, true) : Plan();
  if (plan.nth == 0) return RAYEN_E_UNSUPPORTED;
  if (B == 0) return RAYEN_OK;
  const size_t lds = plan.lds;
  const int nth = plan.nth;
  int cus = 256;
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, p->device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
  }
  with_bwd_instance<T>(plan, [&](auto kern) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, nth, lds) != hipSuccess || per_cu < 1) {
      (void)hipGetLastError();
      per_cu = 1;
    }
    per_cu *= grid_mult();
    const int64_t grid = B < (int64_t)cus * per_cu ? B : (int64_t)cus * per_cu;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(nth), lds, stream, static_cast<const T*>(img->gt),
                       static_cast<const T*>(img->dt), static_cast<const T*>(img->nrm), img->rho_of, img->r, img->n, img->k,
                       img->P, img->Pp, img->Mp, img->identity, img->lmi_seg, v, B, ldv, kappa, active, gy, ldg, gv, ldgv, only_lmi,
                       prods, ldt, coeff, ldc, gs_out, img->lmi_row0, img->n_rows, old_mode);
  });
  return hipGetLastError() == hipSuccess ? RAYEN_OK : RAYEN_E_LAUNCH;
}

}  // namespace lb
}  // namespace rayen
