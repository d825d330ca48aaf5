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

#include "rayen_lmi_wave.h"

namespace rayen {
namespace lb {

constexpr int kWaves = 16;                   // slots of the reduction scratch (the largest workgroup's waves)
constexpr size_t kLdsMax = 160 * 1024;
// threads that share a row of the matrix, and threads of the workgroup: every row has its threads
__host__ __device__ inline int split_for(int r) { return r <= 128 ? 4 : 2; }
__host__ __device__ inline int threads_for(int r) { return r <= 256 ? 512 : 1024; }

// LDS of a workgroup (units of T): A[P] | dd[r] | ee[r] | vv[r] | ww[r] | red[3][2 kWaves] | vs[n]
__host__ __device__ inline size_t lds_elems(int r, int n) {
  return (size_t)r * (r + 1) / 2 + 4 * (size_t)r + 6 * kWaves + (size_t)n + 8;
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

template <typename T, int SPLIT, int NTH>
__global__ __launch_bounds__(NTH) void lmi_block_kernel(
    const T* __restrict__ gt, const T* __restrict__ dt, const T* __restrict__ nat, const T* __restrict__ y0,
    const int32_t* __restrict__ lin_id, int r, int n, int k, int m, int P, int Pp, int Mp, int Kp, int identity,
    int lmi_seg, const T* __restrict__ v, int64_t B, int64_t ldv, T* __restrict__ y, int64_t ldy,
    T* __restrict__ kappa_out, int32_t* __restrict__ active_out, int32_t* __restrict__ nan_flag) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lb_smem[];
  T* A = reinterpret_cast<T*>(lb_smem);
  T* dd = A + P;
  T* ee = dd + r;
  T* vv = ee + r;
  T* ww = vv + r;
  T* red = ww + r;            // [3][2 * kWaves]
  T* vs = red + 6 * kWaves;
  constexpr int kThreads = NTH, NW = NTH / 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  bool bad = false;

  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    __syncthreads();          // (the previous sample's last readers of vs / dd / ee)
    for (int a = tid; a < n; a += kThreads) vs[a] = v[b * ldv + a];
    __syncthreads();

    // ---- linear rows: (value, index among the linear rows) of the largest D_i . v; the lowest index wins a tie
    T kap = T(0);
    int who = -1;
    for (int i = tid; i < m; i += kThreads) {
      T acc = T(0);
      const T* col = dt + i;
      for (int a = 0; a < n; ++a) acc = fma(vs[a], col[(size_t)a * Mp], acc);
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

    // ---- S(v), lower triangle packed
    for (int idx = tid; idx < P; idx += kThreads) {
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
      A[idx] = (p0 + p1) + (p2 + p3);
    }
    __syncthreads();
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

    // ---- Householder reduction to tridiagonal form: dd (diagonal), ee (signed sub-diagonal)
    // Thread tid works on row i0 + tid / SPLIT, columns i0 + tid % SPLIT, + SPLIT, ...: SPLIT threads per row keep all eight
    // waves busy on matrices of fewer than 512 rows (one wave per SIMD reads LDS at a fraction of its rate).
    for (int kc = 0; kc + 2 < r; ++kc) {
      const int i0 = kc + 1, i = i0 + tid / SPLIT, part = tid % SPLIT;
      const bool has = i < r;
      const int Ti = i * (i + 1) / 2;
      T* rd = red + (kc & 1) * 2 * kWaves;
      const T x = has ? A[Ti + kc] : T(0);
      if (has && part == 0) vv[i] = x;
      const T sigma = bsum<T, NW>(part == 0 ? x * x : T(0), rd, tid);   // barrier 1 (vv = the raw column is visible too)
      const T x0 = vv[i0];
      const T below = sigma - x0 * x0;                               // what the reflector has to annihilate
      if (!(below > lw::Eps<T>::tiny * lw::Eps<T>::tiny)) {          // nothing to do: H = I  (the same for every thread)
        if (tid == 0) { dd[kc] = A[kc * (kc + 1) / 2 + kc]; ee[kc] = x0; }
        __syncthreads();
        continue;
      }
      const T alpha = (x0 >= T(0) ? T(-1) : T(1)) * sqrt(sigma);
      const T taup = T(1) / (sigma - x0 * alpha);
      // p = tau A v with v = x - alpha e_i0 (the raw column x is in vv; the correction is one extra term)
      T p = T(0), vi = T(0);
      if (i0 + (64 / SPLIT) * wave < r) {     // (a wave without a live row has nothing to sum)
        // the column index runs for the whole wave at once: lanes read A(i, j) = A[Ti + j] (j <= i: the row starts Ti are
        // distinct modulo 32 for 32 consecutive i) or A(j, i) = A[Tj + i] (j > i: consecutive words); v_j is a broadcast
        T q0 = T(0), q1 = T(0), q2 = T(0), q3 = T(0);
        const int ic = has ? i : r - 1;                      // (idle threads read a valid row; their sum is dropped)
        const int Tc = ic * (ic + 1) / 2;
        int j = i0 + part;
        int Tj = j * (j + 1) / 2;
        for (; j + 3 * SPLIT < r; j += 4 * SPLIT) {
          const int j1 = j + SPLIT, j2 = j + 2 * SPLIT, j3 = j + 3 * SPLIT;
          const int T1 = j1 * (j1 + 1) / 2, T2 = j2 * (j2 + 1) / 2, T3 = j3 * (j3 + 1) / 2;
          const T a0 = A[j <= ic ? Tc + j : Tj + ic];
          const T a1 = A[j1 <= ic ? Tc + j1 : T1 + ic];
          const T a2 = A[j2 <= ic ? Tc + j2 : T2 + ic];
          const T a3 = A[j3 <= ic ? Tc + j3 : T3 + ic];
          const T v0 = vv[j], v1 = vv[j1], v2 = vv[j2], v3 = vv[j3];
          q0 = fma(a0, v0, q0);
          q1 = fma(a1, v1, q1);
          q2 = fma(a2, v2, q2);
          q3 = fma(a3, v3, q3);
          const int j4 = j + 4 * SPLIT;
          Tj = j4 * (j4 + 1) / 2;
        }
        for (; j < r; j += SPLIT) q0 = fma(A[j <= ic ? Tc + j : j * (j + 1) / 2 + ic], vv[j], q0);
        T q = (q0 + q1) + (q2 + q3);
        if constexpr (SPLIT >= 2) q += __shfl_xor(q, 1);
        if constexpr (SPLIT >= 4) q += __shfl_xor(q, 2);
        if (has) {
          p = taup * (q - alpha * A[Ti + i0]);
          vi = i == i0 ? x - alpha : x;
        }
      }
      const T pv = bsum<T, NW>(part == 0 ? p * vi : T(0), rd + kWaves, tid);                // barrier 2
      const T K = T(0.5) * taup * pv;
      const T w = fma(-K, vi, p);
      if (has && part == 0) {
        ww[i] = w;
        if (i == i0) vv[i0] = vi;
      }
      __syncthreads();                                               // barrier 3
      // A -= v w' + w v' on the lower triangle of the live block
      if (has) {
        T* row = A + Ti;
        int j = i0 + part;
        for (; j + SPLIT <= i; j += 2 * SPLIT) {
          const T r0 = row[j], r1 = row[j + SPLIT];
          const T w0 = ww[j], w1 = ww[j + SPLIT], u0 = vv[j], u1 = vv[j + SPLIT];
          row[j] = r0 - (vi * w0 + w * u0);
          row[j + SPLIT] = r1 - (vi * w1 + w * u1);
        }
        if (j <= i) row[j] = row[j] - (vi * ww[j] + w * vv[j]);
      }
      if (tid == 0) { dd[kc] = A[kc * (kc + 1) / 2 + kc]; ee[kc] = alpha; }
      __syncthreads();                                               // barrier 4
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
    constexpr int kRounds = sizeof(T) == 4 ? (NTH == 1024 ? 3 : 4) : (NTH == 1024 ? 6 : 7);
    for (int round = 0; round < kRounds; ++round) {
      const T step = (hi - lo) * (T(1) / T(kThreads + 1));
      const T sig = lo + step * (T)(tid + 1);
      int cnt = 0;                                               // eigenvalues below sig = negative pivots of T - sig I
      T q = dd[0] - sig;
      cnt += q < T(0);
      for (int i = 1; i < r; ++i) {
        if (fabs(q) < floor_q) q = q < T(0) ? -floor_q : floor_q;
        const T e = ee[i - 1];
        q = dd[i] - sig - e * e / q;
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
    if (lam > kap) { kap = lam; aseg = lmi_seg; arow = 0; }

    const T scl = T(1) / fmax(T(1), kap);
    if (tid == 0) {
      if (kappa_out) kappa_out[b] = kap;
      if (active_out) { active_out[2 * b] = aseg; active_out[2 * b + 1] = arow; }
    }
    T* yrow = y + b * ldy;
    for (int i = tid; i < k; i += kThreads) {
      T val;
      if (identity) {
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

// ---------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------
template <typename T>
bool lmi_block_serves_t(const LmiWaveImage* img) {
  return img != nullptr && img->r >= 2 && lds_elems(img->r, img->n) * sizeof(T) <= kLdsMax;
}

template <typename T, typename F>
void with_instance(int r, F f) {
  if (r <= 128) f(lmi_block_kernel<T, 4, 512>, 512);
  else if (r <= 256) f(lmi_block_kernel<T, 2, 512>, 512);
  else f(lmi_block_kernel<T, 2, 1024>, 1024);
}

// called by rayen_pack_create (the only place that may touch function attributes)
template <typename T>
int lmi_block_prepare_t(const LmiWaveImage* img) {
  if (!lmi_block_serves_t<T>(img)) return RAYEN_OK;
  bool ok = true;
  with_instance<T>(img->r, [&](auto kern, int) {
    ok = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsMax) == hipSuccess;
  });
  if (!ok) { (void)hipGetLastError(); return RAYEN_E_LAUNCH; }
  return RAYEN_OK;
}

template <typename T>
int lmi_block_forward_t(const RayenPack* p, const LmiWaveImage* img, const T* v, int64_t B, int64_t ldv, T* y, int64_t ldy,
                        T* kappa, int32_t* active, int32_t* nan_flag, hipStream_t stream) {
  if (!lmi_block_serves_t<T>(img)) return RAYEN_E_UNSUPPORTED;
  if (B == 0) return RAYEN_OK;
  const size_t lds = lds_elems(img->r, img->n) * sizeof(T);
  int cus = 256;
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, p->device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
  }
  with_instance<T>(img->r, [&](auto kern, int nth) {
    // persistent: as many workgroups as the chip holds at once (LDS; 2048 threads per compute unit)
    int per_cu = (int)(kLdsMax / lds);
    const int by_threads = 2048 / nth;
    per_cu = per_cu < 1 ? 1 : (per_cu > by_threads ? by_threads : per_cu);
    const int64_t grid = B < (int64_t)cus * per_cu ? B : (int64_t)cus * per_cu;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(nth), lds, stream, static_cast<const T*>(img->gt),
                       static_cast<const T*>(img->dt), static_cast<const T*>(img->nat), static_cast<const T*>(img->y0),
                       img->lin_id, img->r, img->n, img->k, img->m, img->P, img->Pp, img->Mp, img->Kp, img->identity,
                       img->lmi_seg, v, B, ldv, y, ldy, kappa, active, nan_flag);
  });
  return hipGetLastError() == hipSuccess ? RAYEN_OK : RAYEN_E_LAUNCH;
}

}  // namespace lb
}  // namespace rayen
