// Grouping of a batch by active constraint for the matrix-core backward kernels (rayen_mfma_bwd.hip, rayen_mfma_bwd64.hip;
// private to librayen_hip.so).
#pragma once

#include "rayen_internal.h"
#include "rayen_mfma_kernel.h"

namespace rayen {

// ---------------------------------------------------------------------------------------------
// Bucketed walk.  grad kappa belongs to ONE constraint per sample, but a wave of 64 arbitrary samples meets nearly
// every segment, so the plain kernel evaluates S_s v for EVERY dense form s (config 3: 12 tiles; 0.122 of its
// 0.173 ms).  With the samples grouped by active segment a wave walks only its own form (2 tiles) -- or nothing,
// for samples clipped by a linear row or not clipped at all.  Three small launches ahead of the walk, all in a
// caller-provided workspace (no allocation, no host synchronisation):
//   1. bucket_count_kernel   bucket of every sample (0 none | 1 linear row | 2 + d dense form d) -> per-block counts; perm := -1
//   2. bucket_scatter_kernel offsets = running sum of the bucket totals rounded up to 64 (a wave never straddles two
//                            buckets); perm[offset + position] = sample (the order inside a block's share of a bucket
//                            is arbitrary, which no result depends on -- samples are independent)
//   3. the walk, reading and writing rows through perm (whole 4 n-byte rows: the gather costs no bandwidth)
// ---------------------------------------------------------------------------------------------
constexpr int kMaxBuckets = 30, kBucketBlocks = 256;
// workspace: int32 header [kBucketBlocks][32] per-block bucket counts | [kWsOffsets .. +32] padded bucket offsets; then perm
constexpr int kWsOffsets = kBucketBlocks * 32, kWsHeader = kWsOffsets + 64;

template <typename T>
__device__ __forceinline__ int bucket_of(const T kap, const int aseg, const int32_t* __restrict__ seg_bucket) {
  return (aseg < 0 || !(kap > T(1))) ? 0 : seg_bucket[aseg];
}

// Block `blk` owns the samples [blk chunk, (blk + 1) chunk).  No global atomics (same-address atomics on a handful of
// counters cost ~5 ns each and there would be thousands): the counts go to the block's own slots, and the scatter
// kernel rebuilds every block's starting position inside every bucket from them -- which also makes the permutation
// deterministic across blocks.
template <typename T>
__global__ __launch_bounds__(256) void bucket_count_kernel(const T* __restrict__ kappa,
                                                           const int32_t* __restrict__ active, int64_t B, int64_t chunk,
                                                           const int32_t* __restrict__ seg_bucket, int nb,
                                                           int32_t* __restrict__ ws) {
  __shared__ int cnt[32];
  if (threadIdx.x < 32) cnt[threadIdx.x] = 0;
  __syncthreads();
  int32_t* perm = ws + kWsHeader;
  const int64_t total = B + 64 * (int64_t)nb;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) perm[i] = -1;
  const int64_t lo = blockIdx.x * chunk, hi = (lo + chunk < B) ? lo + chunk : B;
  const int lane = threadIdx.x & 63;
  for (int64_t s0 = lo + (threadIdx.x & ~63); s0 < hi; s0 += 256) {   // wave-uniform trip count
    const int64_t s = s0 + lane;
    const int b = s < hi ? bucket_of(kappa[s], active[2 * s], seg_bucket) : -1;
    for (int i = 0; i < nb; ++i) {    // one LDS atomic per wave and bucket, not per sample
      const int c = __popcll(__ballot(b == i));
      if (lane == 0 && c) atomicAdd(&cnt[i], c);
    }
  }
  __syncthreads();
  if (threadIdx.x < 32) ws[blockIdx.x * 32 + threadIdx.x] = cnt[threadIdx.x];
}

template <typename T>
__global__ __launch_bounds__(256) void bucket_scatter_kernel(const T* __restrict__ kappa,
                                                             const int32_t* __restrict__ active, int64_t B, int64_t chunk,
                                                             const int32_t* __restrict__ seg_bucket, int nb,
                                                             int32_t* __restrict__ ws) {
  __shared__ int table[kBucketBlocks][33];
  __shared__ int cursor[32];
  for (int e = threadIdx.x; e < (int)gridDim.x * 32; e += 256) table[e >> 5][e & 31] = ws[e];   // (coalesced)
  __syncthreads();
  // bucket i: samples of the blocks before this one, and of all blocks.  Eight partial sums of 32 blocks each per bucket
  // (thread = (part, bucket)), then eight additions: the one-thread-per-bucket loop over all 256 blocks that stood here
  // was 256 dependent LDS round trips -- most of this launch's 10.7 us on config 3 (round 4)
  __shared__ int part_before[8][33], part_all[8][33];
  {
    const int i = threadIdx.x & 31, part = threadIdx.x >> 5;
    int before = 0, all = 0;
    for (int blk = part * 32; blk < part * 32 + 32 && blk < (int)gridDim.x; ++blk) {
      const int c = table[blk][i];
      before += blk < (int)blockIdx.x ? c : 0;
      all += c;
    }
    part_before[part][i] = before;
    part_all[part][i] = all;
  }
  __syncthreads();
  if (threadIdx.x < 32) {
    int before = 0, all = 0;
#pragma unroll
    for (int part = 0; part < 8; ++part) { before += part_before[part][threadIdx.x]; all += part_all[part][threadIdx.x]; }
    table[0][threadIdx.x] = before;   // (row 0 is dead now: every thread has read it)
    table[1][threadIdx.x] = all;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int i = 0; i < nb; ++i) {
      cursor[i] = run + table[0][i];  // this block's first slot in bucket i
      if (blockIdx.x == 0) ws[kWsOffsets + i] = run;
      run += (table[1][i] + 63) & ~63;   // buckets start on wave boundaries
    }
    if (blockIdx.x == 0) ws[kWsOffsets + nb] = run;
  }
  __syncthreads();
  const int64_t lo = blockIdx.x * chunk, hi = (lo + chunk < B) ? lo + chunk : B;
  const int lane = threadIdx.x & 63;
  for (int64_t s0 = lo + (threadIdx.x & ~63); s0 < hi; s0 += 256) {
    const int64_t s = s0 + lane;
    const int b = s < hi ? bucket_of(kappa[s], active[2 * s], seg_bucket) : -1;
    for (int i = 0; i < nb; ++i) {
      const unsigned long long m = __ballot(b == i);
      if (m == 0) continue;
      int first = 0;
      if (lane == 0) first = atomicAdd(&cursor[i], __popcll(m));
      first = __shfl(first, 0);
      if (b == i) ws[kWsHeader + first + __popcll(m & ((1ull << lane) - 1ull))] = (int32_t)s;
    }
  }
}

// rows of a [B, ld] matrix chosen by `rowix` (this lane's entry of the group's 64 row numbers, -1 = none) ->
// B-operand registers, like load_rows: whole rows through the patch when they are full lines, else 16-byte pieces
template <int NT, int NK, int LSTR>
__device__ __forceinline__ void load_rows_ix(float (&dst)[NT][NK * 16], const float* __restrict__ src, int64_t ld,
                                             int width, int vec, const int rowix, float (*patch)[LSTR], int lane) {
  constexpr int NQ = NK * 4;
  const int col = lane & 31, hi = lane >> 5;
  if (vec && width == NK * 32) {
    f32x4 piece[NT][NQ];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int j = 0; j < NQ; ++j) {
        const int idx = lane + 64 * j;
        const int s = __shfl(rowix, t * 32 + idx / (NK * 8));
        piece[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (s >= 0) piece[t][j] = *reinterpret_cast<const f32x4*>(src + (int64_t)s * ld + 4 * (idx % (NK * 8)));
      }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int j = 0; j < NQ; ++j) {
        const int idx = lane + 64 * j;
        *reinterpret_cast<f32x4*>(&patch[idx / (NK * 8)][4 * (idx % (NK * 8))]) = piece[t][j];
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(&patch[col][8 * q + 4 * hi]);
        dst[t][4 * q + 0] = x[0];
        dst[t][4 * q + 1] = x[1];
        dst[t][4 * q + 2] = x[2];
        dst[t][4 * q + 3] = x[3];
      }
      __builtin_amdgcn_wave_barrier();
    }
  } else {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int s = __shfl(rowix, t * 32 + col);
      const float* row = src + (int64_t)(s >= 0 ? s : 0) * ld;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int c0 = 8 * q + 4 * hi;
#pragma unroll
        for (int c = 0; c < 4; ++c) dst[t][4 * q + c] = (s >= 0 && c0 + c < width) ? row[c0 + c] : 0.f;
      }
    }
  }
}

template <int NT, int NK, int LSTR>
__device__ __forceinline__ void store_rows_ix(const float (&val)[NT][NK * 16], float* __restrict__ dst, int64_t ld,
                                              int width, int vec, const int rowix, float (*patch)[LSTR], int lane) {
  constexpr int NQ = NK * 4;
  const int col = lane & 31, hi = lane >> 5;
  if (vec && width == NK * 32) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        *reinterpret_cast<f32x4*>(&patch[col][8 * q + 4 * hi]) =
            f32x4{val[t][4 * q], val[t][4 * q + 1], val[t][4 * q + 2], val[t][4 * q + 3]};
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int j = 0; j < NQ; ++j) {
        const int idx = lane + 64 * j;
        const int s = __shfl(rowix, t * 32 + idx / (NK * 8));
        const f32x4 o = *reinterpret_cast<const f32x4*>(&patch[idx / (NK * 8)][4 * (idx % (NK * 8))]);
        if (s >= 0) *reinterpret_cast<f32x4*>(dst + (int64_t)s * ld + 4 * (idx % (NK * 8))) = o;
      }
      __builtin_amdgcn_wave_barrier();
    }
  } else {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int s = __shfl(rowix, t * 32 + col);
      if (s < 0) continue;
      float* row = dst + (int64_t)s * ld;
#pragma unroll
      for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (8 * q + 4 * hi + c < width) row[8 * q + 4 * hi + c] = val[t][4 * q + c];
    }
  }
}

// bytes of workspace the bucketed walk of a pack with `n_dense` dense forms of `nkk` tiles each wants for a batch of
// B (0: plain walk).  The grouping costs two small launches (~10 us): it pays from four tiles of walk up (measured:
// 2 forms x 2 tiles 0.095 -> 0.080 ms, 6 x 2 0.173 -> 0.093 ms; 2 x 1 loses) and for batches that fill the chip.
inline int64_t bucket_workspace_bytes(int n_dense, int nkk, int64_t B) {
  if (n_dense < 2 || n_dense * nkk < 4 || n_dense + 2 > kMaxBuckets || B < 32768 || B > (int64_t)2000000000) return 0;
  return (int64_t)sizeof(int32_t) * (kWsHeader + B + 64 * (int64_t)(n_dense + 2));
}
// the same for a walk made of `n_groups` item groups (dense forms and packed tile pairs) with `n_tiles` tiles in all
inline int64_t bucket_workspace_bytes_groups(int n_groups, int n_tiles, int64_t B) {
  if (n_groups < 2 || n_tiles < 4 || n_groups + 2 > kMaxBuckets || B < 32768 || B > (int64_t)2000000000) return 0;
  return (int64_t)sizeof(int32_t) * (kWsHeader + B + 64 * (int64_t)(n_groups + 2));
}

// seg -> bucket table of a pack: 1 = linear rows, 2 + d = the d-th segment with a dense form
template <typename IsDense>
inline std::vector<int32_t> bucket_table(const RayenPack* p, IsDense is_dense, int* n_dense) {
  std::vector<int32_t> table(p->segs.size() + 1, 0);
  *n_dense = 0;
  for (size_t s = 0; s < p->segs.size(); ++s) table[s] = is_dense(p->segs[s]) ? 2 + (*n_dense)++ : 1;
  return table;
}

template <typename T>
inline void launch_bucket_sort(const T* kappa, const int32_t* active, int64_t B, const int32_t* seg_bucket, int nb,
                               int32_t* ws, hipStream_t stream) {
  const int64_t chunk = ((B + kBucketBlocks - 1) / kBucketBlocks + 255) / 256 * 256;
  const unsigned blocks = (unsigned)((B + chunk - 1) / chunk);
  hipLaunchKernelGGL(bucket_count_kernel<T>, dim3(blocks), dim3(256), 0, stream, kappa, active, B, chunk, seg_bucket, nb, ws);
  hipLaunchKernelGGL(bucket_scatter_kernel<T>, dim3(blocks), dim3(256), 0, stream, kappa, active, B, chunk, seg_bucket, nb, ws);
}

}  // namespace rayen
