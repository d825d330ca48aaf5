// Grouping of a batch by active constraint for the matrix-core backward kernels (rayen_mfma_bwd.hip, rayen_mfma_bwd64.hip;
// private to librayen_hip.so).
#pragma once

#include "rayen_internal.h"

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
  for (int blk = threadIdx.x; blk < (int)gridDim.x; blk += 256)
    for (int i = 0; i < 32; ++i) table[blk][i] = ws[blk * 32 + i];
  __syncthreads();
  if (threadIdx.x < 32) {             // bucket threadIdx.x: samples of the blocks before this one, and of all blocks
    int before = 0, all = 0;
    for (int blk = 0; blk < (int)gridDim.x; ++blk) {
      const int c = table[blk][threadIdx.x];
      before += blk < (int)blockIdx.x ? c : 0;
      all += c;
    }
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

// bytes of workspace the bucketed walk of a pack with `n_dense` dense forms of `nkk` tiles each wants for a batch of
// B (0: plain walk).  The grouping costs two small launches (~10 us): it pays from four tiles of walk up (measured:
// 2 forms x 2 tiles 0.095 -> 0.080 ms, 6 x 2 0.173 -> 0.093 ms; 2 x 1 loses) and for batches that fill the chip.
inline int64_t bucket_workspace_bytes(int n_dense, int nkk, int64_t B) {
  if (n_dense < 2 || n_dense * nkk < 4 || n_dense + 2 > kMaxBuckets || B < 32768 || B > (int64_t)2000000000) return 0;
  return (int64_t)sizeof(int32_t) * (kWsHeader + B + 64 * (int64_t)(n_dense + 2));
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
