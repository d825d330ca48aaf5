// fp32 MFMA path, host side: eligibility, image construction, dispatch; plus the experimental
// split-operand kernel.  The forward kernel itself lives in rayen_mfma_kernel.h.
#include "rayen_mfma_kernel.h"

#include <cstring>
#include <vector>

namespace rayen {

// ---------------------------------------------------------------------------------------------
// Split-operand variant: the same tile walk on v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate).
// Every fp32 operand is split exactly into three bf16 pieces x = x1 + x2 + x3 (8 significant bits
// each, bf16 has fp32's exponent range, so no scaling is involved); the product is rebuilt from
// the six piece products of order <= 2^-16 (x1y1, x1y2, x2y1, x1y3, x2y2, x3y1), each exact in the
// fp32 accumulator.  The dropped terms are <= 2^-24 relative, i.e. below the rounding error of an
// fp32 FMA chain: the result is fp32-grade at 6/16 of the fp32 MFMA time.  A operands (three bf16
// images of W, 12 KiB per tile at n = 64) are staged through LDS once per workgroup per tile --
// the workgroup's waves walk the tiles in lockstep, one barrier per tile (two 4-wave workgroups
// share a CU so that one can compute while the other sits at a barrier or a group boundary) --
// because at this MFMA rate the per-wave L2 stream of the fp32 kernel would exceed the L1 bandwidth.
// EXPERIMENTAL (opt-in with RAYEN_SPLIT_BF16=1 at pack creation): parity-tested like the fp32
// kernel, but with the matrix work 2.7x cheaper the group boundaries and epilogues dominate and the
// measured gain is only 5-25 %; DESIGN.md lists what it takes to turn it into the default.
// ---------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#ifndef RAYEN_SPLIT_WAVES
#define RAYEN_SPLIT_WAVES 4
#endif
constexpr int kSplitWaves = RAYEN_SPLIT_WAVES;
constexpr int kSplitStage = (12 + kSplitWaves - 1) / kSplitWaves;  // chunks a wave stages per tile (<= 12 chunks)

template <int NKK, bool TRACK>
__global__ __launch_bounds__(kSplitWaves * 64, 2) void mfma_split_kernel(
    const bf16x8* __restrict__ Wb, const MItem* __restrict__ items, int n_items,
    const MPack* __restrict__ packs, const float* __restrict__ y0, int identity, int k, int n,
    const float* __restrict__ v, int64_t B, int64_t ldv, int vec_in, float* __restrict__ y, int64_t ldy,
    int vec_out, float* __restrict__ kappa_out, int32_t* __restrict__ active_out,
    int32_t* __restrict__ nan_flag) {
  constexpr int NT = 2, NQ = NKK * 4, NS = NKK * 2, NCH = NS * 3;  // NS K-steps of 16, NCH 1-KiB chunks per tile
  constexpr int kMfmaWaves = kSplitWaves;                          // (the shared epilogue text indexes aux_lds by wave)
  __shared__ float aux_lds[kSplitWaves][NT][32][32];               // [wave][sample tile][aux row][sample]
  __shared__ bf16x8 a_lds[2][NCH][64];                             // two tiles of A fragments, [chunk][lane]

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int col = lane & 31;
  const int hi = lane >> 5;
  const int64_t n_groups = (B + NT * 32 - 1) / (NT * 32);
  const int64_t groups_per_round = (int64_t)gridDim.x * kSplitWaves;
  const int64_t rounds = (n_groups + groups_per_round - 1) / groups_per_round;
  bool bad = false;
  // every wave of a workgroup runs the same number of rounds (the tile loop holds barriers);
  // a wave whose group index is past the end just carries dead samples
  for (int64_t round = 0; round < rounds; ++round) {
  const int64_t grp = (round * gridDim.x + blockIdx.x) * kSplitWaves + wave;
  const int64_t s_base = grp * (NT * 32);

  // ---- this lane's half of v: fp32 (epilogues, output) and split into bf16 pieces
  // vb[t][piece][k-step] = 8 elements = the B operand of one MFMA
  float vr[NT][NKK * 16];
  bf16x8 vb[NT][3][NS];
  bool live[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int64_t s = s_base + t * 32 + col;
    live[t] = s < B;
    const float* row = v + (live[t] ? s : 0) * ldv;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int c0 = 8 * q + 4 * hi;
      f32x4 x = {0.f, 0.f, 0.f, 0.f};
      if (live[t]) {
        if (vec_in && c0 + 3 < n) {
          x = *reinterpret_cast<const f32x4*>(row + c0);
        } else {
          if (c0 + 0 < n) x[0] = row[c0 + 0];
          if (c0 + 1 < n) x[1] = row[c0 + 1];
          if (c0 + 2 < n) x[2] = row[c0 + 2];
          if (c0 + 3 < n) x[3] = row[c0 + 3];
        }
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int i = (q & 1) * 4 + c;  // position inside K-step q >> 1
        const __bf16 p1 = (__bf16)x[c];
        const float r1 = x[c] - (float)p1;
        const __bf16 p2 = (__bf16)r1;
        const float r2 = r1 - (float)p2;
        vr[t][4 * q + c] = x[c];
        vb[t][0][q >> 1][i] = p1;
        vb[t][1][q >> 1][i] = p2;
        vb[t][2][q >> 1][i] = (__bf16)r2;
      }
    }
  }
  auto vget = [&](int t, int idx) -> float { return vr[t][idx]; };

  float kap[NT], part[NT], scale[NT];
  int aseg[NT], arow[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) { kap[t] = 0.f; part[t] = 0.f; scale[t] = 1.f; aseg[t] = -1; arow[t] = 0; }

  // ---- A staging: wave w carries chunks w, w + #waves, ... of a tile from global memory into LDS
  bf16x8 stage[kSplitStage];
  auto stage_load = [&](int tile) {
    const bf16x8* src = Wb + (size_t)tile * NCH * 64 + lane;
#pragma unroll
    for (int j = 0; j < kSplitStage; ++j)
      if (wave + kSplitWaves * j < NCH) stage[j] = src[(size_t)(wave + kSplitWaves * j) * 64];
  };
  auto stage_store = [&](int buf) {
#pragma unroll
    for (int j = 0; j < kSplitStage; ++j)
      if (wave + kSplitWaves * j < NCH) a_lds[buf][wave + kSplitWaves * j][lane] = stage[j];
  };
  stage_load(0);
  stage_store(0);
  if (n_items > 1) stage_load(1);
  __syncthreads();

  auto finish_kappa = [&]() {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float other = xhalf(kap[t]);
      if (TRACK) {
        const int oseg = __shfl_xor(aseg[t], 32), orow = __shfl_xor(arow[t], 32);
        if (other > kap[t] || (other == kap[t] && hi == 1)) { aseg[t] = oseg; arow[t] = orow; }
      }
      kap[t] = fmaxf(kap[t], other);
      scale[t] = 1.0f / fmaxf(1.0f, kap[t]);
    }
  };

  f32x16 acc[NT];
  for (int it = 0; it < n_items; ++it) {
    const int buf = it & 1;
    // tile it+1 (loaded into `stage` one tile ago) goes to the other LDS buffer, whose last readers
    // all passed the barrier that closed tile it-1; tile it+2 starts its trip from L2
    if (it + 1 < n_items) stage_store(buf ^ 1);
    if (it + 2 < n_items) stage_load(it + 2);
    const MItem item = items[it];
    if (item.type != MI_NOP) {
    if (item.type == MI_OUT && (item.flags & MF_FIRST)) finish_kappa();
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[t][g] = 0.f;
#pragma unroll
    for (int sp = 0; sp < NS; ++sp) {
      if (2 * sp < item.qbegin) continue;  // 32-column blocks folded into their transpose (wave-uniform)
      const bf16x8 a1 = a_lds[buf][sp * 3 + 0][lane];
      const bf16x8 a2 = a_lds[buf][sp * 3 + 1][lane];
      const bf16x8 a3 = a_lds[buf][sp * 3 + 2][lane];
      // smallest products first
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, vb[t][0][sp], acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, vb[t][1][sp], acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, vb[t][2][sp], acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, vb[t][0][sp], acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, vb[t][1][sp], acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, vb[t][0][sp], acc[t], 0, 0, 0);
    }
    if (item.type == MI_LIN) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (TRACK) {
#pragma unroll
          for (int g = 0; g < 16; ++g)
            if (acc[t][g] > kap[t]) {
              kap[t] = acc[t][g];
              aseg[t] = item.seg;
              arow[t] = item.row0 + (g & 3) + 8 * (g >> 2) + 4 * hi;
            }
        } else {
#pragma unroll
          for (int g = 0; g < 16; ++g) kap[t] = fmaxf(kap[t], acc[t][g]);
        }
      }
    } else if (item.type == MI_AUX) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 16; ++g)
          aux_lds[wave][t][(g & 3) + 8 * (g >> 2) + 4 * hi][col] = acc[t][g];
      __builtin_amdgcn_wave_barrier();
    } else if (item.type == MI_OUT) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (!live[t]) continue;
        float* yrow = y + (s_base + t * 32 + col) * ldy;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const int r0 = item.row0 + 8 * a + 4 * hi;
          if (r0 >= k) continue;
          f32x4 o;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            o[c] = fmaf(acc[t][4 * a + c], scale[t], y0[r0 + c]);  // y0 is padded to a tile multiple
            bad |= (o[c] != o[c]) && (r0 + c < k);
          }
          if (vec_out && r0 + 3 < k) {
            *reinterpret_cast<f32x4*>(yrow + r0) = o;
          } else {
#pragma unroll
            for (int c = 0; c < 4; ++c)
              if (r0 + c < k) yrow[r0 + c] = o[c];
          }
        }
      }
    } else if (item.type == MI_PACK) {
      // eight small factor segments in one tile: ||U v||^2 of each is a 4-register sum
      const MPack pk = packs[item.aux];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int slot = hi ? pk.aux[a][1] : pk.aux[a][0];
        const int sid = hi ? pk.seg[a][1] : pk.seg[a][0];
        const bool pair = (item.row0 >> a) & 1;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          float qs = acc[t][4 * a] * acc[t][4 * a];
#pragma unroll
          for (int c = 1; c < 4; ++c) qs = fmaf(acc[t][4 * a + c], acc[t][4 * a + c], qs);
          if (pair) qs += xhalf(qs);
          const float kc = aux_lds[wave][t][slot & 31][col] + sqrtf(qs);
          if (sid >= 0 && kc > kap[t]) { kap[t] = kc; aseg[t] = sid; arow[t] = 0; }
        }
      }
    } else {
      // QSYM / QFAC / SOC: a running sum over the segment's tiles, closed on its last tile
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        float sum = (item.flags & MF_FIRST) ? 0.f : part[t];
        if (item.flags & MF_SYM) {
          // radicand v'Gv = sum_j (G v)_j v_j ; v_j of row tile tp is register 16*tp+g of vr
#pragma unroll
          for (int tp = 0; tp < NKK; ++tp)
            if (item.row0 == tp) {
#pragma unroll
              for (int g = 0; g < 16; ++g) sum = fmaf(acc[t][g], vget(t, 16 * tp + g), sum);
            }
        } else {
#pragma unroll
          for (int g = 0; g < 16; ++g) sum = fmaf(acc[t][g], acc[t][g], sum);
        }
        part[t] = sum;
      }
      if (item.flags & MF_LAST) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const float total = part[t] + xhalf(part[t]);
          const float a0 = aux_lds[wave][t][item.aux][col];
          float kc;
          if (item.type != MI_SOC) {
            kc = a0 + sqrtf(fmaxf(total, 0.f));
          } else {
            // a' x^2 + b' x + c' = 0  (rayen/constraint_module.py:392-396, 339-348), a' < 0
            const float br = aux_lds[wave][t][item.aux + 1][col];
            const float cp = total - a0 * a0;
            const float bp = 2.f * br - 2.f * a0 * item.f0;
            const float disc = bp * bp - 4.f * item.f1 * cp;
            kc = 0.f;
            if (disc >= 0.f) {
              const float root = sqrtf(disc);
              const float inv2a = 0.5f / item.f1;
              kc = fmaxf((-bp - root) * inv2a, (-bp + root) * inv2a);
            }
          }
          if (kc > kap[t]) { kap[t] = kc; aseg[t] = item.seg; arow[t] = 0; }
        }
      }
    }
    }  // not a filler tile
    __syncthreads();  // everyone is done reading a_lds[buf] and writing a_lds[buf ^ 1]
  }

  if (identity) {
    finish_kappa();
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (!live[t]) continue;
      float* yrow = y + (s_base + t * 32 + col) * ldy;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int c0 = 8 * q + 4 * hi;
        if (c0 >= k) continue;
        f32x4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          o[c] = fmaf(vget(t, 4 * q + c), scale[t], y0[c0 + c]);
          bad |= (o[c] != o[c]) && (c0 + c < k);
        }
        if (vec_out && c0 + 3 < k) {
          *reinterpret_cast<f32x4*>(yrow + c0) = o;
        } else {
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (c0 + c < k) yrow[c0 + c] = o[c];
        }
      }
    }
  }

  if (hi == 0) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (!live[t]) continue;
      const int64_t s = s_base + t * 32 + col;
      if (kappa_out) kappa_out[s] = kap[t];
      if (TRACK) { active_out[2 * s] = aseg[t]; active_out[2 * s + 1] = arow[t]; }
    }
  }
  }  // rounds
  if (nan_flag && bad) atomicOr(nan_flag, 1);
}

// ---------------------------------------------------------------------------------------------
// host: eligibility, image construction, launch
// ---------------------------------------------------------------------------------------------


bool mfma_eligible(const RayenPack* p) {
  if (p->n > 128) return false;  // v lives in registers: n_pad/2 VGPRs per sample tile
  for (const RayenSegment& g : p->segs)
    if (g.type == RAYEN_SEG_LMI) return false;  // eigen-solve epilogue lives on the generic path
  TileLayout b(p->n);
  if (layout_tiles(p, b, /*allow_pack=*/true) != RAYEN_OK || b.items.empty()) return false;
  // 32-row tiles must be reasonably full, and columns not mostly padding; otherwise the
  // 8-row generic path wastes less
  const int64_t padded = (int64_t)b.items.size() * 32;
  return b.useful_rows * 2 >= padded && p->n * 2 >= b.n_pad;
}

int mfma_build(const RayenPack* p, MfmaImage** out, int64_t* bytes) {
  TileLayout b(p->n);
  const int rc = layout_tiles(p, b, /*allow_pack=*/true);
  if (rc != RAYEN_OK) return rc;
  if (b.items.size() % 2) {  // the kernel walks tiles in pairs
    MItem it;
    std::memset(&it, 0, sizeof(it));
    it.type = MI_NOP;
    b.items.push_back(it);
    b.add_tile({}, p->n);
  }
  b.add_tile({}, p->n);  // spare tile: the prefetch runs one tile past the end (no item refers to it)
  const std::vector<float> frag = b.fragments_f32();
  if (b.packs.empty()) b.packs.push_back(MPack());

  MfmaImage* img = new MfmaImage();
  img->nkk = b.n_pad / 32;
  img->identity = p->out_identity;
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, p->device) == hipSuccess && prop.multiProcessorCount > 0) {
      img->n_simd = prop.multiProcessorCount * 4;
      img->n_cu = prop.multiProcessorCount;
    }
  }
  img->n_items = (int)b.items.size();
  const int k_tiles = (p->k + 31) / 32;
  std::vector<float> y0((size_t)k_tiles * 32 + 32, 0.f);
  for (int i = 0; i < p->k; ++i) y0[i] = (float)p->y0[i];
  bool ok = hipMalloc(&img->W, frag.size() * sizeof(float)) == hipSuccess &&
            hipMemcpy(img->W, frag.data(), frag.size() * sizeof(float), hipMemcpyHostToDevice) == hipSuccess &&
            hipMalloc(&img->y0, y0.size() * sizeof(float)) == hipSuccess &&
            hipMemcpy(img->y0, y0.data(), y0.size() * sizeof(float), hipMemcpyHostToDevice) == hipSuccess &&
            hipMalloc(&img->items, b.items.size() * sizeof(MItem)) == hipSuccess &&
            hipMemcpy(img->items, b.items.data(), b.items.size() * sizeof(MItem), hipMemcpyHostToDevice) == hipSuccess &&
            hipMalloc(&img->packs, b.packs.size() * sizeof(MPack)) == hipSuccess &&
            hipMemcpy(img->packs, b.packs.data(), b.packs.size() * sizeof(MPack), hipMemcpyHostToDevice) == hipSuccess;
  if (ok && p->split_bf16 && img->nkk <= 2) {
    // three bf16 pieces of every entry, in the fragment order of v_mfma_f32_32x32x16_bf16:
    // chunk (tile, k-step s, piece p) = 64 lanes x 8 elements, element i of lane l = column
    // 16s + 8(i>>2) + 4(l>>5) + (i&3) of row l&31, i.e. entry [2s + (i>>2)][l][i&3] of the fp32 image
    auto rne = [](float x) -> uint16_t {
      uint32_t u;
      std::memcpy(&u, &x, 4);
      u += 0x7FFFu + ((u >> 16) & 1u);
      return (uint16_t)(u >> 16);
    };
    auto widen = [](uint16_t h) -> float {
      const uint32_t u = (uint32_t)h << 16;
      float x;
      std::memcpy(&x, &u, 4);
      return x;
    };
    const int n_tiles = (int)b.items.size(), ns = b.nq() / 2;
    std::vector<uint16_t> wb((size_t)n_tiles * ns * 3 * 64 * 8);
    for (int t = 0; t < n_tiles; ++t)
      for (int sp = 0; sp < ns; ++sp)
        for (int l = 0; l < 64; ++l)
          for (int i = 0; i < 8; ++i) {
            const float x = frag[(((size_t)t * b.nq() + 2 * sp + (i >> 2)) * 64 + l) * 4 + (i & 3)];
            const uint16_t h1 = rne(x);
            const float r1 = x - widen(h1);
            const uint16_t h2 = rne(r1);
            const float r2 = r1 - widen(h2);
            const uint16_t h3 = rne(r2);
            const size_t base = (((size_t)t * ns + sp) * 3) * 64 * 8 + (size_t)l * 8 + i;
            wb[base] = h1;
            wb[base + 64 * 8] = h2;
            wb[base + 2 * 64 * 8] = h3;
          }
    ok = hipMalloc(&img->Wb, wb.size() * 2) == hipSuccess &&
         hipMemcpy(img->Wb, wb.data(), wb.size() * 2, hipMemcpyHostToDevice) == hipSuccess;
    if (ok) img->bytes += (int64_t)wb.size() * 2;
  }
  if (!ok) { mfma_free(img); return RAYEN_E_ALLOC; }
  img->bytes += (int64_t)(frag.size() * sizeof(float) + y0.size() * sizeof(float) +
                         b.items.size() * sizeof(MItem) + b.packs.size() * sizeof(MPack));
  *bytes = img->bytes;
  *out = img;
  return RAYEN_OK;
}

void mfma_free(MfmaImage* img) {
  if (img == nullptr) return;
  if (img->W) (void)hipFree(img->W);
  if (img->items) (void)hipFree(img->items);
  if (img->packs) (void)hipFree(img->packs);
  if (img->Wb) (void)hipFree(img->Wb);
  if (img->y0) (void)hipFree(img->y0);
  delete img;
}

template <int NKK>
static int launch_split(const RayenPack* p, const MfmaImage* img, const float* v, int64_t B, int64_t ldv,
                        float* y, int64_t ldy, float* kappa, int32_t* active, int32_t* nan_flag,
                        hipStream_t stream) {
  // one workgroup (8 waves x 64 samples) per CU, every workgroup the same number of rounds
  const int64_t blocks_needed = (B + kSplitWaves * 64 - 1) / (kSplitWaves * 64);
  const int64_t slots = (int64_t)img->n_cu * (8 / kSplitWaves);  // two 4-wave workgroups share a CU
  const int64_t rounds = (blocks_needed + slots - 1) / slots;
  const int64_t grid = (blocks_needed + rounds - 1) / rounds;
  const int vec_in = (ldv % 4 == 0) && ((reinterpret_cast<uintptr_t>(v) & 15) == 0);
  const int vec_out = (ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0);
  const bf16x8* wb = static_cast<const bf16x8*>(img->Wb);
  if (active != nullptr) {
    hipLaunchKernelGGL((mfma_split_kernel<NKK, true>), dim3((unsigned)grid), dim3(kSplitWaves * 64), 0, stream,
                       wb, img->items, img->n_items, img->packs, img->y0, img->identity, p->k, p->n, v, B, ldv,
                       vec_in, y, ldy, vec_out, kappa, active, nan_flag);
  } else {
    hipLaunchKernelGGL((mfma_split_kernel<NKK, false>), dim3((unsigned)grid), dim3(kSplitWaves * 64), 0, stream,
                       wb, img->items, img->n_items, img->packs, img->y0, img->identity, p->k, p->n, v, B, ldv,
                       vec_in, y, ldy, vec_out, kappa, active, nan_flag);
  }
  return hipGetLastError() == hipSuccess ? RAYEN_OK : RAYEN_E_LAUNCH;
}

int mfma_forward(const RayenPack* p, const MfmaImage* img, const float* v, int64_t B, int64_t ldv, float* y,
                 int64_t ldy, float* kappa, int32_t* active, int32_t* nan_flag, int old_mode,
                 hipStream_t stream) {
  if (B == 0) return RAYEN_OK;
  if (img->Wb != nullptr && !old_mode) {
    if (img->nkk == 1) return launch_split<1>(p, img, v, B, ldv, y, ldy, kappa, active, nan_flag, stream);
    if (img->nkk == 2) return launch_split<2>(p, img, v, B, ldv, y, ldy, kappa, active, nan_flag, stream);
  }
  switch (img->nkk) {
    case 1: return launch_mfma<1, 0>(p, img, v, B, ldv, y, ldy, kappa, active, nan_flag, old_mode, MapperArgs(), stream);
    case 2: return launch_mfma<2, 0>(p, img, v, B, ldv, y, ldy, kappa, active, nan_flag, old_mode, MapperArgs(), stream);
    case 3: return launch_mfma<3, 0>(p, img, v, B, ldv, y, ldy, kappa, active, nan_flag, old_mode, MapperArgs(), stream);
    case 4: return launch_mfma<4, 0>(p, img, v, B, ldv, y, ldy, kappa, active, nan_flag, old_mode, MapperArgs(), stream);
    default: return RAYEN_E_UNSUPPORTED;
  }
}

}  // namespace rayen
