// Paired-half forward with the IMAGE OF W RESIDENT IN LDS (round 6; rayen/constraint_module.py:468-474, 351-458 in one
// launch; the arithmetic is rayen_mfma_pair.hip's, bit for bit).
//
// Why: in rayen_mfma_pair.hip / rayen_mfma_pair_io.hip every wave pulls the whole image through the CU's vector-memory
// path once per 64 samples -- eight 1-KiB global_load_dwordx4 per tile and wave, sixteen cycles of that path each
// (scripts/ubench/mfma_coissue.hip: 64 B per clock and CU), ten with the two row operations: eight waves keep it busy
// 1 280 of the ~2 400 cycles a tile takes, and a wave that cannot hand over its re-load cannot issue its next MFMA either
// (HISTORY.md, "the same ablations on config 3's walk": no A re-loads -18 %, neither re-loads nor row operations -29 %).
// Here the image is copied into LDS ONCE per workgroup (one workgroup per CU; config 3: 14 tiles = 112 KiB) and an A
// operand is a ds_read_b128: 4 LDS cycles instead of 16 vector-memory cycles, on a path nothing else uses.  What that
// buys besides:
//   * no row buffers in LDS (there is no room for them and no need): a lane reads ITS sample's row straight from memory as
//     eight 16-byte pieces (fragment shape: lanes (col, hi) of a row cover 32 contiguous bytes, four consecutive
//     instructions complete a 128-byte line) at the top of its group -- the wait is left to the other waves of the SIMD --
//     through a buffer descriptor whose extent is the batch (one add per group; rows beyond the batch and pieces beyond
//     a ragged width are out of range: loads return 0, stores are dropped), and writes its row of y through 2 KiB of LDS
//     as whole lines;
//   * vmcnt counts nothing but those rows: no counted waits, no stand-in operations;
//   * groups of 32 samples (one sample tile per wave): ~120 instead of 160+ live registers, so FOUR waves per SIMD
//     (sixteen per workgroup, dealt groups on demand from a counter in LDS) take turns on the matrix pipe -- by the same
//     microbenchmark a partner's plain VALU / SALU / LDS instructions issue while a wave's MFMAs execute; measured
//     monotone in the wave count (16: 46.4 us, 12: 49.5, 8: 54.7) -- and the first rows a wave waits for are 8 KiB, not 16;
//   * the image itself arrives by LDS-DMA (global_load_lds_dwordx4: nothing through the registers);
//   * every batch size (one workgroup per CU as soon as there is a group for it), n = k below the padded width, the
//     module's mapper in front (NKX > 0: its image next to W's).
// The item list, the image, the epilogues and the order of the products inside a tile are the other schedules': same bits.
//
// Served: NA_E = I, n = k in (32 (NKK - 1), 32 NKK] and a multiple of 4, rows 16-byte aligned, the image + the aux patches within 160 KiB of LDS, at most
// sixteen aux rows at NKK = 2 (eight behind a mapper).  Everything else stays on the other schedules.
#include "rayen_split_image.h"

#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <type_traits>

namespace rayen {

namespace {
#ifndef RAYEN_WL_WAVES
#define RAYEN_WL_WAVES 16
#endif
#ifndef RAYEN_WL_NT
#define RAYEN_WL_NT 1
#endif
// 1: a group's rows are requested one group ahead (64 registers at two sample tiles) | 0: at the top of the group, the wait
// left to the other waves of the SIMD
#ifndef RAYEN_WL_AHEAD
#define RAYEN_WL_AHEAD 0
#endif
constexpr int kWlWaves = RAYEN_WL_WAVES;      // waves per workgroup = per CU
constexpr int kWlNT = RAYEN_WL_NT;            // sample tiles (of 32) per wave and group
// developer ablation builds (scripts/ubench/tu_variant.sh rayen_mfma_pair_wl <name> -DRAYEN_WL_ABL=<bits>; WRONG RESULTS):
// 1 rows requested once per wave | 2 no rows of y stored | 8 plain stores | 16 no epilogues | 32 no MFMAs
#ifndef RAYEN_WL_ABL
#define RAYEN_WL_ABL 0
#endif
// bit 0: the image is copied by LDS-DMA (0: through registers) | bit 1: a wave's first rows are pulled towards L2 meanwhile
// (config 3, B = 262 144, gpurun_out/r06zza: 0 -> 46.1-46.6 us, 1 -> 45.3-45.9, 2 -> 47.4-49.4: the extra requests cost more than the
// round trip they hide)
#ifndef RAYEN_WL_HEAD
#define RAYEN_WL_HEAD 1
#endif
// aux rows a wave keeps per sample tile: 32 at NKK = 1; at NKK = 2 sixteen (2 KiB: what the staging of y needs anyway), eight in the
// mapped instances (1 KiB: the mapper's image takes the room)
#ifndef RAYEN_WL_AUXR2
#define RAYEN_WL_AUXR2 16
#endif
template <int NKK, bool MAPPED = false> struct WlGeom { static constexpr int AUXR = NKK == 2 ? (MAPPED ? 8 : RAYEN_WL_AUXR2) : 32; };
// a wave's own LDS: the aux patch during the walk ([sample tile][aux row][sample]), then the 4 KiB through which its rows of
// y leave as whole 128-byte lines (32 rows x one line)
// (one sample tile per wave: 16 rows at a time through 2 KiB, so that three or four waves per SIMD fit next to the image)
#ifndef RAYEN_WL_SR
#define RAYEN_WL_SR 0
#endif
constexpr int wl_stage_rows(int nt) { return RAYEN_WL_SR ? RAYEN_WL_SR : (nt == 1 ? 16 : 32); }
constexpr int wl_region_bytes(int nkk, int nt) {
  return nt * (nkk == 2 ? WlGeom<2>::AUXR : WlGeom<1>::AUXR) * 128 > wl_stage_rows(nt) * 128 ? nt * (nkk == 2 ? WlGeom<2>::AUXR : WlGeom<1>::AUXR) * 128
                                                                                             : wl_stage_rows(nt) * 128;
}
// one LDS-DMA: every lane fetches 16 bytes from gbase + voff; lane L lands at LDS byte lds + 16 L.  Nothing returns through the
// vector registers -- by scripts/ubench/mfma_coissue.hip such loads run beside a partner wave's MFMAs, register loads do not.
// (M0 = the LDS base; written in the statement that reads it, restored behind it)
__device__ __forceinline__ void wl_dma16(const char* gbase, const unsigned voff, const unsigned lds) {
  unsigned keep;
  uint64_t asm_base;
  asm volatile(RAYEN_ASM_BASE_COPY "s_mov_b32 %[k], m0\n\ts_mov_b32 m0, %[lds]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[off], " RAYEN_ASM_BASE "\n\ts_mov_b32 m0, %[k]"
               : [k] "=&s"(keep), [b] "=&s"(asm_base)
               : [off] "v"(voff), [base] "s"(gbase), [lds] "s"(lds)
               : "memory");
}
// the mapped instances carry the mapper's image (up to 16 KiB) next to W's: eight rows at a time through 1 KiB
constexpr int kWlStageRowsMapped = 8;
constexpr int wl_region_bytes_mapped(int nkk, int nt) {
  return nt * (nkk == 2 ? WlGeom<2, true>::AUXR : WlGeom<1, true>::AUXR) * 128 > kWlStageRowsMapped * 128
             ? nt * (nkk == 2 ? WlGeom<2, true>::AUXR : WlGeom<1, true>::AUXR) * 128
             : kWlStageRowsMapped * 128;
}
}  // namespace

// developer build (scripts/ubench/tu_variant.sh rayen_mfma_pair_wl clock -DRAYEN_WL_CLOCK; scripts/ubench/wl_clock.py): s_memtime
// (shader clocks) and s_memrealtime (100 MHz) at entry and exit of wave 0 of every 16th workgroup -- the shader clock the
// kernel actually ran at (the chip lowers it under matrix load).  Nothing in the library build.
// developer build (-DRAYEN_WL_STAMPS; scripts/ubench/wl_stamps.py): s_memtime per item of the SECOND group of waves 0 and 4 of
// workgroup 0 -- [item top | burst and the next item's reads issued | epilogue done].  Nothing in the library build.
#ifdef RAYEN_WL_STAMPS
#ifndef RAYEN_WL_STAMP_ROUND
#define RAYEN_WL_STAMP_ROUND 1
#endif
__device__ unsigned long long wl_stamp_buf[2 * 40 * 4];
extern "C" int rayen_debug_wl_stamps(void* dst, size_t bytes) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(wl_stamp_buf), bytes < sizeof(wl_stamp_buf) ? bytes : sizeof(wl_stamp_buf)) == hipSuccess ? 0 : -1;
}
#define RAYEN_WL_STAMP(item, slot)                                                                              \
  do {                                                                                                          \
    if (stamp_on && lane == 0 && (item) < 40) wl_stamp_buf[((wave >> 2) * 40 + (item)) * 4 + (slot)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define RAYEN_WL_STAMP(item, slot) do { } while (0)
#endif
#ifdef RAYEN_WL_CLOCK
__device__ unsigned long long wl_clock_buf[16 * 4];
extern "C" int rayen_debug_wl_clock(void* dst, size_t bytes) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(wl_clock_buf), bytes < sizeof(wl_clock_buf) ? bytes : sizeof(wl_clock_buf)) == hipSuccess ? 0 : -1;
}
#endif

// NKX > 0: the module's mapper v = Wm x + b in front of the walk (rayen/constraint_module.py:259-263, 525; the arithmetic of
// rayen_mfma_pair.hip's mapped instances, bit for bit): `v` / `ldv` / `n_in` are then x, its leading dimension and its width, the
// mapper's f16-pair image (NKK x 2 NKX K-steps x 2 pieces of 1 KiB, then bias, gM, 1 / gM: pair_mapper_image_kernel) is copied into
// LDS behind the image of W, and v reaches memory only when v_out != null (training).
template <int NKK, bool TRACK, int NT, int NW, int NKX = 0>
__global__ __launch_bounds__(NW * 64, 1) void mfma_pair_wl_kernel(
    const f16x8* __restrict__ Wh, const MItem* __restrict__ items, int n_items,
    const MPack* __restrict__ packs, const float* __restrict__ y0, int n_tiles, int n,
    const float* __restrict__ v, int64_t B, int64_t ldv, float* __restrict__ y, int64_t ldy,
    float* __restrict__ kappa_out, int32_t* __restrict__ active_out,
    int32_t* __restrict__ nan_flag, const float w_scale, const float w_inv,
    const f16x8* __restrict__ mimg, int n_in, float* __restrict__ v_out, int64_t ldvo) {
  constexpr int NS = NKK * 2, NCH = NS * 2, NQ = NKK * 4, AUXR = WlGeom<NKK, (NKX > 0)>::AUXR;
  constexpr int NSX = NKX * 2, NQX = NKX * 4, MCH = NKK * NSX * 2;   // the mapper's K-steps, row pieces, 1-KiB chunks
  extern __shared__ __attribute__((aligned(1024))) char wl_smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int col = lane & 31;
  const int hi = lane >> 5;
  char* const wimg = wl_smem;                                                   // [n_tiles][NS][2][64] x 16 bytes
  constexpr int REGION = NKX > 0 ? wl_region_bytes_mapped(NKK, NT) : wl_region_bytes(NKK, NT);
  constexpr int SR = NKX > 0 ? kWlStageRowsMapped : wl_stage_rows(NT);
  char* const mlds = wl_smem + (size_t)n_tiles * (NCH * 1024);                  // (mapped) [NKK][NSX][2][64] x 16 bytes
  char* const stage = mlds + MCH * 1024 + wave * REGION;
  float (*const aux_lds)[AUXR][32] = reinterpret_cast<float (*)[AUXR][32]>(stage);   // [sample tile][row][sample]
  float* const y0_lds = reinterpret_cast<float*>(mlds + MCH * 1024 + NW * REGION);
  float* const bias_lds = y0_lds + NKK * 32;                                    // (mapped) gM b, zero-padded
  unsigned* const take_lds = reinterpret_cast<unsigned*>(bias_lds + (NKX > 0 ? NKK * 32 : 0));   // the workgroup's next unclaimed group
  // Groups are dealt to WORKGROUPS statically (group b + j gridDim.x is the j-th of workgroup b) and to the waves of a
  // workgroup on demand, j from a counter in LDS: the issue arbiter prefers the older wave of a SIMD, so with equal shares
  // the first waves of a workgroup finish early and their partners walk alone at the end (s_memtime / s_memrealtime probes,
  // B = 1 048 576: wave 0 lives 156 us of a 191 us launch)
  const int64_t n_groups = (B + NT * 32 - 1) / (NT * 32);
  const int64_t grp_stride = (int64_t)gridDim.x;
  int64_t grp = (int64_t)blockIdx.x + (int64_t)wave * grp_stride;
  bool bad = false;
  bool first_walk = true;
#ifdef RAYEN_WL_STAMPS
  int stamp_round = 0;
  bool stamp_on = false;
#endif
#ifdef RAYEN_WL_CLOCK
  const bool probe = (blockIdx.x & 15) == 0 && threadIdx.x == 0;
  if (probe) {
    wl_clock_buf[(blockIdx.x >> 4) * 4 + 0] = __builtin_amdgcn_s_memtime();
    wl_clock_buf[(blockIdx.x >> 4) * 4 + 1] = __builtin_amdgcn_s_memrealtime();
  }
#endif

  // ---- the image, once: chunk c (1 KiB) by wave c mod NW, by LDS-DMA (RAYEN_WL_HEAD; nothing passes through the registers, nothing
  // is waited for before the barrier).  See below the request lambda for the rest of the head.
  for (int i = threadIdx.x; i < NKK * 32; i += NW * 64) y0_lds[i] = y0[i];
  if (threadIdx.x == 0) *take_lds = NW;     // (the first NW are the waves' first groups)
  float gm_inv = 1.f;
  int gm_exp = 0;
  if constexpr (NKX > 0) {
    const float* tail = reinterpret_cast<const float*>(mimg + (size_t)MCH * 64);
    const float gm = tail[NKK * 32];
    gm_inv = tail[NKK * 32 + 1];
    gm_exp = (int)((__builtin_bit_cast(unsigned, gm) >> 23) & 255u) - 127;
    for (int i = threadIdx.x; i < NKK * 32; i += NW * 64) bias_lds[i] = tail[i] * gm;
  }
#if !(RAYEN_WL_HEAD & 1)
  {
    const int n_chunks = n_tiles * NCH;
    const char* src = reinterpret_cast<const char*>(Wh) + lane * 16;
#pragma unroll 4
    for (int c = wave; c < n_chunks; c += NW)
      *reinterpret_cast<u32x4*>(wimg + c * 1024 + lane * 16) = *reinterpret_cast<const u32x4*>(src + (size_t)c * 1024);
  }
#endif

  // this lane's rows of group g (one per sample tile): pieces 2 q + hi (columns 8 q + 4 hi .. + 3), zero beyond the batch.
  // Buffer addressing (round 6, from the ISA: the 64-bit `row * ldv` products, the predicates and the zeroing of the flat form
  // were ~70 vector instructions per group, the multiplies at a quarter rate; the stores of y ~80): the descriptor's extent is
  // B rows, the lane's byte offset inside a group is a constant, a group adds one uniform term -- and rows beyond the batch are
  // out of range: their loads return 0, their stores are dropped by the hardware.  (Offsets are 32-bit: mfma_pair_wl_serves.)
  const __amdgpu_buffer_rsrc_t v_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(v), 0, (int)(unsigned)((uint64_t)B * (uint64_t)ldv * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t y_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(y, 0, (int)(unsigned)((uint64_t)B * (uint64_t)ldy * 4u), 0x00020000);
  const unsigned v_lane_off = (unsigned)col * (unsigned)ldv * 4u + 16u * (unsigned)hi;
  const unsigned y_lane_off = (unsigned)(lane >> 3) * (unsigned)ldy * 4u + 16u * (unsigned)((lane & 7) ^ ((lane >> 3) & 7));
  f32x4 vraw[NT][NQ];
  f32x4 xraw[NT][NKX > 0 ? NQX : 1];     // (mapped) the rows of x; vraw then holds gM sx v out of the mapper's accumulators
  // n = k below the padded width (a multiple of 4: whole pieces): the pieces beyond a row's n columns would be the next row's --
  // their offset is sent out of range instead (loads return 0, stores are dropped), one select per piece
  const bool ragged_n = n < NKK * 32;
  const bool ragged_in = NKX > 0 && n_in < NKX * 32;
  constexpr unsigned kNowhere = 0xFFFFFFF0u;
  auto request = [&](const int64_t g) {
    const unsigned goff = v_lane_off + (unsigned)g * (unsigned)(NT * 32) * (unsigned)ldv * 4u;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if constexpr (NKX > 0) {
        if (__builtin_expect(ragged_in, 0)) {
          asm volatile("; ragged width" ::: "memory");
#pragma unroll
          for (int q = 0; q < NQX; ++q) {
            unsigned off = goff + (unsigned)t * 32u * (unsigned)ldv * 4u + 32u * q;
            off = (8 * q + 4 * hi < n_in) ? off : kNowhere;
            xraw[t][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(v_rsrc, off, 0, 0));
          }
        } else {
#pragma unroll
          for (int q = 0; q < NQX; ++q)
            xraw[t][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                v_rsrc, goff + (unsigned)t * 32u * (unsigned)ldv * 4u + 32u * q, 0, 0));
        }
      } else {
        // (two copies behind a wave-uniform branch, the empty statement keeps hipcc from folding them back into one with a select
        // per piece: at the padded width -- the headline -- the selects were 3 % of the kernel's vector instructions)
        if (__builtin_expect(ragged_n, 0)) {
          asm volatile("; ragged width" ::: "memory");
#pragma unroll
          for (int q = 0; q < NQ; ++q) {
            unsigned off = goff + (unsigned)t * 32u * (unsigned)ldv * 4u + 32u * q;
            off = (8 * q + 4 * hi < n) ? off : kNowhere;
            vraw[t][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(v_rsrc, off, 0, 0));
          }
        } else {
#pragma unroll
          for (int q = 0; q < NQ; ++q)
            vraw[t][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                v_rsrc, goff + (unsigned)t * 32u * (unsigned)ldv * 4u + 32u * q, 0, 0));
        }
      }
    }
  };
  // largest |component| of the requested rows (this lane's half of each): the FIRST use of the rows' registers.  It sits in
  // front of the previous group's stores of y -- vmcnt retires in order and counts stores, so a wait for these loads that
  // came behind the stores would wait for the stores' HBM round trip as well
  float m_half[NT];
  auto half_max = [&]() {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float m = 0.f;
      if constexpr (NKX > 0) {
#pragma unroll
        for (int q = 0; q < NQX; ++q)
#pragma unroll
          for (int c = 0; c < 4; ++c) m = fmaxf(m, __builtin_fabsf(xraw[t][q][c]));
      } else {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
          for (int c = 0; c < 4; ++c) m = fmaxf(m, __builtin_fabsf(vraw[t][q][c]));
      }
      m_half[t] = m;
      asm volatile("" : "+v"(m_half[t]));   // (pinned here: hipcc otherwise sinks the maxima -- and their wait -- behind the stores)
    }
  };
#pragma unroll
  for (int t = 0; t < NT; ++t) m_half[t] = 0.f;
  // One 128-byte line (h) of each of a sample tile's 32 rows out of the registers that hold pieces 2 qq + hi of the lane's OWN row:
  // through the wave's LDS (slot = piece ^ (row & 7): the eight lanes of a write or read group hit eight different 16-byte bank
  // groups), SR rows at a time, and out as whole lines -- lane L stores slot L & 7 of row 8 i + (L >> 3).  `goff` = the lane's
  // constant offset + the group's rows, `ld4` = the leading dimension in bytes; pieces beyond `width` columns and rows beyond
  // the batch are out of the descriptor's range.
  auto put_lines = [&](const f32x4 (&o)[4], const int h, const int t, const __amdgpu_buffer_rsrc_t rsrc, const unsigned goff,
                       const unsigned ld4, const int width, const bool ragged) {
#pragma unroll
    for (int part = 0; part < 32 / SR; ++part) {      // rows [SR part, SR part + SR) of the tile
      if (SR == 32 || (col / SR) == part) {
        char* slot = stage + (col % SR) * 128;
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) *reinterpret_cast<f32x4*>(slot + (((2 * qq + hi) ^ (col & 7)) * 16)) = o[qq];
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < SR / 8; ++i) {
        const int r = 8 * i + (lane >> 3);
        const f32x4 x = *reinterpret_cast<const f32x4*>(stage + r * 128 + (lane & 7) * 16);
        // row (lane >> 3) of the eight, slot (lane & 7) ^ (row & 7): the lane's constant; rows beyond the batch are dropped
        unsigned off = goff + (unsigned)(t * 32 + part * SR + 8 * i) * ld4 + 128u * h;
        if (__builtin_expect(ragged, 0)) {
          asm volatile("; ragged width" ::: "memory");     // (a branch, not a select per store: see request)
          off = (32 * h + 4 * ((lane & 7) ^ ((lane >> 3) & 7)) < width) ? off : kNowhere;
        }
        if constexpr (RAYEN_WL_ABL & 2) { if (x[0] == 123.456f) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, x), rsrc, off, 0, 0); }
        else if constexpr (RAYEN_WL_ABL & 8) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, x), rsrc, off, 0, 0);
        else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, x), rsrc, off, 0, 2);   // (2: non-temporal)
      }
      __builtin_amdgcn_wave_barrier();
    }
  };
#if RAYEN_WL_HEAD
  {
    // the image's chunks, then -- behind them, loads retire in order -- the wave's first rows, pulled towards L2 into the wave's own
    // (still unused) kilobyte and thrown away: the group's real request at the top of the walk finds the lines in L2 instead of
    // paying an HBM round trip behind the barrier.  Whole groups only (this path has no bounds).  Only the image is waited for.
    const int n_chunks = n_tiles * NCH;
    const unsigned img_at = __builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<uintptr_t>(wimg));
    const unsigned lane16 = lane * 16;
    if (RAYEN_WL_HEAD & 1)
      for (int c = wave; c < n_chunks; c += NW)
        wl_dma16(reinterpret_cast<const char*>(Wh) + (size_t)c * 1024, lane16, img_at + c * 1024);
    if constexpr (NKX > 0)
      for (int c = wave; c < MCH; c += NW)
        wl_dma16(reinterpret_cast<const char*>(mimg) + (size_t)c * 1024, lane16, img_at + (n_chunks + c) * 1024);
    if ((RAYEN_WL_HEAD & 2) && (grp + 1) * (NT * 32) <= B) {
      const char* gb = reinterpret_cast<const char*>(v + grp * (NT * 32) * ldv);
      const unsigned lds_at = __builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<uintptr_t>(stage));
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < NQ; ++q) wl_dma16(gb, v_lane_off + (unsigned)t * 32u * (unsigned)ldv * 4u + 32u * q, lds_at);
      asm volatile("s_waitcnt vmcnt(%0)" : : "n"(NT * NQ) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
#endif
  if (RAYEN_WL_AHEAD && grp < n_groups) {
    request(grp);
    half_max();
  }
  __syncthreads();  // the only workgroup barrier: the image is in LDS

  while (grp < n_groups) {
    const int64_t s_base = grp * (NT * 32);
    int taken = 0;
    if (lane == 0) taken = (int)atomicAdd(take_lds, 1u);
    const int64_t next = (int64_t)blockIdx.x + (int64_t)__builtin_amdgcn_readfirstlane(taken) * grp_stride;
    bool live[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) live[t] = (s_base + t * 32 + col) < B;
    if constexpr (!RAYEN_WL_AHEAD) {
      request(grp);
      half_max();
    }

#ifdef RAYEN_WL_STAMPS
    stamp_on = blockIdx.x == 0 && (wave == 0 || wave == 4) && stamp_round == RAYEN_WL_STAMP_ROUND;
    ++stamp_round;
    RAYEN_WL_STAMP(38, 0);
#endif
    // ---- rows -> scaled f16 pairs.  sv = 2^(13 - floor(log2 max|v|)): exponent arithmetic only (rayen_mfma_pair.hip);
    // vb[t][piece][k-step] = 8 f16 = the B operand of one MFMA; element i = column 16 sp + 8 (i >> 2) + 4 hi + (i & 3)
    f16x8 vb[NT][2][NS];
    float v_scl[NT], v_inv[NT];
    bool nan_row[NT];
    int sx_exp[NT];
    if constexpr (NKX > 0) {
      // ---- the mapper: x scaled per sample (sx, a power of two) and split like v; the accumulators start at gM sx b; per K-step
      // of x and row tile of v the three products in the order of rayen_mfma_pair.hip (a2 x1, a1 x2, a1 x1); A operands from LDS.
      // The fp32 results ARE in B-operand order: register 4 a + c of row tile tp is direction element 32 tp + 8 a + 4 hi + c.
      static_assert(NT == 1, "the mapped instances walk one sample tile per wave");
      static_assert((RAYEN_WL_HEAD & 1) != 0, "the mapper's image is copied by the LDS-DMA head");
      const unsigned vo_goff = (unsigned)(lane >> 3) * (unsigned)ldvo * 4u + 16u * (unsigned)((lane & 7) ^ ((lane >> 3) & 7)) +
                               (unsigned)s_base * (unsigned)ldvo * 4u;
      const __amdgpu_buffer_rsrc_t vo_rsrc =
          __builtin_amdgcn_make_buffer_rsrc(v_out, 0, (int)(unsigned)((uint64_t)B * (uint64_t)ldvo * 4u), 0x00020000);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float mx = fmaxf(m_half[t], xhalf(m_half[t]));
        float sx, sx_inv;
        pow2_scale(mx, sx, sx_inv, sx_exp[t]);
        f32x16 macc[NKK];
#pragma unroll
        for (int tp = 0; tp < NKK; ++tp)
#pragma unroll
          for (int a4 = 0; a4 < 4; ++a4) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(&bias_lds[32 * tp + 8 * a4 + 4 * hi]);
#pragma unroll
            for (int c = 0; c < 4; ++c) macc[tp][4 * a4 + c] = b4[c] * sx;
          }
#pragma unroll
        for (int sxs = 0; sxs < NSX; ++sxs) {
          u32x4 w1, w2;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int q = 2 * sxs + (j >> 1), c = 2 * (j & 1);
            unsigned a, b;
            pair_split_lo(a, b, xraw[t][q][c], sx);
            pair_split_hi(a, b, xraw[t][q][c + 1], sx);
            w1[j] = a;
            w2[j] = b;
          }
          const f16x8 xb0 = __builtin_bit_cast(f16x8, w1), xb1 = __builtin_bit_cast(f16x8, w2);
#pragma unroll
          for (int tp = 0; tp < NKK; ++tp) {
            const char* ch = mlds + (size_t)((tp * NSX + sxs) * 2) * 1024 + lane * 16;
            const f16x8 a1 = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(ch));
            const f16x8 a2 = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(ch + 1024));
            macc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, xb0, macc[tp], 0, 0, 0);
            macc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, xb1, macc[tp], 0, 0, 0);
            macc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, xb0, macc[tp], 0, 0, 0);
          }
        }
        // macc = gM sx v.  v itself leaves only for the backward (training): the same lines as y
        if (v_out != nullptr) {
#pragma unroll
          for (int h = 0; h < NKK; ++h) {
            f32x4 o[4];
#pragma unroll
            for (int qq = 0; qq < 4; ++qq)
#pragma unroll
              for (int c = 0; c < 4; ++c) o[qq][c] = fmaf(macc[h][4 * qq + c] * gm_inv, sx_inv, 0.f);
            put_lines(o, h, t, vo_rsrc, vo_goff, (unsigned)ldvo * 4u, n, ragged_n);
          }
        }
        float m = 0.f;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            vraw[t][q][c] = macc[q >> 2][4 * (q & 3) + c];
            m = fmaxf(m, __builtin_fabsf(vraw[t][q][c]));
          }
        m_half[t] = m;
      }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float m = fmaxf(m_half[t], xhalf(m_half[t]));
      float sv;
      int sv_exp;
      pow2_scale(m, sv, v_inv[t], sv_exp);
      v_scl[t] = sv;
      if constexpr (NKX > 0) {
        // the pieces are split from f x (gM sx v), f = the power of two above; sv = f gM sx as a power of two (the exponents are
        // added: the product of the floats could overflow on the way)
        int e = sv_exp + gm_exp + sx_exp[t];
        e = e > 126 ? 126 : (e < -126 ? -126 : e);
        v_scl[t] = __builtin_bit_cast(float, (unsigned)(127 + e) << 23);
        v_inv[t] = __builtin_bit_cast(float, (unsigned)(127 - e) << 23);
      }
      f16x2 z = {(_Float16)0.f, (_Float16)0.f};
#pragma unroll
      for (int sp = 0; sp < NS; ++sp) {
        u32x4 w1, w2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {           // register j of the K-step: columns q = 2 sp + (j >> 1), c = 2 (j & 1) + {0, 1}
          const int q = 2 * sp + (j >> 1), c = 2 * (j & 1);
          unsigned a, b;
          pair_split_lo(a, b, vraw[t][q][c], sv);
          pair_split_hi(a, b, vraw[t][q][c + 1], sv);
          w1[j] = a;
          w2[j] = b;
          pair_nan_fold(z, a);
        }
        vb[t][0][sp] = __builtin_bit_cast(f16x8, w1);
        vb[t][1][sp] = __builtin_bit_cast(f16x8, w2);
      }
      nan_row[t] = pair_nan_seen(z);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- the next group's rows: in flight for the whole walk
    if (RAYEN_WL_AHEAD && next < n_groups && !(RAYEN_WL_ABL & 1)) request(next);
    __builtin_amdgcn_sched_barrier(0);

    // kap, part, the aux patch and the accumulators live in the SCALED domain (gW sv times the natural value)
    float kap[NT], part[NT];
    int acode[NT];   // arg-max bookkeeping in one register: (segment << 20) | row, -1 = none
#pragma unroll
    for (int t = 0; t < NT; ++t) { kap[t] = 0.f; part[t] = 0.f; acode[t] = -1; }
    {
      f32x16 acc[NT];
      const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      // The A operands of an item are read from LDS one item AHEAD, right behind the previous item's MFMAs: their latency
      // runs under that item's epilogue (the chunk registers are dead once its MFMAs have read them).  The last item of a
      // group fetches the first item's chunks for the next group.
      const int ts_first = items[0].tile_shape;
      int ts_next = ts_first;
      u32x4 abuf[NCH];   // [2 sp + 0] leading piece, [2 sp + 1] second piece of K-step sp
      auto fetch = [&](const int ts_of) {
        const char* tb = wimg + (size_t)(ts_of & 0xFFFFFF) * (NCH * 1024) + lane * 16;
        if constexpr (NS == 4) {
          const int shape = __builtin_amdgcn_readfirstlane((ts_of >> 24) & 3);
          if (shape != MS_HALF_B) {
#pragma unroll
            for (int c = 0; c < 4; ++c) abuf[c] = *reinterpret_cast<const u32x4*>(tb + c * 1024);
          }
          if (shape != MS_HALF_A) {
#pragma unroll
            for (int c = 4; c < 8; ++c) abuf[c] = *reinterpret_cast<const u32x4*>(tb + c * 1024);
          }
        } else {
#pragma unroll
          for (int c = 0; c < NCH; ++c) abuf[c] = *reinterpret_cast<const u32x4*>(tb + c * 1024);
        }
      };
      if (first_walk) {
        fetch(ts_first);
        first_walk = false;
      }
      for (int it = 0; it < n_items; ++it) {
        RAYEN_WL_STAMP(it, 0);
        const int ts = ts_next;      // (known since the previous item: nothing in front of the burst waits for a scalar load)
        // Two passes over the item's K-steps, by product size: the 2^-11 cross products first, the leading products last
        // (rayen_mfma_pair.hip).  A full tile multiplies all K-steps; the halves of a shared tile (rayen_tiles.h) K-steps
        // 0,1 or 2,3, the accumulator starting afresh at the first of them.
        auto burst = [&](auto S0, auto S1) {
          constexpr int s0 = decltype(S0)::value, s1 = decltype(S1)::value;
#pragma unroll
          for (int sp = s0; sp < s1; ++sp) {
            const f16x8 a1 = __builtin_bit_cast(f16x8, abuf[2 * sp + 0]), a2 = __builtin_bit_cast(f16x8, abuf[2 * sp + 1]);
#pragma unroll
            for (int t = 0; t < NT; ++t)
              acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, vb[t][0][sp], sp == s0 ? zero : acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, vb[t][1][sp], acc[t], 0, 0, 0);
          }
#pragma unroll
          for (int sp = s0; sp < s1; ++sp) {
            const f16x8 a1 = __builtin_bit_cast(f16x8, abuf[2 * sp + 0]);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, vb[t][0][sp], acc[t], 0, 0, 0);
          }
        };
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (RAYEN_WL_ABL & 32) {
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[t][g] = __builtin_bit_cast(float, __builtin_bit_cast(u32x4, vb[t][0][g & 3])[g >> 2]);
        } else if constexpr (NS == 4) {
          const int shape = __builtin_amdgcn_readfirstlane((ts >> 24) & 3);
          if (shape == MS_FULL) burst(std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{});
          else if (shape == MS_HALF_A) burst(std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
          else burst(std::integral_constant<int, 2>{}, std::integral_constant<int, 4>{});
        } else {
          burst(std::integral_constant<int, 0>{}, std::integral_constant<int, NS>{});
        }
        __builtin_amdgcn_sched_barrier(0);
        // (the item's record is requested BEHIND its MFMAs: LDS and scalar loads share a counter that hipcc can only wait
        // out in full, so a request in front of the burst would hold the MFMAs back for the scalar cache's latency)
        const MItem item = items[it];
        ts_next = (it + 1 < n_items) ? item.qbegin : ts_first;      // (qbegin: the NEXT item's tile and shape, mfma_pair_build)
        if constexpr (!(RAYEN_WL_ABL & 32)) fetch(ts_next);
        RAYEN_WL_STAMP(it, 1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (RAYEN_WL_ABL & 16) {
#pragma unroll
          for (int t = 0; t < NT; ++t) kap[t] = fmaxf(kap[t], acc[t][0]);
          continue;
        }

        if (item.type == MI_LIN) {
          const int lin_code = (item.seg << 20) + item.row0 + 4 * hi;
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            if (TRACK) {
#pragma unroll
              for (int g = 0; g < 16; ++g)
                if (acc[t][g] > kap[t]) {
                  kap[t] = acc[t][g];
                  acode[t] = lin_code + ((g & 3) + 8 * (g >> 2));
                }
            } else {
#pragma unroll
              for (int g = 0; g < 16; ++g) kap[t] = fmaxf(kap[t], acc[t][g]);
            }
          }
        } else if (item.type == MI_QFAC || item.type == MI_SOC) {
          // a running sum of squares over the segment's tiles, closed on its last tile (two chains of plain FMAs: the values
          // of the other schedules' packed FMA, in instructions that issue under a partner's MFMAs)
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            float s0 = (item.flags & MF_FIRST) ? 0.f : part[t], s1 = 0.f;
#pragma unroll
            for (int g = 0; g < 16; g += 2) {
              s0 = fmaf(acc[t][g], acc[t][g], s0);
              s1 = fmaf(acc[t][g + 1], acc[t][g + 1], s1);
            }
            part[t] = s0 + s1;
          }
          if (item.flags & MF_LAST) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
              const float total = part[t] + xhalf(part[t]);
              const float a0 = aux_lds[t][item.aux][col];
              float kc;
              if (item.type != MI_SOC) {
                kc = (a0 + __builtin_amdgcn_sqrtf(fmaxf(total, 0.f))) * item.seg_inv;   // (the segment's own power of two undone)
              } else {
                kc = pair_soc_candidate(a0, aux_lds[t][item.aux + 1][col], total, w_inv * item.seg_inv, v_inv[t], item.f0,
                                        item.f1, v_scl[t], w_scale);
              }
              if (kc > kap[t]) { kap[t] = kc; acode[t] = item.seg << 20; }
            }
          }
        } else if (item.type == MI_AUX) {
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int g = 0; g < AUXR / 2; ++g) aux_lds[t][(g & 3) + 8 * (g >> 2) + 4 * hi][col] = acc[t][g];   // (rows 0 .. AUXR - 1 of the tile)
          __builtin_amdgcn_wave_barrier();
        } else if (item.type == MI_PACK) {
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
              const float kc = (aux_lds[t][slot & (AUXR - 1)][col] + __builtin_amdgcn_sqrtf(qs)) * (hi ? pk.inv[a][1] : pk.inv[a][0]);
              if (sid >= 0 && kc > kap[t]) { kap[t] = kc; acode[t] = sid << 20; }
            }
          }
        }
        RAYEN_WL_STAMP(it, 2);
      }
    }
    RAYEN_WL_STAMP(38, 1);

    // ---- kappa is final
    float knat[NT], scale[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float other = xhalf(kap[t]);
      if (TRACK) {
        const int ocode = __shfl_xor(acode[t], 32);
        if (other > kap[t] || (other == kap[t] && hi == 1)) acode[t] = ocode;
      }
      kap[t] = fmaxf(kap[t], other);
      knat[t] = (kap[t] * w_inv) * v_inv[t];
      scale[t] = v_inv[t] * (1.0f / fmaxf(1.0f, knat[t]));   // (the rebuilt direction carries sv only)
    }
    if (hi == 0) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (!live[t]) continue;
        const int64_t s = s_base + t * 32 + col;
        if (kappa_out) kappa_out[s] = knat[t];
        if (TRACK) { active_out[2 * s] = acode[t] >> 20; active_out[2 * s + 1] = acode[t] < 0 ? 0 : (acode[t] & 0xFFFFF); }
      }
    }

    // the next rows have had the walk to arrive: first use (see half_max; unconditional -- on every path into the next group
    // the loads have been waited for)
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (RAYEN_WL_AHEAD) half_max();
    __builtin_amdgcn_sched_barrier(0);

    // ---- y = y0 + v / max(1, kappa): v rebuilt from its pieces (22 bits of it; scaled by sv, undone by `scale`).  A lane
    // holds pieces 2 q + hi of ITS row: written to memory as they are, every store instruction would touch 32 lines with 32
    // bytes each (measured: non-temporal 331 us, plain 197 us at B = 1 048 576 against 157 without the stores, and the plain
    // ones leave the L2 dirty for the end of the kernel).  So one 128-byte line of each of the tile's 32 rows at a time goes
    // through the wave's 4 KiB of LDS (slot = piece ^ (row & 7): the eight lanes of a write or read group hit eight different
    // 16-byte bank groups) and leaves as whole lines: lane L stores slot L & 7 of row 8 i + (L >> 3).
    const unsigned y_goff = y_lane_off + (unsigned)s_base * (unsigned)ldy * 4u;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float one = 1.0f;
      asm volatile("" : "+v"(one));
#pragma unroll
      for (int h = 0; h < NKK; ++h) {          // line h of the tile's rows
        f32x4 o[4];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          const int q = 4 * h + qq;
          const f32x4 o4 = *reinterpret_cast<const f32x4*>(&y0_lds[8 * q + 4 * hi]);
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int i = (q & 1) * 4 + c;
            const unsigned w1 = __builtin_bit_cast(u32x4, vb[t][0][q >> 1])[i >> 1], w2 = __builtin_bit_cast(u32x4, vb[t][1][q >> 1])[i >> 1];
            const float val = (i & 1) ? pair_rebuild_hi(w1, w2, one) : pair_rebuild_lo(w1, w2, one);     // fl32(p1 + p2)
            o[qq][c] = fmaf(val, scale[t], o4[c]);
          }
        }
        put_lines(o, h, t, y_rsrc, y_goff, (unsigned)ldy * 4u, n, ragged_n);
      }
      bad |= live[t] && nan_row[t];      // (the row's own components: y is NaN exactly when one of them is NaN or Inf)
    }
    RAYEN_WL_STAMP(38, 2);
    grp = next;
  }  // persistent loop over sample groups
  if (nan_flag && bad) atomicOr(nan_flag, 1);
#ifdef RAYEN_WL_CLOCK
  if (probe) {
    wl_clock_buf[(blockIdx.x >> 4) * 4 + 2] = __builtin_amdgcn_s_memtime();
    wl_clock_buf[(blockIdx.x >> 4) * 4 + 3] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

// ---------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------
static int pair_wl_lds_bytes_mapped(const PairImage* img, int nkx) {
  return img->n_tiles * (img->nkk * 4 * 1024) + img->nkk * (nkx * 2) * 2 * 1024 + kWlWaves * wl_region_bytes_mapped(img->nkk, 1) +
         2 * img->nkk * 128 + 16;
}
static int pair_wl_lds_bytes(const PairImage* img) {
  return img->n_tiles * (img->nkk * 4 * 1024) + kWlWaves * wl_region_bytes(img->nkk, kWlNT) + img->nkk * 128 + 16;
}

bool mfma_pair_wl_serves(const RayenPack* p, const PairImage* img, const float* v, int64_t B, int64_t ldv,
                         const float* y, int64_t ldy) {
  if (img == nullptr || img->nkk < 1 || img->nkk > 2 || img->n_tiles <= 0 || !img->wl_ready) return false;
  // (n = k: the padded width or, round 6, anything down to the next narrower instance in whole 16-byte pieces -- config 2's 16)
  if (!img->identity || p->k != p->n || p->n > img->nkk * 32 || p->n <= (img->nkk - 1) * 32 || (p->n % 4) != 0) return false;
  if (ldv < p->n || ldy < p->n) return false;
  if ((ldv % 4) != 0 || (ldy % 4) != 0) return false;
  // (buffer addressing with 32-bit byte offsets: mfma_pair_wl_forward cuts a batch beyond 4 GiB of rows into several launches)
  if (ldv > (1 << 22) || ldy > (1 << 22)) return false;
  if ((reinterpret_cast<uintptr_t>(v) & 15) != 0 || (reinterpret_cast<uintptr_t>(y) & 15) != 0) return false;
  if (img->nkk == 2 && img->aux_rows > WlGeom<2>::AUXR) return false;
  if (pair_wl_lds_bytes(img) > 160 * 1024) return false;
  // Every batch size (round 6, gpurun_out/r06zzd, r06zze; config 3, us per call): the grid is one workgroup per CU as soon as there is
  // a group for it, so a small batch leaves most waves of a workgroup idle and its busy waves alone on their SIMDs --
  //   B = 32: 10.7 (plain 15.6) | 4 096: 11.3 (18.6) | 32 768: 13.6 (19.3) | 65 536: 15.9 (20.9, W-stationary) | 98 304: 20.5 (24.9) | 131 072: 25.5 (29.7)
  // (the other schedules pull the 112 KiB image through every wave's vector-memory path from L2; here one copy per CU, then LDS).
  // Until the grid covered every CU (it was n_groups / 16 workgroups) this schedule lost below 98 304 rows.
  const int64_t n_groups = (B + kWlNT * 32 - 1) / (kWlNT * 32);
  static const int64_t min_groups_env = [] { const char* e = getenv("RAYEN_WL_MIN_GROUPS"); return e ? atoll(e) : 1ll; }();   // developer sweeps
  return n_groups >= min_groups_env;
}

// called by rayen_pack_create (the only place that may touch function attributes)
int mfma_pair_wl_prepare(const RayenPack* p, PairImage* img) {
  (void)p;
  if (img == nullptr || img->nkk < 1 || img->nkk > 2 || img->n_tiles <= 0 || !img->identity) return RAYEN_OK;
  const int lds = pair_wl_lds_bytes(img);
  if (lds > 160 * 1024) return RAYEN_OK;
  static std::mutex mu;
  static int promised = 0;      // (the attribute belongs to the kernel instance: every pack asks for the running maximum)
  std::lock_guard<std::mutex> hold(mu);
  promised = std::max(promised, lds);
  const int ask = promised;
  bool ok = true;
  auto want = [&](auto kern) {
    ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, ask) == hipSuccess;
  };
  want(mfma_pair_wl_kernel<1, false, kWlNT, kWlWaves>); want(mfma_pair_wl_kernel<1, true, kWlNT, kWlWaves>);
  want(mfma_pair_wl_kernel<2, false, kWlNT, kWlWaves>); want(mfma_pair_wl_kernel<2, true, kWlNT, kWlWaves>);
  if (!ok) return RAYEN_E_LAUNCH;
  img->wl_ready = true;
  // the mapped instances, for the widest mapper this pack can be given (in_dim = the padded width): attributes are set HERE, never
  // on a launch path; a pack whose image leaves no room for it keeps its mapped calls on rayen_mfma_pair.hip
  const int lds_m = pair_wl_lds_bytes_mapped(img, img->nkk);
  if (kWlNT == 1 && lds_m <= 160 * 1024) {
    static int promised_m = 0;
    promised_m = std::max(promised_m, lds_m);
    const int ask_m = promised_m;
    auto want_m = [&](auto kern) {
      ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, ask_m) == hipSuccess;
    };
    want_m(mfma_pair_wl_kernel<1, false, 1, kWlWaves, 1>); want_m(mfma_pair_wl_kernel<1, true, 1, kWlWaves, 1>);
    want_m(mfma_pair_wl_kernel<2, false, 1, kWlWaves, 1>); want_m(mfma_pair_wl_kernel<2, true, 1, kWlWaves, 1>);
    want_m(mfma_pair_wl_kernel<2, false, 1, kWlWaves, 2>); want_m(mfma_pair_wl_kernel<2, true, 1, kWlWaves, 2>);
    if (!ok) return RAYEN_E_LAUNCH;
    img->wl_mapped_ready = true;
  }
  return RAYEN_OK;
}

// ---- the module's mapper in front (rayen_ray_project_mapped_image_f32): x [B, in_dim] -> v = Wm x + b -> y, one launch
bool mfma_pair_wl_serves_mapped(const RayenPack* p, const PairImage* img, const float* x, int64_t B, int64_t ldx, int in_dim,
                                const float* v_out, int64_t ldvo, const float* y, int64_t ldy) {
  if (kWlNT != 1 || img == nullptr || img->nkk < 1 || img->nkk > 2 || img->n_tiles <= 0 || !img->wl_mapped_ready) return false;
  if (!img->identity || p->k != p->n || p->n > img->nkk * 32 || p->n <= (img->nkk - 1) * 32 || (p->n % 4) != 0) return false;
  if (in_dim < 4 || in_dim > img->nkk * 32 || (in_dim % 4) != 0 || ldx < in_dim || ldy < p->n) return false;
  if ((ldx % 4) != 0 || (ldy % 4) != 0 || ldx > (1 << 22) || ldy > (1 << 22)) return false;
  if ((reinterpret_cast<uintptr_t>(x) & 15) != 0 || (reinterpret_cast<uintptr_t>(y) & 15) != 0) return false;
  if (v_out != nullptr && ((ldvo % 4) != 0 || ldvo < p->n || ldvo > (1 << 22) || (reinterpret_cast<uintptr_t>(v_out) & 15) != 0)) return false;
  if (img->nkk == 2 && img->aux_rows > WlGeom<2, true>::AUXR) return false;
  if (pair_wl_lds_bytes_mapped(img, (in_dim + 31) / 32) > 160 * 1024) return false;
  static const int64_t min_groups_env = [] { const char* e = getenv("RAYEN_WL_MIN_GROUPS"); return e ? atoll(e) : 1ll; }();   // developer sweeps
  return B >= 1 && (B + 31) / 32 >= min_groups_env;
}

int mfma_pair_wl_forward_mapped(const RayenPack* p, const PairImage* img, const float* x, int64_t B, int64_t ldx, int in_dim,
                                const void* image, float* v_out, int64_t ldvo, float* y, int64_t ldy, float* kappa,
                                int32_t* active, int32_t* nan_flag, hipStream_t stream) {
  if (B == 0) return RAYEN_OK;
  if (image == nullptr || !mfma_pair_wl_serves_mapped(p, img, x, B, ldx, in_dim, v_out, ldvo, y, ldy)) return RAYEN_E_UNSUPPORTED;
  const int nkx = (in_dim + 31) / 32;
  const int64_t cus = launch_simds(img->n_simd) / 4;
  const int lds = pair_wl_lds_bytes_mapped(img, nkx);
  const int64_t ld_max = std::max<int64_t>(std::max(std::max(ldx, ldy), v_out ? ldvo : 1), 1);
  const int64_t rows_max = (((int64_t)0xFFFFFFFFll / (ld_max * 4)) - 64) / 32 * 32;
  for (int64_t r0 = 0; r0 < B; r0 += rows_max) {
    const int64_t Bc = std::min(B - r0, rows_max);
    const int64_t n_groups = (Bc + 31) / 32;
    const unsigned grid = (unsigned)std::min<int64_t>(cus, n_groups);
    const float* xc = x + r0 * ldx;
    float* yc = y + r0 * ldy;
    float* voc = v_out ? v_out + r0 * ldvo : nullptr;
    float* kc = kappa ? kappa + r0 : nullptr;
    int32_t* ac = active ? active + 2 * r0 : nullptr;
    auto go = [&](auto kern) {
      hipLaunchKernelGGL(kern, dim3(grid), dim3(kWlWaves * 64), lds, stream, static_cast<const f16x8*>(img->Wh), img->items,
                         img->n_items, img->packs, img->y0, img->n_tiles, p->n, xc, Bc, ldx, yc, ldy, kc, ac, nan_flag,
                         img->w_scale, img->w_inv, static_cast<const f16x8*>(image), in_dim, voc, ldvo);
    };
    if (img->nkk == 1) {
      if (active != nullptr) go(mfma_pair_wl_kernel<1, true, 1, kWlWaves, 1>);
      else go(mfma_pair_wl_kernel<1, false, 1, kWlWaves, 1>);
    } else if (nkx == 1) {
      if (active != nullptr) go(mfma_pair_wl_kernel<2, true, 1, kWlWaves, 1>);
      else go(mfma_pair_wl_kernel<2, false, 1, kWlWaves, 1>);
    } else {
      if (active != nullptr) go(mfma_pair_wl_kernel<2, true, 1, kWlWaves, 2>);
      else go(mfma_pair_wl_kernel<2, false, 1, kWlWaves, 2>);
    }
  }
  return hipGetLastError() == hipSuccess ? RAYEN_OK : RAYEN_E_LAUNCH;
}

int mfma_pair_wl_forward(const RayenPack* p, const PairImage* img, const float* v, int64_t B, int64_t ldv,
                         float* y, int64_t ldy, float* kappa, int32_t* active, int32_t* nan_flag,
                         hipStream_t stream) {
  if (B == 0) return RAYEN_OK;
  if (!mfma_pair_wl_serves(p, img, v, B, ldv, y, ldy)) return RAYEN_E_UNSUPPORTED;
  const int64_t cus = launch_simds(img->n_simd) / 4;
  const int lds = pair_wl_lds_bytes(img);
  // The descriptors address 32-bit byte offsets, and the rows of a ragged last group beyond the batch must not wrap: a
  // batch whose rows span more than that goes out as several launches over whole groups (same groups, same bits).
  const int64_t ld_max = std::max<int64_t>(std::max(ldv, ldy), 1);
  const int64_t rows_max = (((int64_t)0xFFFFFFFFll / (ld_max * 4)) - 64) / (kWlNT * 32) * (kWlNT * 32);
  for (int64_t r0 = 0; r0 < B; r0 += rows_max) {
    const int64_t Bc = std::min(B - r0, rows_max);
    const int64_t n_groups = (Bc + kWlNT * 32 - 1) / (kWlNT * 32);
    // one workgroup per CU; every CU takes part as soon as there is a group for it (a workgroup whose waves have no group of their own
    // to start with just leaves them idle: two busy waves on a SIMD run twice as fast as four)
    const unsigned grid = (unsigned)std::min<int64_t>(cus, n_groups);
    const float* vc = v + r0 * ldv;
    float* yc = y + r0 * ldy;
    float* kc = kappa ? kappa + r0 : nullptr;
    int32_t* ac = active ? active + 2 * r0 : nullptr;
    auto go = [&](auto kern) {
      hipLaunchKernelGGL(kern, dim3(grid), dim3(kWlWaves * 64), lds, stream, static_cast<const f16x8*>(img->Wh), img->items,
                         img->n_items, img->packs, img->y0, img->n_tiles, p->n, vc, Bc, ldv, yc, ldy, kc, ac, nan_flag,
                         img->w_scale, img->w_inv, static_cast<const f16x8*>(nullptr), 0, static_cast<float*>(nullptr), (int64_t)0);
    };
    if (img->nkk == 1) {
      if (active != nullptr) go(mfma_pair_wl_kernel<1, true, kWlNT, kWlWaves>);
      else go(mfma_pair_wl_kernel<1, false, kWlNT, kWlWaves>);
    } else {
      if (active != nullptr) go(mfma_pair_wl_kernel<2, true, kWlNT, kWlWaves>);
      else go(mfma_pair_wl_kernel<2, false, kWlNT, kWlWaves>);
    }
  }
  return hipGetLastError() == hipSuccess ? RAYEN_OK : RAYEN_E_LAUNCH;
}

}  // namespace rayen
