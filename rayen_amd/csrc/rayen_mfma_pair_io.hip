// Paired-half forward with the rows of v and y TRICKLED through LDS under the tile walk
// (rayen/constraint_module.py:468-474, 351-458 in one launch; the arithmetic is rayen_mfma_pair.hip's, bit for bit).
//
// What this kernel changes is WHEN the HBM traffic happens.  In rayen_mfma_pair.hip every wave loads its 64 rows, walks
// the tiles, stores its 64 rows -- and since every wave of the chip does so at the same time, the loads and stores are
// chip-wide bursts at the HBM roofline with the matrix pipe idle, and the walks run with HBM idle (DESIGN.md 4.0b:
// 19 k of 52 k cycles per group).  Here a wave owns ONE LDS buffer of 64 rows and, while it walks group g:
//   * chunk c (1 KiB = 4 rows of 64 floats | 8 rows of 32) of y(g-1), staged in the buffer at the last boundary, is read
//     back (ds_read_b128) and stored (global_store_dwordx4, whole 128-byte lines) at the end of tile c's MFMA burst,
//   * and right behind it the same 1 KiB of the buffer is refilled with chunk c of v(g+1) by an LDS-DMA
//     (global_load_lds_dwordx4: no VGPRs, no instructions at arrival),
// so that a group's 32 KiB of HBM traffic is spread over the ~33 k cycles of a walk (chip-wide ~4.6 TB/s, continuous)
// instead of two bursts.  At the group boundary nothing goes to memory: y(g) is rebuilt from the B-operand registers
// and written into the buffer in place of v(g+1), which has just been read into registers.
//
// vmcnt retires in order and counts stores, and the A stream (rolling register buffer, hand-placed loads, counted
// waits: rayen_mfma_split.hip) shares the counter: the two I/O operations of a tile are issued BEHIND the tile's last
// A re-load, so the first wait that has to cover them is the first K-step of the tile after the next (a full tile
// plus an epilogue later, ~2.8 k cycles); the waits of the next tile are counted past them (vmcnt(NS - 1 + 2)).  The
// count of a wait is an immediate, so EVERY tile issues exactly two vector-memory operations there: where there is
// nothing to store (a wave's first group) or nothing to fetch (its last), a 4-byte LDS-DMA of a cached word into a
// scratch slot takes the place (no VGPR destination: nothing the compiler could hand out while it is in flight).
// The walk therefore has ONE instruction stream whatever the round (three instances of it, one per case, made hipcc
// spill ~115 VGPRs, and a scratch reload inside the walk is a compiler-counted vmcnt(0): the trickle would drain).
//
// LDS image of a group's rows: the buffer holds the rows as 16-byte pieces; block j (1 KiB) = RPB consecutive rows,
// and inside the block piece p of row r sits in slot s = p ^ x(j, r) of that row (x: see swz()).  LDS-DMA writes
// lane-linear (lane L -> byte 16 L of the block), so the swizzle is applied to the per-lane SOURCE address; the
// fragment-shaped reads (lane (col, hi) reads piece 2 q + hi of row col) are then conflict-free ds_read_b128s, and
// every DMA / store instruction still covers whole 128-byte lines (the XOR permutes pieces inside aligned lines).
//
// Served shapes: NA_E = I, n = k = 32 NKK exactly, rows 16-byte aligned, at most 8 aux rows at NKK = 2 (LDS: 8 waves
// x 16 KiB of rows + the aux patch), n_items >= blocks per group.  Everything else stays on rayen_mfma_pair.hip.
#include "rayen_split_image.h"

#include <algorithm>
#include <mutex>
#include <type_traits>

namespace rayen {

namespace {

// issue priority of a wave inside the tile walk (developer A/B: -DRAYEN_IO_PRIO=n)
//   0  burst at priority 0, epilogue at 1 (rounds 2-4)          1  no priority changes at all
//   2  STATIC: the first wave of every SIMD (waves 0-3 of the workgroup) at 2, its partner at 0, no per-tile flips
//   3  burst at 1, epilogue at 0
#ifndef RAYEN_IO_PRIO
#define RAYEN_IO_PRIO 0
#endif
__device__ __forceinline__ void prio_burst() {
  if constexpr (RAYEN_IO_PRIO == 0) __builtin_amdgcn_s_setprio(0);
  else if constexpr (RAYEN_IO_PRIO == 3) __builtin_amdgcn_s_setprio(1);
}
__device__ __forceinline__ void prio_epilogue() {
  if constexpr (RAYEN_IO_PRIO == 0) __builtin_amdgcn_s_setprio(1);
  else if constexpr (RAYEN_IO_PRIO == 3) __builtin_amdgcn_s_setprio(0);
}
__device__ __forceinline__ void prio_static(const int wave) {
  if constexpr (RAYEN_IO_PRIO == 2) {
    if (wave < 4) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(0);
  }
}

template <int NKK>
struct IoGeom {
  static constexpr int NT = 2;
  static constexpr int PIECES = NKK * 8;          // 16-byte pieces per row
  static constexpr int RPB = 64 / PIECES;         // rows per 1-KiB block
  static constexpr int NBLK = NT * 32 / RPB;      // blocks per group of 64 rows
  static constexpr int SWB = NKK == 2 ? 64 : 32;  // byte weight of (block & 3) in the piece swizzle
  static constexpr int BYTES = NT * 32 * NKK * 128;
  static constexpr int AUXR = NKK == 2 ? 8 : 32;  // aux rows the patch holds
};

// slot swizzle of row r of block j (see the header): the 16 rows a ds_read_b128 lane group touches get 16 distinct
// 16-byte bank slots
template <int NKK>
__device__ __forceinline__ int swz(const int j, const int r) {
  if constexpr (NKK == 2) return (4 * (j & 3) + r) & 15;
  else return 2 * (j & 3) + ((r >> 1) & 1);
}

__device__ __forceinline__ const char* uniform_ptr(const void* p) {
  const uint64_t x = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)x), hi = __builtin_amdgcn_readfirstlane((uint32_t)(x >> 32));
  return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
}

// one LDS-DMA: every lane fetches 16 bytes from gbase + voff; lane L lands at LDS byte lds + 16 L
// (M0 = the LDS base; written in the statement that reads it, restored behind it)
__device__ __forceinline__ void dma16(const char* gbase, const unsigned voff, const unsigned lds) {
  unsigned keep;
  uint64_t asm_base;
  asm volatile(RAYEN_ASM_BASE_COPY "s_mov_b32 %[k], m0\n\ts_mov_b32 m0, %[lds]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[off], " RAYEN_ASM_BASE "\n\ts_mov_b32 m0, %[k]"
               : [k] "=&s"(keep), [b] "=&s"(asm_base)
               : [off] "v"(voff), [base] "s"(gbase), [lds] "s"(lds)
               : "memory");
}

// the stand-in: 4 bytes per lane of a cached word into a scratch slot of LDS (keeps the per-tile operation count)
__device__ __forceinline__ void dma4_dummy(const char* gbase, const unsigned voff, const unsigned lds) {
  unsigned keep;
  uint64_t asm_base;
  asm volatile(RAYEN_ASM_BASE_COPY "s_mov_b32 %[k], m0\n\ts_mov_b32 m0, %[lds]\n\ts_nop 0\n\tglobal_load_lds_dword %[off], " RAYEN_ASM_BASE "\n\ts_mov_b32 m0, %[k]"
               : [k] "=&s"(keep), [b] "=&s"(asm_base)
               : [off] "v"(voff), [base] "s"(gbase), [lds] "s"(lds)
               : "memory");
}

__device__ __forceinline__ void store16_nt(char* gbase, const unsigned voff, const f32x4 x) {
  uint64_t asm_base;
  asm volatile(RAYEN_ASM_BASE_COPY "global_store_dwordx4 %[off], %[x], " RAYEN_ASM_BASE " nt\n\ts_nop 1"
               : [b] "=&s"(asm_base)
               : [off] "v"(voff), [x] "v"(x), [base] "s"(gbase)
               : "memory");
}

}  // namespace

// developer build (scripts/ubench/tu_variant.sh rayen_mfma_pair_io iostamps -DRAYEN_IO_STAMPS; scripts/ubench/io_stamps.py):
// s_memtime per tile of BOTH groups of waves 0 / 3 / 4 of every 64th workgroup -- [tile top | burst + row operations issued |
// epilogue done] -- and the group's top, end of walk, end of drain, end of the boundary code.  Nothing in the library build.
#ifdef RAYEN_IO_STAMPS
__device__ unsigned long long io_stamp_buf[16 * 3 * 2 * 32 * 4];
extern "C" int rayen_debug_io_stamps(void* dst, size_t bytes) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(io_stamp_buf), bytes < sizeof(io_stamp_buf) ? bytes : sizeof(io_stamp_buf)) == hipSuccess ? 0 : -1;
}
#define RAYEN_IO_STAMP(tile, slot)                                                                                   \
  do {                                                                                                               \
    if (stamp_on && lane == 0 && (tile) < 32)                                                                        \
      io_stamp_buf[(((stamp_row * 2 + stamp_round) * 32) + (tile)) * 4 + (slot)] = __builtin_amdgcn_s_memtime();     \
  } while (0)
#else
#define RAYEN_IO_STAMP(tile, slot) do { } while (0)
#endif

// developer build (-DRAYEN_IO_CLOCK; scripts/ubench/wl_clock.py): s_memtime and s_memrealtime (100 MHz) at entry and exit of wave
// 0 of every 16th workgroup -- the shader clock the kernel actually ran at.  Nothing in the library build.
#ifdef RAYEN_IO_CLOCK
__device__ unsigned long long io_clock_buf[16 * 4];
extern "C" int rayen_debug_io_clock(void* dst, size_t bytes) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(io_clock_buf), bytes < sizeof(io_clock_buf) ? bytes : sizeof(io_clock_buf)) == hipSuccess ? 0 : -1;
}
#endif

template <int NKK, bool TRACK>
__global__ __launch_bounds__(kMfmaWaves * 64, kMfmaWaves / 4) void mfma_pair_io_kernel(
    const f16x8* __restrict__ Wh, const MItem* __restrict__ items, int n_items,
    const MPack* __restrict__ packs, const float* __restrict__ y0, int k, int n,
    const float* __restrict__ v, int64_t B, int64_t ldv, float* __restrict__ y, int64_t ldy,
    float* __restrict__ kappa_out, int32_t* __restrict__ active_out,
    int32_t* __restrict__ nan_flag, const float w_scale, const float w_inv) {
  using G = IoGeom<NKK>;
  constexpr int NT = G::NT, NS = NKK * 2, NCH = NS * 2, KK = NKK * 16, NQ = NKK * 4;
  constexpr int PIECES = G::PIECES, RPB = G::RPB, NBLK = G::NBLK, SWB = G::SWB, AUXR = G::AUXR;
  __shared__ float aux_lds[kMfmaWaves][NT][AUXR][32];
  __shared__ __attribute__((aligned(16))) float y0_lds[NKK * 32];
  __shared__ __attribute__((aligned(1024))) char io_lds[kMfmaWaves][G::BYTES];
  __shared__ __attribute__((aligned(256))) char sink_lds[kMfmaWaves][256];   // where the stand-in operations land

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  prio_static(wave);
  const int col = lane & 31;
  const int hi = lane >> 5;
  const int64_t n_groups = (B + NT * 32 - 1) / (NT * 32);
  const int64_t wave_id = (int64_t)blockIdx.x * kMfmaWaves + wave;
  const int64_t wave_stride = (int64_t)gridDim.x * kMfmaWaves;
  bool bad = false;
#ifdef RAYEN_IO_STAMPS
  const int stamp_wsel = wave == 0 ? 0 : wave == 3 ? 1 : wave == 4 ? 2 : -1;
  const int stamp_row = (int)(blockIdx.x >> 6) * 3 + stamp_wsel;           // 16 workgroups x 3 waves at most
  const bool stamp_on = (blockIdx.x & 63) == 0 && (blockIdx.x >> 6) < 16 && stamp_wsel >= 0;
  int stamp_round = -1;
  if (stamp_on && lane == 0) io_stamp_buf[(((stamp_row * 2 + 0) * 32) + 31) * 4 + 3] = __builtin_amdgcn_s_memtime();   // kernel entry
#endif
#ifdef RAYEN_IO_CLOCK
  const bool clk_probe = (blockIdx.x & 15) == 0 && threadIdx.x == 0;
  if (clk_probe) {
    io_clock_buf[(blockIdx.x >> 4) * 4 + 0] = __builtin_amdgcn_s_memtime();
    io_clock_buf[(blockIdx.x >> 4) * 4 + 1] = __builtin_amdgcn_s_memrealtime();
  }
#endif
  for (int i = threadIdx.x; i < NKK * 32; i += kMfmaWaves * 64) y0_lds[i] = y0[i];
  __syncthreads();  // the only workgroup barrier
  (void)k; (void)n;

  char* const io = io_lds[wave];
  const unsigned io_addr = __builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<uintptr_t>(io));  // LDS byte address
  const unsigned sink_addr = __builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<uintptr_t>(&sink_lds[wave][0]));

  // This lane as the reader / writer of ITS samples' rows (fragment shape): piece 2 q + hi of row 32 t + col sits at
  // byte base + ((32 q) ^ xh).  Recomputed from the lane number wherever it is needed: as loop invariants hipcc keeps
  // all sixteen piece addresses live across the walk and spills them (the `asm` makes the lane number opaque).
  auto row_addr = [&](const int t, unsigned& base, unsigned& xh) {
    int l = lane;
    asm volatile("" : "+v"(l));
    const int row = 32 * t + (l & 31), j = row / RPB, r = row % RPB;
    base = (unsigned)(j * 1024 + r * PIECES * 16);
    xh = (unsigned)((swz<NKK>(j, r) ^ (l >> 5)) * 16);
  };
  // this lane as a DMA / store lane of block c: row c RPB + io_r, piece (io_s ^ x(c, io_r))
  const int io_r = lane / PIECES, io_s = lane % PIECES;
  const unsigned ps16 = (unsigned)((NKK == 2 ? (io_s ^ io_r) : (io_s ^ ((io_r >> 1) & 1))) * 16);
  const unsigned ldvB = (unsigned)ldv * 4u, ldyB = (unsigned)ldy * 4u;
  const unsigned lane_rv = (unsigned)io_r * ldvB, lane_ry = (unsigned)io_r * ldyB;

  // ---- A operands: the rolling register buffer of rayen_mfma_split.hip, two chunks (a1, a2) per K-step
  u32x4 abuf[NCH];
  const unsigned lane_off = lane * 16;
#pragma unroll
  for (int sp = 0; sp < NS; ++sp) {
    const char* sb = reinterpret_cast<const char*>(Wh) + sp * 2048;
    uint64_t asm_base;
    asm volatile(RAYEN_ASM_BASE_COPY "global_load_dwordx4 %[d], %[off], " RAYEN_ASM_BASE "" : [d] "=v"(abuf[2 * sp + 0]), [b] "=&s"(asm_base) : [off] "v"(lane_off), [base] "s"(sb));
    asm volatile(RAYEN_ASM_BASE_COPY "global_load_dwordx4 %[d], %[off], " RAYEN_ASM_BASE " offset:1024" : [d] "=v"(abuf[2 * sp + 1]), [b] "=&s"(asm_base) : [off] "v"(lane_off), [base] "s"(sb));
  }

  // the whole group in one burst (first group of a wave; a ragged last group): rows beyond B are not requested
  auto burst_load = [&](const int64_t s_base) {
    const char* vg = uniform_ptr(v + s_base * ldv);
#pragma nounroll
    for (int c = 0; c < NBLK; ++c)
      if (s_base + c * RPB + io_r < B)
        dma16(vg, (unsigned)(c * RPB) * ldvB + lane_rv + (ps16 ^ (unsigned)(SWB * (c & 3))), io_addr + c * 1024);
  };
  auto burst_store = [&](const int64_t s_base) {
    char* yg = const_cast<char*>(uniform_ptr(y + s_base * ldy));
#pragma unroll 4
    for (int c = 0; c < NBLK; ++c) {
      const f32x4 o = *reinterpret_cast<const f32x4*>(io + c * 1024 + lane * 16);
      if (s_base + c * RPB + io_r < B)
        __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(
            yg + ((unsigned)(c * RPB) * ldyB + lane_ry + (ps16 ^ (unsigned)(SWB * (c & 3))))));
    }
  };

  // vb[t][piece][k-step] = 8 f16 = the B operand of one MFMA; element i = column 16 sp + 8 (i >> 2) + 4 hi + (i & 3)
  f16x8 vb[NT][2][NS];
  float v_scl[NT], v_inv[NT];
  bool live[NT];
  // rows of sample tile t: LDS -> registers (fp32) ...
  auto read_rows = [&](float (&vr)[KK], const int t) {
    unsigned base, xh;
    row_addr(t, base, xh);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const f32x4 x = *reinterpret_cast<const f32x4*>(io + base + ((unsigned)(32 * q) ^ xh));
      vr[4 * q + 0] = x[0];
      vr[4 * q + 1] = x[1];
      vr[4 * q + 2] = x[2];
      vr[4 * q + 3] = x[3];
    }
  };
  // ... -> scaled f16 pairs.  sv = 2^(13 - floor(log2 max|v|)): exponent arithmetic only (rayen_mfma_pair.hip)
  // (two instructions per value, rayen_split_image.h::pair_split_lo/_hi; nan_row[t]: a NaN or Inf among the row's components)
  bool nan_row[NT];
  auto split_rows = [&](const float (&vr)[KK], const int t) {
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < KK; ++i) m = fmaxf(m, __builtin_fabsf(vr[i]));
    m = fmaxf(m, xhalf(m));
    float sv;
    int sv_exp;
    pow2_scale(m, sv, v_inv[t], sv_exp);
    v_scl[t] = sv;
    f16x2 z = {(_Float16)0.f, (_Float16)0.f};
#pragma unroll
    for (int sp = 0; sp < NS; ++sp) {
      u32x4 w1, w2;
#pragma unroll
      for (int j = 0; j < 4; ++j) {           // register j of the K-step: elements i = 2 j, 2 j + 1 = columns q = 2 sp + (j >> 1), c = 2 (j & 1) + {0, 1}
        const int q = 2 * sp + (j >> 1), c = 2 * (j & 1);
        unsigned a, b;
        pair_split_lo(a, b, vr[4 * q + c], sv);
        pair_split_hi(a, b, vr[4 * q + c + 1], sv);
        w1[j] = a;
        w2[j] = b;
        pair_nan_fold(z, a);
      }
      vb[t][0][sp] = __builtin_bit_cast(f16x8, w1);
      vb[t][1][sp] = __builtin_bit_cast(f16x8, w2);
    }
    nan_row[t] = pair_nan_seen(z);
  };
  auto fetch_tile = [&](const int t) {
    float vr[KK];
    __builtin_amdgcn_sched_barrier(0);
    read_rows(vr, t);
    __builtin_amdgcn_sched_barrier(0);
    split_rows(vr, t);
    __builtin_amdgcn_sched_barrier(0);
  };

  // every in-flight vector-memory operation of this wave has retired (the chunks of the A buffer are named: they
  // are asm-loaded, the compiler must not move them before this point)
  auto drain = [&]() {
    if constexpr (NCH == 8)
      asm volatile("s_waitcnt vmcnt(0)"
                   : "+v"(abuf[0]), "+v"(abuf[1]), "+v"(abuf[2]), "+v"(abuf[3]), "+v"(abuf[4]), "+v"(abuf[5]), "+v"(abuf[6]), "+v"(abuf[7])
                   :
                   : "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(abuf[0]), "+v"(abuf[1]), "+v"(abuf[2]), "+v"(abuf[3]) : : "memory");
  };

  int64_t grp = wave_id;
  bool need_fetch = false;   // the buffer holds the rows of group `grp`, still to be split
#ifdef RAYEN_IO_STAGGER
  // developer build: the second wave of every SIMD (waves 4..7 of the workgroup) asks for its first rows RAYEN_IO_STAGGER
  // x 64 clocks later -- the first waves' 16 MB then arrive in half the time and their walks start while the others' load
  if ((threadIdx.x >> 6) >= 4) __builtin_amdgcn_s_sleep(RAYEN_IO_STAGGER);
#endif
  if (grp < n_groups) {
    burst_load(grp * (NT * 32));
    drain();
    need_fetch = true;
  }
  bool has_prev = false;
  int64_t prev_base = 0;

  while (grp < n_groups) {
    const int64_t s_base = grp * (NT * 32);
    const int64_t next = grp + wave_stride;
    const bool has_next = next < n_groups;
    const bool trickle_ld = has_next && (next + 1) * (NT * 32) <= B;   // whole groups only: every lane takes part
    const char* vnext = uniform_ptr(v + (has_next ? next : grp) * (NT * 32) * ldv);
    char* yprev = const_cast<char*>(uniform_ptr(y + prev_base * ldy));
#ifdef RAYEN_IO_STAMPS
    stamp_round = stamp_round < 1 ? stamp_round + 1 : 1;
#endif
    RAYEN_IO_STAMP(31, 0);
    if (need_fetch) {   // a wave's first group, or a ragged one (requested in one burst at the last boundary)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        live[t] = (s_base + t * 32 + col) < B;
        fetch_tile(t);
      }
    }

    // kap, part, the aux patch and the accumulators live in the SCALED domain (gW sv times the natural value)
    float kap[NT], part[NT], scale[NT], knat[NT];
    int acode[NT];  // arg-max bookkeeping in one register: (segment << 20) | row, -1 = none
#pragma unroll
    for (int t = 0; t < NT; ++t) { kap[t] = 0.f; part[t] = 0.f; scale[t] = 1.f; knat[t] = 0.f; acode[t] = -1; }

    {
      f32x16 acc[NT];
      // (which tile an item reads and which K-steps of it -- rayen_tiles.h -- come with the PREVIOUS item's record,
      // MItem::qbegin: nothing in front of a burst waits for a scalar load)
      int ts_next = items[0].tile_shape;
      bool after_half_b = false;   // the previous item was the second half of a shared tile
      for (int it = 0; it < n_items; ++it) {
        const MItem item = items[it];
        const int ts = ts_next;
        ts_next = item.qbegin;
        // the tile after this one; the last tile of a group fetches tile 0 for the next group
        const char* next_tile = reinterpret_cast<const char*>(Wh) + (size_t)(((ts >> 30) & 1) ? 0 : (ts & 0xFFFFFF) + 1) * (NCH * 1024);
        const bool slot = it < NBLK;
        f32x4 ytmp = {0.f, 0.f, 0.f, 0.f};
        RAYEN_IO_STAMP(it, 0);
        if (slot && has_prev) ytmp = *reinterpret_cast<const f32x4*>(io + it * 1024 + lane * 16);
        {
          const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          prio_burst();
          // Two passes over the K-steps, by product size (rayen_mfma_pair.hip).  Which K-steps of the tile an item
          // multiplies (`k_*`) and streams (`s_*`): a full tile all of them; the halves of a shared tile (NS = 4,
          // rayen_tiles.h) K-steps 0,1 or 2,3; a block alone in its tile multiplies K-steps 2,3 and streams all.  ONE
          // instruction stream: a K-step -- wait, MFMAs, re-load -- is a single statement with its branches inside
          // (rayen_split_image.h::pair_kstep1 / pair_kstep2).  The counted waits step over the two I/O operations the
          // previous ITEM issued behind its last re-load (tile 0: everything it needs landed before the boundary's
          // vmcnt(0)): behind the chunks of K-step sp there are -- full behind full NS - 1 re-loads + 2 = 5; a first half
          // (always behind a full item) the same 5; a second half the first half's 4 + 2 on top of a tile's remaining
          // 3: 9; K-steps 0,1 of the full item behind a shared tile 9 likewise (their chunks were re-loaded by the FIRST
          // half), K-steps 2,3 of it 5.
          if constexpr (NS == 4) {
            const int shape = (ts >> 24) & 3;
            const int ctrl = pair_item_ctrl(shape, after_half_b);
            auto step1 = [&](auto SP) {
              constexpr int sp = decltype(SP)::value;
              f16x8 b1[NT], b2[NT];
#pragma unroll
              for (int t = 0; t < NT; ++t) { b1[t] = vb[t][0][sp]; b2[t] = vb[t][1][sp]; }
              __builtin_amdgcn_sched_barrier(0);
              pair_kstep1<NT, sp, 5, 9>(abuf[2 * sp + 0], abuf[2 * sp + 1], acc, b1, b2, ctrl, next_tile + (2 * sp + 1) * 1024, lane_off);
              __builtin_amdgcn_sched_barrier(0);
            };
            auto step2 = [&](auto SP) {
              constexpr int sp = decltype(SP)::value;
              f16x8 b1[NT];
#pragma unroll
              for (int t = 0; t < NT; ++t) b1[t] = vb[t][0][sp];
              __builtin_amdgcn_sched_barrier(0);
              pair_kstep2<NT, sp, sp == NS - 1>(abuf[2 * sp + 0], acc, b1, ctrl, next_tile + (2 * sp + 0) * 1024, lane_off);
              __builtin_amdgcn_sched_barrier(0);
            };
            step1(std::integral_constant<int, 0>{}); step1(std::integral_constant<int, 1>{});
            step1(std::integral_constant<int, 2>{}); step1(std::integral_constant<int, 3>{});
            step2(std::integral_constant<int, 0>{}); step2(std::integral_constant<int, 1>{});
            step2(std::integral_constant<int, 2>{}); step2(std::integral_constant<int, 3>{});
            after_half_b = shape == MS_HALF_B;
          } else {
            // n_pad = 32: every item is a full tile; the builtins, scheduled by hipcc (rounds 3-4)
#pragma unroll
            for (int sp = 0; sp < NS; ++sp) {
              __builtin_amdgcn_sched_barrier(0);
              asm volatile("s_waitcnt vmcnt(3)" : "+v"(abuf[2 * sp + 0]), "+v"(abuf[2 * sp + 1]));
              const f16x8 a1 = __builtin_bit_cast(f16x8, abuf[2 * sp + 0]), a2 = __builtin_bit_cast(f16x8, abuf[2 * sp + 1]);
#pragma unroll
              for (int t = 0; t < NT; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, vb[t][0][sp], sp == 0 ? zero : acc[t], 0, 0, 0);
#pragma unroll
              for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, vb[t][1][sp], acc[t], 0, 0, 0);
              __builtin_amdgcn_sched_barrier(0);
              pair_reload(abuf[2 * sp + 1], next_tile + (2 * sp + 1) * 1024, lane_off);
              __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int sp = 0; sp < NS; ++sp) {
              const f16x8 a1 = __builtin_bit_cast(f16x8, abuf[2 * sp + 0]);
#pragma unroll
              for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, vb[t][0][sp], acc[t], 0, 0, 0);
              __builtin_amdgcn_sched_barrier(0);
              pair_reload(abuf[2 * sp + 0], next_tile + (2 * sp + 0) * 1024, lane_off);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
          // ---- the tile's two I/O operations: 1 KiB of y(prev) out, 1 KiB of v(next) in (or their stand-ins)
          {
            const unsigned px = ps16 ^ (unsigned)(SWB * (it & 3));
            if (slot && has_prev) store16_nt(yprev, (unsigned)(it * RPB) * ldyB + lane_ry + px, ytmp);
            else dma4_dummy(reinterpret_cast<const char*>(Wh), lane_off >> 2, sink_addr);
            if (slot && trickle_ld) dma16(vnext, (unsigned)(it * RPB) * ldvB + lane_rv + px, io_addr + it * 1024);
            else dma4_dummy(reinterpret_cast<const char*>(Wh), lane_off >> 2, sink_addr);
          }
          __builtin_amdgcn_sched_barrier(0);
          prio_epilogue();
          RAYEN_IO_STAMP(it, 1);
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
          // a running sum of squares over the segment's tiles, closed on its last tile
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            f32x2 s2 = {(item.flags & MF_FIRST) ? 0.f : part[t], 0.f};
#pragma unroll
            for (int g = 0; g < 16; g += 2) {
              const f32x2 a2 = {acc[t][g], acc[t][g + 1]};
              s2 = __builtin_elementwise_fma(a2, a2, s2);
            }
            part[t] = s2[0] + s2[1];
          }
          if (item.flags & MF_LAST) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
              const float total = part[t] + xhalf(part[t]);
              const float a0 = aux_lds[wave][t][item.aux][col];
              float kc;
              if (item.type != MI_SOC) {
                kc = (a0 + __builtin_amdgcn_sqrtf(fmaxf(total, 0.f))) * item.seg_inv;   // (the segment's own power of two undone)
              } else {
                kc = pair_soc_candidate(a0, aux_lds[wave][t][item.aux + 1][col], total, w_inv * item.seg_inv, v_inv[t],
                                        item.f0, item.f1, v_scl[t], w_scale);
              }
              if (kc > kap[t]) { kap[t] = kc; acode[t] = item.seg << 20; }
            }
          }
        } else if (item.type == MI_AUX) {
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int g = 0; g < (AUXR == 8 ? 4 : 16); ++g)
              aux_lds[wave][t][(g & 3) + 8 * (g >> 2) + 4 * hi][col] = acc[t][g];
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
              const float kc = (aux_lds[wave][t][slot & (AUXR - 1)][col] + __builtin_amdgcn_sqrtf(qs)) * (hi ? pk.inv[a][1] : pk.inv[a][0]);
              if (sid >= 0 && kc > kap[t]) { kap[t] = kc; acode[t] = sid << 20; }
            }
          }
        }
        RAYEN_IO_STAMP(it, 2);
      }
    }
    RAYEN_IO_STAMP(31, 1);
    // every row of v(next) has landed, y(prev) is out, the next group's first tile is in the A buffer
    drain();
    RAYEN_IO_STAMP(31, 2);

    // ---- kappa is final
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

    // y = y0 + v / max(1, kappa): v rebuilt from its pieces (22 bits of it; scaled by sv, undone by `scale`), written
    // into the buffer in place of the row it came from
    auto stage_tile = [&](const int t) {
      unsigned base, xh;
      row_addr(t, base, xh);
      float one = 1.0f;
      asm volatile("" : "+v"(one));
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const f32x4 o4 = *reinterpret_cast<const f32x4*>(&y0_lds[8 * q + 4 * hi]);
        f32x4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int i = (q & 1) * 4 + c;
          const unsigned w1 = __builtin_bit_cast(u32x4, vb[t][0][q >> 1])[i >> 1], w2 = __builtin_bit_cast(u32x4, vb[t][1][q >> 1])[i >> 1];
          const float val = (i & 1) ? pair_rebuild_hi(w1, w2, one) : pair_rebuild_lo(w1, w2, one);     // fl32(p1 + p2)
          o[c] = fmaf(val, scale[t], o4[c]);
        }
        *reinterpret_cast<f32x4*>(io + base + ((unsigned)(32 * q) ^ xh)) = o;
      }
      bad |= live[t] && nan_row[t];      // (the row's own components: y is NaN exactly when one of them is NaN or Inf)
    };

    // The buffer holds v(next) when its rows were trickled in: per sample tile, read the next rows, write this
    // group's y over them, split.  (ONE copy of the staging code for both cases: written once per arm, hipcc hoists the
    // arms' common arithmetic -- all 64 outputs of a lane -- in front of the branch and spills it.)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float vr[KK];
      __builtin_amdgcn_sched_barrier(0);
      if (trickle_ld) read_rows(vr, t);
      __builtin_amdgcn_sched_barrier(0);
      stage_tile(t);
      __builtin_amdgcn_sched_barrier(0);
      if (trickle_ld) {
        split_rows(vr, t);
        live[t] = true;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    has_prev = trickle_ld;
    prev_base = s_base;
    need_fetch = false;
    RAYEN_IO_STAMP(30, 0);
    if (!trickle_ld) {
      burst_store(s_base);
      RAYEN_IO_STAMP(30, 1);
      if (has_next) {   // a ragged last group: requested only now, in one burst
        burst_load(next * (NT * 32));
        drain();
        need_fetch = true;
      }
    }
    grp = next;
  }  // persistent loop over sample groups
  if (nan_flag && bad) atomicOr(nan_flag, 1);
#ifdef RAYEN_IO_CLOCK
  if (clk_probe) {
    io_clock_buf[(blockIdx.x >> 4) * 4 + 2] = __builtin_amdgcn_s_memtime();
    io_clock_buf[(blockIdx.x >> 4) * 4 + 3] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

// =============================================================================================
// The same schedule for n <= 32 with rows stored back to back (ldv == n, ldy == k: what a contiguous torch tensor
// is), any n, and for sets with equality constraints (k > n, rows of y from the NA_E tiles): config-5-like shapes.
//
// A group's 64 rows are ONE contiguous block of memory (256 n bytes of v, 256 k bytes of y), so the buffer holds a
// flat copy: chunk c = bytes [1024 c, 1024 c + 1024) of the block, fetched / stored by one instruction as 64 pieces
// of 16 bytes whatever n and k are -- every store covers whole 128-byte lines, where rayen_mfma_pair.hip writes the
// 180-byte rows of config 5 with 4-byte stores at a stride of 45 floats (83 MB of HBM writes for 47 MB of y).  The
// fragment-shaped accesses (a lane reads its sample's row, writes its sample's outputs) are 4-byte LDS operations on
// the flat block.  With NA_E != I the rows of y come out of the NA_E tiles at the END of the walk, while the buffer
// still holds v(next): the next rows are read into registers in front of the first of those tiles.
// LDS (dynamic): the aux patch (64 KiB) + 8 x (256 max(n, k)) bytes.
// developer build (scripts/ubench/tu_variant.sh rayen_mfma_pair_io stamps -DRAYEN_IOF_STAMPS; scripts/ubench/iof_stamps.py):
// s_memtime per tile of the flat-row kernel's SECOND group, waves 0 / 3 / 4 of every 64th workgroup --
// [tile top | burst + row operations issued | epilogue done], and the group boundary.  Nothing in the library build.
#ifdef RAYEN_IOF_STAMPS
__device__ unsigned long long iof_stamp_buf[16 * 3 * 256 * 4];
extern "C" int rayen_debug_iof_stamps(void* dst, size_t bytes) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(iof_stamp_buf), bytes < sizeof(iof_stamp_buf) ? bytes : sizeof(iof_stamp_buf)) == hipSuccess ? 0 : -1;
}
#define RAYEN_IOF_STAMP(tile, slot)                                                                                  \
  do {                                                                                                               \
    if (stamp_on && lane == 0 && (tile) < 256)                                                                       \
      iof_stamp_buf[((stamp_row * 256) + (tile)) * 4 + (slot)] = __builtin_amdgcn_s_memtime();                       \
  } while (0)
#else
#define RAYEN_IOF_STAMP(tile, slot) do { } while (0)
#endif

template <bool TRACK, bool STAGED>
__global__ __launch_bounds__(kMfmaWaves * 64, kMfmaWaves / 4) void mfma_pair_iof_kernel(
    const f16x8* __restrict__ Wh, const MItem* __restrict__ items, int n_items,
    const MPack* __restrict__ packs, const float* __restrict__ y0, int k, int n,
    const float* __restrict__ v, int64_t B, float* __restrict__ y,
    float* __restrict__ kappa_out, int32_t* __restrict__ active_out,
    int32_t* __restrict__ nan_flag, const float w_scale, const float w_inv, const int buf_bytes, const int first_out) {
  constexpr int NT = 2, NS = 2, NCH = 4, KK = 16, NQ = 4, AUXR = 32;
  extern __shared__ __attribute__((aligned(1024))) char iof_smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  prio_static(wave);
#ifdef RAYEN_IOF_STAMPS
  const int stamp_wsel = wave == 0 ? 0 : wave == 3 ? 1 : wave == 4 ? 2 : -1;
  const int stamp_row = (int)(blockIdx.x >> 6) * 3 + stamp_wsel;           // 16 workgroups x 3 waves at most
  const bool stamp_wave = (blockIdx.x & 63) == 0 && (blockIdx.x >> 6) < 16 && stamp_wsel >= 0;
  bool stamp_on = false;
  int stamp_round = 0;
#endif
  float (*aux_lds)[AUXR][32] = reinterpret_cast<float (*)[AUXR][32]>(iof_smem) + wave * NT;   // [t][row][sample]
  char* const io_all = iof_smem + kMfmaWaves * NT * AUXR * 32 * 4;
  char* const io = io_all + wave * buf_bytes;
  float* const io_f = reinterpret_cast<float*>(io);
  char* const sink = io_all + kMfmaWaves * buf_bytes;
  float* const y0_lds = reinterpret_cast<float*>(sink + 256);   // [32] (identity write-out) | [96] (NA_E tiles: k <= 64, padded)
  const int col = lane & 31;
  const int hi = lane >> 5;
  const int64_t n_groups = (B + NT * 32 - 1) / (NT * 32);
  const int64_t wave_id = (int64_t)blockIdx.x * kMfmaWaves + wave;
  const int64_t wave_stride = (int64_t)gridDim.x * kMfmaWaves;
  bool bad = false;
  // (y0 is zero-padded to a tile multiple + 32.  In LDS for the NA_E tiles too: a global read of y0 inside the walk makes
  // hipcc wait with vmcnt(0), which drains the trickled rows and the A prefetch -- twice per group on config 5)
  for (int i = threadIdx.x; i < (STAGED ? 96 : 32); i += kMfmaWaves * 64) y0_lds[i] = (!STAGED || i < ((k + 31) / 32) * 32 + 32) ? y0[i] : 0.f;
  __syncthreads();  // the only workgroup barrier

  const unsigned io_addr = __builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<uintptr_t>(io));
  const unsigned sink_addr = __builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<uintptr_t>(sink));
  const int pieces_v = 16 * n, pieces_y = 16 * k;                           // 16-byte pieces of a whole group's block
  const int chunks_v = (pieces_v + 63) / 64, chunks_y = (pieces_y + 63) / 64;

  // ---- A operands: the rolling register buffer of rayen_mfma_split.hip, two chunks (a1, a2) per K-step
  u32x4 abuf[NCH];
  const unsigned lane_off = lane * 16;
#pragma unroll
  for (int sp = 0; sp < NS; ++sp) {
    const char* sb = reinterpret_cast<const char*>(Wh) + sp * 2048;
    uint64_t asm_base;
    asm volatile(RAYEN_ASM_BASE_COPY "global_load_dwordx4 %[d], %[off], " RAYEN_ASM_BASE "" : [d] "=v"(abuf[2 * sp + 0]), [b] "=&s"(asm_base) : [off] "v"(lane_off), [base] "s"(sb));
    asm volatile(RAYEN_ASM_BASE_COPY "global_load_dwordx4 %[d], %[off], " RAYEN_ASM_BASE " offset:1024" : [d] "=v"(abuf[2 * sp + 1]), [b] "=&s"(asm_base) : [off] "v"(lane_off), [base] "s"(sb));
  }
  auto drain = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(abuf[0]), "+v"(abuf[1]), "+v"(abuf[2]), "+v"(abuf[3]) : : "memory");
  };

  // a whole group in one burst (a wave's first group): LDS-DMA of its chunks; a ragged group (the last of the batch)
  // word by word -- its last 16-byte piece could reach past the end of v
  auto burst_load = [&](const int64_t s_base) {
    const int64_t left = B - s_base;
    if (left >= NT * 32) {
      const char* vg = uniform_ptr(v + s_base * n);
#pragma nounroll
      for (int c = 0; c < chunks_v; ++c)
        if (c * 64 + lane < pieces_v) dma16(vg, (unsigned)(c * 1024) + lane_off, io_addr + c * 1024);
    } else {
      const float* vg = v + s_base * n;
      const int words = (int)left * n;
      for (int i = lane; i < words; i += 64) io_f[i] = vg[i];
    }
  };
  auto burst_store = [&](const int64_t s_base) {
    const int64_t left = B - s_base;
    if (left >= NT * 32) {
      char* yg = const_cast<char*>(uniform_ptr(y + s_base * k));
#pragma nounroll
      for (int c = 0; c < chunks_y; ++c) {
        if (c * 64 + lane < pieces_y) {
          const f32x4 o = *reinterpret_cast<const f32x4*>(io + c * 1024 + lane * 16);
          __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(yg + ((unsigned)(c * 1024) + lane_off)));
        }
      }
    } else {
      float* yg = y + s_base * k;
      const int words = (int)left * k;
      for (int i = lane; i < words; i += 64) yg[i] = io_f[i];
    }
  };

  // vb[t][piece][k-step] = 8 f16 = the B operand of one MFMA; element i = column 16 sp + 8 (i >> 2) + 4 hi + (i & 3)
  f16x8 vb[NT][2][NS];
  float v_scl[NT], v_inv[NT];
  bool live[NT];
  // this lane's part of row 32 t + col of the flat block: columns 8 q + 4 hi + c (zero beyond n)
  auto read_rows = [&](float (&vr)[KK], const int t) {
    int l = lane;
    asm volatile("" : "+v"(l));                       // (opaque: the addresses are not carried across the walk)
    const float* row = io_f + (32 * t + (l & 31)) * n + 4 * (l >> 5);
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int c = 0; c < 4; ++c) vr[4 * q + c] = (8 * q + 4 * (l >> 5) + c < n) ? row[8 * q + c] : 0.f;
  };
  bool nan_row[NT];
  auto split_rows = [&](const float (&vr)[KK], const int t) {
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < KK; ++i) m = fmaxf(m, __builtin_fabsf(vr[i]));
    m = fmaxf(m, xhalf(m));
    float sv;
    int sv_exp;
    pow2_scale(m, sv, v_inv[t], sv_exp);
    v_scl[t] = sv;
    f16x2 z = {(_Float16)0.f, (_Float16)0.f};
#pragma unroll
    for (int sp = 0; sp < NS; ++sp) {          // (two instructions per value: rayen_split_image.h::pair_split_lo/_hi)
      u32x4 w1, w2;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int q = 2 * sp + (j >> 1), c = 2 * (j & 1);
        unsigned a, b;
        pair_split_lo(a, b, vr[4 * q + c], sv);
        pair_split_hi(a, b, vr[4 * q + c + 1], sv);
        w1[j] = a;
        w2[j] = b;
        pair_nan_fold(z, a);
      }
      vb[t][0][sp] = __builtin_bit_cast(f16x8, w1);
      vb[t][1][sp] = __builtin_bit_cast(f16x8, w2);
    }
    nan_row[t] = pair_nan_seen(z);
  };

  int64_t grp = wave_id;
  bool need_fetch = false;   // the buffer holds the rows of group `grp`, still to be split
  if (grp < n_groups) {
    burst_load(grp * (NT * 32));
    drain();
    need_fetch = true;
  }
  bool has_prev = false;
  int64_t prev_base = 0;

  while (grp < n_groups) {
    const int64_t s_base = grp * (NT * 32);
    const int64_t next = grp + wave_stride;
    const bool has_next = next < n_groups;
    const bool trickle_ld = has_next && (next + 1) * (NT * 32) <= B;   // whole groups only
    const char* vnext = uniform_ptr(v + (has_next ? next : grp) * (NT * 32) * n);
    char* yprev = const_cast<char*>(uniform_ptr(y + prev_base * k));
#ifdef RAYEN_IOF_STAMPS
    stamp_on = stamp_wave && stamp_round == 1;
    ++stamp_round;
    RAYEN_IOF_STAMP(255, 0);
#endif
    if (need_fetch) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        float vr[KK];
        live[t] = (s_base + t * 32 + col) < B;
        __builtin_amdgcn_sched_barrier(0);
        read_rows(vr, t);
        __builtin_amdgcn_sched_barrier(0);
        split_rows(vr, t);
        __builtin_amdgcn_sched_barrier(0);
      }
    }

    // kap, part, the aux patch and the accumulators live in the SCALED domain (gW sv times the natural value)
    float kap[NT], part[NT], scale[NT], knat[NT];
    int acode[NT];  // arg-max bookkeeping in one register: (segment << 20) | row, -1 = none
#pragma unroll
    for (int t = 0; t < NT; ++t) { kap[t] = 0.f; part[t] = 0.f; scale[t] = 1.f; knat[t] = 0.f; acode[t] = -1; }
    float vnx[NT][KK];   // STAGED: the next group's rows, read in front of the first NA_E tile

    auto finish_kappa = [&]() {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float other = xhalf(kap[t]);
        if (TRACK) {
          const int ocode = __shfl_xor(acode[t], 32);
          if (other > kap[t] || (other == kap[t] && hi == 1)) acode[t] = ocode;
        }
        kap[t] = fmaxf(kap[t], other);
        knat[t] = (kap[t] * w_inv) * v_inv[t];
        // what multiplies the scaled numbers on the way out: the accumulators of NA_E rows carry gW sv, the rebuilt
        // direction of the NA_E = I write-out only sv
        const float out = v_inv[t] * (1.0f / fmaxf(1.0f, knat[t]));
        scale[t] = STAGED ? out * w_inv : out;
      }
    };

    {
      f32x16 acc[NT];
      for (int it = 0; it < n_items; ++it) {
        RAYEN_IOF_STAMP(it, 0);
        const MItem item = items[it];
        // (the index, not the descriptor: the scalar loads of `item` then wait behind the MFMA burst instead of in front of it)
        if (STAGED && it == first_out) {
          // kappa is final; the rows of v(next) leave the buffer (the DMAs that brought them were issued in the first
          // tiles of this walk: the counted waits of the tiles since have stepped over them), y(prev) left it long ago
          finish_kappa();
          if (trickle_ld) {
#pragma unroll
            for (int t = 0; t < NT; ++t) read_rows(vnx[t], t);
          }
          __builtin_amdgcn_wave_barrier();
        }
        // the tile after this one; the last tile of a group fetches tile 0 for the next group
        const char* next_tile = reinterpret_cast<const char*>(Wh) + (size_t)(it + 1 == n_items ? 0 : it + 1) * (NCH * 1024);
        const bool st = has_prev && it < chunks_y, ld = trickle_ld && it < chunks_v;
        f32x4 ytmp = {0.f, 0.f, 0.f, 0.f};
        if (st && it * 64 + lane < pieces_y) ytmp = *reinterpret_cast<const f32x4*>(io + it * 1024 + lane * 16);
        {
          const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          prio_burst();
          auto load_chunk = [&](const int idx) {
            const char* sb = next_tile + idx * 1024;
            uint64_t asm_base;
            asm volatile(RAYEN_ASM_BASE_COPY "global_load_dwordx4 %[d], %[off], " RAYEN_ASM_BASE "" : [d] "+v"(abuf[idx]), [b] "=&s"(asm_base) : [off] "v"(lane_off), [base] "s"(sb));
          };
#pragma unroll
          for (int sp = 0; sp < NS; ++sp) {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(3)" : "+v"(abuf[2 * sp + 0]), "+v"(abuf[2 * sp + 1]));
            const f16x8 a1 = __builtin_bit_cast(f16x8, abuf[2 * sp + 0]), a2 = __builtin_bit_cast(f16x8, abuf[2 * sp + 1]);
#pragma unroll
            for (int t = 0; t < NT; ++t)
              acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, vb[t][0][sp], sp == 0 ? zero : acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, vb[t][1][sp], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            load_chunk(2 * sp + 1);
            __builtin_amdgcn_sched_barrier(0);
          }
#pragma unroll
          for (int sp = 0; sp < NS; ++sp) {
            const f16x8 a1 = __builtin_bit_cast(f16x8, abuf[2 * sp + 0]);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, vb[t][0][sp], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            load_chunk(2 * sp + 0);
            __builtin_amdgcn_sched_barrier(0);
          }
          // ---- the tile's two I/O operations: 1 KiB of y(prev) out, 1 KiB of v(next) in (or their stand-ins)
          {
            const unsigned off = (unsigned)(it * 1024) + lane_off;
            const int piece = it * 64 + lane;
            if (st) { if (piece < pieces_y) store16_nt(yprev, off, ytmp); }
            else dma4_dummy(reinterpret_cast<const char*>(Wh), lane_off >> 2, sink_addr);
            if (ld) { if (piece < pieces_v) dma16(vnext, off, io_addr + it * 1024); }
            else dma4_dummy(reinterpret_cast<const char*>(Wh), lane_off >> 2, sink_addr);
          }
          __builtin_amdgcn_sched_barrier(0);
          prio_epilogue();
        }
        RAYEN_IOF_STAMP(it, 1);
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
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            f32x2 s2 = {(item.flags & MF_FIRST) ? 0.f : part[t], 0.f};
#pragma unroll
            for (int g = 0; g < 16; g += 2) {
              const f32x2 a2 = {acc[t][g], acc[t][g + 1]};
              s2 = __builtin_elementwise_fma(a2, a2, s2);
            }
            part[t] = s2[0] + s2[1];
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
            for (int g = 0; g < 16; ++g) aux_lds[t][(g & 3) + 8 * (g >> 2) + 4 * hi][col] = acc[t][g];
          __builtin_amdgcn_wave_barrier();
        } else if (item.type == MI_PACK) {
          // (round 4, measured with scripts/ubench/iof_stamps.py: this epilogue takes 1 550 cycles against 290 for a tile
          // of linear rows -- 18 % of config 5's walk.  It is NOT the eight exchanges: batching them and the aux reads
          // (all in flight at once, the descriptor's words pinned in SGPRs) is bit-identical and takes 1 580-1 630; one
          // stamp behind the descriptor load says `pk` -- a scalar load that depends on `item` -- arrives 375 cycles after the
          // burst; the other ~1 200 are the ~150 vector instructions of the eight candidates and their half-wave selects.  Without the pinning hipcc turned `hi ? pk.x[a][1] : pk.x[a][0]`
          // into an indexed load from a SCRATCH copy of `pk`: 5 600 cycles per tile.  Left as it was.)
          const MPack pk = packs[item.aux];
#ifdef RAYEN_IOF_STAMPS
          { int probe = pk.aux[0][0] + pk.seg[3][1]; asm volatile("" : "+s"(probe)); }   // (the descriptor has arrived)
          RAYEN_IOF_STAMP(it, 3);
#endif
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
              const float kc = (aux_lds[t][slot & 31][col] + __builtin_amdgcn_sqrtf(qs)) * (hi ? pk.inv[a][1] : pk.inv[a][0]);
              if (sid >= 0 && kc > kap[t]) { kap[t] = kc; acode[t] = sid << 20; }
            }
          }
        } else if (STAGED && item.type == MI_OUT) {
          // rows of NA_E: y = y0 + (N v) / max(1, kappa) into the flat block [64][k] (4-byte LDS stores at a stride of k
          // words: conflict-free for odd k); it leaves as whole lines during the next walk.
          // Round 4 (stamps: 3 300 cycles per tile): y0 of the tile's sixteen rows in ONE batch of reads -- it was a
          // ds_read_b32 + lgkmcnt(0) inside every element's predicate, 32 exposed LDS round trips per tile --, and a tile
          // that lies wholly inside the k rows (wave-uniform) stores without predicates.  Same fma, same values.
          float y0r[16];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 y4 = *reinterpret_cast<const f32x4*>(&y0_lds[item.row0 + 4 * hi + 8 * q]);   // (padded to a tile multiple)
#pragma unroll
            for (int c = 0; c < 4; ++c) y0r[4 * q + c] = y4[c];
          }
          const bool full = item.row0 + 32 <= k;
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            float* yrow = io_f + (32 * t + col) * k + item.row0 + 4 * hi;
            float o[16];
#pragma unroll
            for (int g = 0; g < 16; ++g) o[g] = fmaf(acc[t][g], scale[t], y0r[g]);
            // (Round 4 saw 16 rows of one column of y equal to y0 in 22 of 3 000 launches of config 5 at B = 655 360, from this
            // statement as hipcc's SLP vectoriser had packed it: `v_pk_fma_f32 o, acc, scale, y0 op_sel:[0,1,0]`, scale[t]
            // broadcast out of the HIGH half of the (scale[0], scale[1]) pair.  Round 5 found what that is: on gfx950 exactly
            // this operand selection -- low result = src0.lo x src1.HI -- reads src1 as 0 in lanes 48-63 now and then while an
            // MFMA is executing on the SIMD (the partner wave's is enough), so the product vanishes and o = y0;
            // scripts/ubench/pkfma_hazard.hip reproduces it in isolation, every other operand selection is clean.  The library
            // is built without the SLP vectoriser and scripts/check_packed_opsel.py audits every kernel's ISA for the form.)
            if (full) {
              bool nn = false;
#pragma unroll
              for (int g = 0; g < 16; ++g) {
                nn |= (o[g] != o[g]);
                yrow[(g & 3) + 8 * (g >> 2)] = o[g];
              }
              bad |= live[t] && nn;
            } else {
              // (the last, partial tile keeps its predicates: sending the rows beyond k to a dummy LDS word under a select
              // instead was slower -- 76.6 against 73.6 us on config 5 -- and not bit-identical on the record path)
#pragma unroll
              for (int g = 0; g < 16; ++g) {
                const int r = (g & 3) + 8 * (g >> 2);
                if (item.row0 + 4 * hi + r < k) {
                  bad |= live[t] && (o[g] != o[g]);
                  yrow[r] = o[g];
                }
              }
            }
          }
        }
        RAYEN_IOF_STAMP(it, 2);
      }
    }
    RAYEN_IOF_STAMP(255, 1);
    // every row of v(next) has landed, y(prev) is out, the next group's first tile is in the A buffer
    drain();
    RAYEN_IOF_STAMP(255, 2);

    if (!STAGED) finish_kappa();
    if (hi == 0) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (!live[t]) continue;
        const int64_t s = s_base + t * 32 + col;
        if (kappa_out) kappa_out[s] = knat[t];
        if (TRACK) { active_out[2 * s] = acode[t] >> 20; active_out[2 * s + 1] = acode[t] < 0 ? 0 : (acode[t] & 0xFFFFF); }
      }
    }

    if constexpr (STAGED) {
      // y(g) is in the buffer already; the next rows are in registers
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        __builtin_amdgcn_sched_barrier(0);
        if (trickle_ld) {
          split_rows(vnx[t], t);
          live[t] = true;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    } else {
      // y = y0 + v / max(1, kappa) from the B-operand registers into the flat block, in place of the rows of v(next)
      // (read first)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        float vr[KK];
        __builtin_amdgcn_sched_barrier(0);
        if (trickle_ld) read_rows(vr, t);
        __builtin_amdgcn_sched_barrier(0);
        {
          int l = lane;
          asm volatile("" : "+v"(l));
          float* row = io_f + (32 * t + (l & 31)) * k + 4 * (l >> 5);
          float one = 1.0f;
          asm volatile("" : "+v"(one));
#pragma unroll
          for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const int i = (q & 1) * 4 + c;
              const unsigned w1 = __builtin_bit_cast(u32x4, vb[t][0][q >> 1])[i >> 1], w2 = __builtin_bit_cast(u32x4, vb[t][1][q >> 1])[i >> 1];
              const float val = (i & 1) ? pair_rebuild_hi(w1, w2, one) : pair_rebuild_lo(w1, w2, one);     // fl32(p1 + p2)
              const float o = fmaf(val, scale[t], y0_lds[8 * q + 4 * (l >> 5) + c]);
              if (8 * q + 4 * (l >> 5) + c < k) row[8 * q + c] = o;
            }
          bad |= live[t] && nan_row[t];    // (columns beyond n are zero: the row's pieces are NaN / Inf exactly when y is NaN)
        }
        __builtin_amdgcn_sched_barrier(0);
        if (trickle_ld) {
          split_rows(vr, t);
          live[t] = true;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_wave_barrier();
    has_prev = trickle_ld;
    prev_base = s_base;
    need_fetch = false;
    if (!trickle_ld) {
      burst_store(s_base);
      if (has_next) {   // a ragged last group: requested only now
        __builtin_amdgcn_wave_barrier();
        burst_load(next * (NT * 32));
        drain();
        need_fetch = true;
      }
    }
    grp = next;
  }  // persistent loop over sample groups
  if (nan_flag && bad) atomicOr(nan_flag, 1);
}

// ---------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------
static int pair_iof_buf_bytes(const RayenPack* p) { return ((p->n > p->k ? p->n : p->k) * 256 + 255) / 256 * 256; }
static int pair_iof_lds_bytes(const RayenPack* p) {
  return kMfmaWaves * 2 * 32 * 32 * 4 + kMfmaWaves * pair_iof_buf_bytes(p) + 256 + 384;
}

// 0 = not served | 1 = rows at any 16-byte aligned stride, n = k = 32 NKK exactly (mfma_pair_io_kernel) |
// 2 = n <= 32, rows stored back to back, NA_E = I or not (mfma_pair_iof_kernel)
static int pair_io_mode(const RayenPack* p, const PairImage* img, const float* v, int64_t B, int64_t ldv,
                        const float* y, int64_t ldy) {
  if (img == nullptr || img->nkk < 1 || img->nkk > 2) return 0;
  // batches that cannot give every resident wave a 64-row group are launch-latency work: rayen_mfma_pair.hip runs them with 32
  // rows per wave (nothing to trickle in a single short round)
  if ((B + 63) / 64 < (int64_t)img->n_simd * kMfmaWavesPerSimd) return 0;
  if ((reinterpret_cast<uintptr_t>(v) & 15) != 0 || (reinterpret_cast<uintptr_t>(y) & 15) != 0) return 0;
  if (img->identity && p->n == img->nkk * 32 && p->k == p->n && (ldv % 4) == 0 && (ldy % 4) == 0 &&
      ldv <= (1 << 22) && ldy <= (1 << 22) && !(img->nkk == 2 && img->aux_rows > IoGeom<2>::AUXR) &&
      img->n_items >= (img->nkk == 2 ? IoGeom<2>::NBLK : IoGeom<1>::NBLK))
    return 1;
  if (img->nkk == 1 && ldv == p->n && ldy == p->k && (img->identity ? p->k == p->n : true)) {
    const int chunks_v = (16 * p->n + 63) / 64, chunks_y = (16 * p->k + 63) / 64;
    // the rows of v(next) are read in front of the first NA_E tile (two tiles behind the last fetch at the least),
    // the last chunk of y(prev) has left before that tile writes
    const int f = img->identity ? img->n_items : img->first_out;
    if (chunks_y > f || chunks_v + (img->identity ? 0 : 2) > f) return 0;
    // sets of a few tiles (config 2: five) carry two I/O operations per tile for little walk to hide them under:
    // measured slower than the boundary bursts (config 2, B = 262144: 19.8 against 18.1 us)
    if (img->n_items < 12) return 0;
    if (pair_iof_lds_bytes(p) > 160 * 1024) return 0;
    return 2;
  }
  return 0;
}

bool mfma_pair_io_serves(const RayenPack* p, const PairImage* img, const float* v, int64_t B, int64_t ldv,
                         const float* y, int64_t ldy) {
  return pair_io_mode(p, img, v, B, ldv, y, ldy) != 0;
}

template <int NKK>
static int launch_pair_io(const RayenPack* p, const PairImage* img, const float* v, int64_t B, int64_t ldv,
                          float* y, int64_t ldy, float* kappa, int32_t* active, int32_t* nan_flag,
                          hipStream_t stream) {
  constexpr int per_wave = 64;
  const int64_t n_groups = (B + per_wave - 1) / per_wave;
  const int64_t slots = (int64_t)launch_simds(img->n_simd) * kMfmaWavesPerSimd;
  const int64_t rounds = (n_groups + slots - 1) / slots;
  const int64_t waves = (n_groups + rounds - 1) / rounds;
  const int64_t grid = (waves + kMfmaWaves - 1) / kMfmaWaves;
  auto go = [&](auto kern) {
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kMfmaWaves * 64), 0, stream,
                       static_cast<const f16x8*>(img->Wh), img->items, img->n_items, img->packs, img->y0,
                       p->k, p->n, v, B, ldv, y, ldy, kappa, active, nan_flag, img->w_scale, img->w_inv);
  };
  if (active != nullptr) go(mfma_pair_io_kernel<NKK, true>);
  else go(mfma_pair_io_kernel<NKK, false>);
  return hipGetLastError() == hipSuccess ? RAYEN_OK : RAYEN_E_LAUNCH;
}

template <bool TRACK, bool STAGED>
static int launch_pair_iof_one(const RayenPack* p, const PairImage* img, const float* v, int64_t B, float* y,
                               float* kappa, int32_t* active, int32_t* nan_flag, unsigned grid, hipStream_t stream) {
  // (more than 64 KiB of dynamic LDS has to be asked for, once per kernel and device context; the attribute is set
  // at pack creation -- mfma_pair_io_prepare -- never on a launch path that may be under stream capture)
  const int lds = pair_iof_lds_bytes(p);
  hipLaunchKernelGGL((mfma_pair_iof_kernel<TRACK, STAGED>), dim3(grid), dim3(kMfmaWaves * 64), lds, stream,
                     static_cast<const f16x8*>(img->Wh), img->items, img->n_items, img->packs, img->y0, p->k, p->n, v, B,
                     y, kappa, active, nan_flag, img->w_scale, img->w_inv, pair_iof_buf_bytes(p),
                     img->identity ? img->n_items : img->first_out);
  return hipGetLastError() == hipSuccess ? RAYEN_OK : RAYEN_E_LAUNCH;
}

// called by rayen_pack_create (the only place that may touch function attributes)
int mfma_pair_io_prepare(const RayenPack* p, const PairImage* img) {
  if (img == nullptr || img->nkk != 1) return RAYEN_OK;
  const int lds = pair_iof_lds_bytes(p);
  if (lds > 160 * 1024) return RAYEN_OK;   // (such packs are never served by the flat kernel)
  // The attribute belongs to the kernel instance, not to the pack: a later, smaller pack must not lower what an
  // earlier one was promised (c5 followed by an n = 20 set in one process).  Every pack asks for the running maximum.
  static std::mutex mu;
  static int promised = 0;
  std::lock_guard<std::mutex> hold(mu);
  promised = std::max(promised, lds);
  const int ask = promised;
  bool ok = true;
  auto want = [&](auto kern) {
    ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, ask) == hipSuccess;
  };
  if (img->identity) { want(mfma_pair_iof_kernel<false, false>); want(mfma_pair_iof_kernel<true, false>); }
  else { want(mfma_pair_iof_kernel<false, true>); want(mfma_pair_iof_kernel<true, true>); }
  return ok ? RAYEN_OK : RAYEN_E_LAUNCH;
}

int mfma_pair_io_forward(const RayenPack* p, const PairImage* img, const float* v, int64_t B, int64_t ldv,
                         float* y, int64_t ldy, float* kappa, int32_t* active, int32_t* nan_flag,
                         hipStream_t stream) {
  if (B == 0) return RAYEN_OK;
  const int mode = pair_io_mode(p, img, v, B, ldv, y, ldy);
  if (mode == 0) return RAYEN_E_UNSUPPORTED;
  if (mode == 2) {
    constexpr int per_wave = 64;
    const int64_t n_groups = (B + per_wave - 1) / per_wave;
    const int64_t slots = (int64_t)launch_simds(img->n_simd) * kMfmaWavesPerSimd;
    const int64_t rounds = (n_groups + slots - 1) / slots;
    const int64_t waves = (n_groups + rounds - 1) / rounds;
    const unsigned grid = (unsigned)((waves + kMfmaWaves - 1) / kMfmaWaves);
    if (img->identity)
      return active != nullptr ? launch_pair_iof_one<true, false>(p, img, v, B, y, kappa, active, nan_flag, grid, stream)
                               : launch_pair_iof_one<false, false>(p, img, v, B, y, kappa, active, nan_flag, grid, stream);
    return active != nullptr ? launch_pair_iof_one<true, true>(p, img, v, B, y, kappa, active, nan_flag, grid, stream)
                             : launch_pair_iof_one<false, true>(p, img, v, B, y, kappa, active, nan_flag, grid, stream);
  }
  if (img->nkk == 1) return launch_pair_io<1>(p, img, v, B, ldv, y, ldy, kappa, active, nan_flag, stream);
  return launch_pair_io<2>(p, img, v, B, ldv, y, ldy, kappa, active, nan_flag, stream);
}

}  // namespace rayen
