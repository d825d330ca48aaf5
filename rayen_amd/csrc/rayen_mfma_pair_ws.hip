// Paired-half forward, W-STATIONARY: the tiles of W live in the register files of a workgroup's four waves and never
// move again; what streams is the batch (rayen/constraint_module.py:468-474, 351-458 in one launch; the arithmetic is
// rayen_mfma_pair.hip's, bit for bit).
//
// rayen_mfma_pair.hip / rayen_mfma_pair_io.hip give every wave its own 64 samples and make it walk ALL tiles of W, so
// each wave pulls the whole image (config 3: 136 KB) through the vector-memory path once per 64 samples -- half of a
// CU's L1 bandwidth and a third of a walk's time -- and at two waves per SIMD nothing but the other
// wave's MFMAs hides a wave's epilogues.  Here the roles are swapped:
//   * one workgroup = four waves = ONE wave per SIMD with the whole 512-entry register file.  Wave w keeps ITS tiles
//     of W (a quarter of the item list, whole segments) as MFMA A operands in ACCUMULATOR registers (a[0:159] at config
//     3) for the life of the kernel: the MFMAs are asm statements with an "a" A operand and VGPR accumulators, so the
//     epilogues read their inputs without v_accvgpr_read and the 256 architectural VGPRs stay free for everything else
//     (with builtins hipcc keeps A and B in VGPRs, the accumulators in AGPRs, and spills ~800 registers);
//   * a group of 64 samples is split ONCE into scaled f16 pairs by the workgroup (each wave 16 rows, each lane a
//     quarter row) and published as a B-operand image in LDS; every wave reads the image into registers (16
//     ds_read_b128) and runs its own tiles on it: no A stream at all, and the split costs a quarter per wave;
//   * every wave's candidates of kappa meet in LDS (one float per wave and sample); the rows of y are rebuilt from
//     the image (22 bits, as rayen_mfma_pair.hip does from its B registers), scaled and stored by the lanes that
//     loaded them.
// One workgroup barrier per group, behind the second tile; the image has four generations in LDS and the aux patch
// and the candidates two, so that nothing else needs ordering.  Iteration g of a workgroup, stage = one tile's MFMAs:
//   stage 0: rows(g+1) -> image(g+1)
//   stage 1: epilogue of tile 0 (the aux tile's results -> LDS)                                   | BARRIER
//   stage 2: epilogue of tile 1; y(g-1) rebuilt, scaled, stored; rows(g+2) requested
//   stage s: epilogue of tile s-1;   last stage: image(g+1) -> B registers as the MFMAs release them
//   behind the last stage: epilogue of the last tile (not overlapped: a third accumulator set would cost 32 registers
//   this kernel does not have), the wave's candidates of group g -> LDS
// Everything but the MFMAs is FILLER: at one wave per SIMD a 32-cycle MFMA hides about five single-issue
// instructions of its own wave (MI355X_MICROARCH.md), so every stage is written as 12 NKK slots of `MFMA, a chunk of
// filler` with sched_barrier(0) on both sides of each chunk (hipcc knows nothing of an asm statement's latency and would
// put the chunks wherever its register pressure heuristics like).  Epilogue chunks start behind the stage's third
// MFMA: hipcc pads no hazard of an asm MFMA, and an accumulator is read by the VALU only 64+ cycles after the MFMA
// that wrote it was issued (8 passes + write-back = 11 wait states of 4 cycles are required).
//
// Served: NA_E = I, n = k = 32 NKK, 16-byte aligned rows, one aux tile, the tiles dealt out to four waves fit the
// compiled instances (TPW tiles per wave).
#include "rayen_split_image.h"

#include <algorithm>
#include <type_traits>
#include <utility>

namespace rayen {

namespace {

constexpr int kWsWaves = 4;
constexpr int kWsGen = 4;     // generations of the B-operand image in LDS

struct WsItem {    // what an epilogue needs of an MItem (32 bytes: two s_load_dwordx4)
  int32_t type, flags, seg, row0;
  int32_t aux_order;   // aux row | position in the pack's item list << 8 (ties between waves go to the earlier item)
  float seg_inv, f0, f1;
  __host__ __device__ int aux() const { return aux_order & 255; }
  __host__ __device__ int order() const { return aux_order >> 8; }
};

template <int V> using ic = std::integral_constant<int, V>;

template <int... I, typename F>
__device__ __forceinline__ void static_for_seq(std::integer_sequence<int, I...>, F&& f) {
  (f(ic<I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_seq(std::make_integer_sequence<int, (N > 0 ? N : 0)>{}, static_cast<F&&>(f));
}

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float dpp_xor1(float m) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_xor2(float m) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x4E, 0xF, 0xF, true));
}

// What of this kernel hipcc does NOT see.  Three kinds of registers are named literally in the asm statements and kept
// out of hipcc's hands by amdgpu_num_vgpr(184) (which caps its VGPRs AND its accumulator registers at 184):
//   v[192:255]  the accumulators of the tile walk: accumulator R = 2 (tile & 1) + sample tile = v[192 + 16 R .. +15]
//   v[184:189]  the running values of the epilogues: kappa candidate of sample tile 0 / 1 (v184, v185), the two lanes of
//               a segment's sum of squares for sample tile 0 / 1 (v186, v187 | v188, v189)
//   a[192:255]  the A operands of a wave's first two tiles (chunk c = 2 K-step + piece of tile tt at a[192 + 32 tt + 4 c ..]),
//               which leaves hipcc 88 accumulator registers of overflow for its VGPRs instead of 24 (five tiles per wave)
// Why: (1) whatever hipcc knows as a value it may copy.  With the accumulators as "+v" operands (even pinned to their
// registers by "{v[192:207]}" constraints) it moved an accumulator out of its tuple right behind the MFMA that was
// writing it -- sixteen v_mov in front of a branch of the filler -- and hipcc pads no hazard of an asm MFMA: the copies
// read stale registers (2 900 of 32 768 rows wrong on the first run).  (2) Issue slots are this kernel's budget: at one
// wave per SIMD an instruction costs 4-5 cycles of issue whatever it is, a 32-cycle MFMA hides five of them, and hipcc
// puts an s_nop behind every asm statement that defines a VGPR (output operand or clobber) when another asm statement
// follows.  Statements that only READ hipcc's registers and write named ones carry no pad: the MFMAs, the register chunks
// of the epilogues (v_max3 / v_fma / ds_write on registers by number) and the slot boundaries (empty statements that take
// a chunk's results as INPUTS, so that its arithmetic cannot sink out of its slot).
// Volatile statements keep their order, and the order in this file puts every read of an accumulator 64+ cycles
// behind the MFMA that wrote it (header); scripts/check_ws_asm.py audits the ISA for all of this.
// developer ablation builds (scripts/ubench/tu_variant.sh rayen_mfma_pair_ws <name> -DRAYEN_WS_ABL=<bits>; WRONG RESULTS):
// 1 no row loads | 2 no row stores | 4 no MFMAs | 8 no epilogue chunks | 16 no barrier in the loop
// 256: s_memtime stamps of every wave of workgroup 0 at the stage boundaries of its iterations 2 and 3 go to kappa_out
// (as 32-bit integers, [wave][iteration - 2][stage boundary]; scripts/ubench/ws_stamps.py reads them)
#ifndef RAYEN_WS_ABL
#define RAYEN_WS_ABL 0
#endif
// workgroup b starts (b & 7) * RAYEN_WS_STAGGER * 64 cycles late: the workgroups run the same schedule from the same
// start, so that the row requests and row stores of all 256 CUs would reach the memory system in the same few hundred
// cycles of every iteration (4 MB bursts each way, the vector-memory queues full, the issuing waves stalled)
#ifndef RAYEN_WS_STAGGER
#define RAYEN_WS_STAGGER 16
#endif
constexpr int kWsAcc0 = 192;
constexpr int kWsNumVgpr = 184;
constexpr int kWsKap = 184, kWsS0 = 186, kWsS1 = 188;     // + sample tile
constexpr int kWsNamedTiles = 2;

// D = A B + C on v_mfma_f32_32x32x16_f16: A from the accumulator file (an "a" operand of hipcc's, AREG < 0, or a named
// register: the builtin would take A from a VGPR and put D into the accumulator file), B a VGPR operand of hipcc's,
// C = D = accumulator R.  The chain's first MFMA (C = 0) carries two wait states in front: its B operand may have been
// moved by a v_mov the compiler placed right before the statement (VALU write -> MFMA read).  "memory": the LDS and
// global accesses of the filler stay on their side of the MFMA.
template <int R, bool FIRST, int AREG>
__device__ __forceinline__ void ws_mfma(const f16x8& a, const f16x8& b) {
  static_assert(R >= 0 && R < 4, "two accumulator sets of two sample tiles");
  constexpr int lo = kWsAcc0 + 16 * R, hi = lo + 15;
  if constexpr ((RAYEN_WS_ABL & 4) != 0) return;
  if constexpr (AREG >= 0) {
    if constexpr (FIRST) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 v[%c1:%c2], a[%c3:%c4], %0, 0" : : "v"(b), "i"(lo), "i"(hi), "i"(AREG), "i"(AREG + 3) : "memory");
    else asm volatile("v_mfma_f32_32x32x16_f16 v[%c1:%c2], a[%c3:%c4], %0, v[%c1:%c2]" : : "v"(b), "i"(lo), "i"(hi), "i"(AREG), "i"(AREG + 3) : "memory");
  } else {
    if constexpr (FIRST) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 v[%c2:%c3], %0, %1, 0" : : "a"(a), "v"(b), "i"(lo), "i"(hi) : "memory");
    else asm volatile("v_mfma_f32_32x32x16_f16 v[%c2:%c3], %0, %1, v[%c2:%c3]" : : "a"(a), "v"(b), "i"(lo), "i"(hi) : "memory");
  }
}
template <int REG, int OFF>
__device__ __forceinline__ void ws_load_named(const char* sb, const int voff) {
  uint64_t asm_base;
  asm volatile(RAYEN_ASM_BASE_COPY "global_load_dwordx4 a[%c[r0]:%c[r1]], %[off], " RAYEN_ASM_BASE " offset:%c[o]"
               : [b] "=&s"(asm_base) : [off] "v"(voff), [base] "s"(sb), [r0] "i"(REG), [r1] "i"(REG + 3), [o] "i"(OFF) : "a255", "v255", "memory");
}
template <int OFF>
__device__ __forceinline__ void ws_load_chunk(f16x8& dst, const char* sb, const int voff) {
  uint64_t asm_base;
  asm volatile(RAYEN_ASM_BASE_COPY "global_load_dwordx4 %[d], %[off], " RAYEN_ASM_BASE " offset:%c[o]"
               : [d] "=a"(dst), [b] "=&s"(asm_base) : [off] "v"(voff), [base] "s"(sb), [o] "i"(OFF));
}

// one LDS-DMA: every lane fetches the 16 bytes at base + off; lane L lands at LDS byte lds + 16 L (M0 = the LDS base:
// written in the statement that reads it, restored behind it).  No VGPR destination, no instruction at arrival: the
// issuing wave counts it in vmcnt.
__device__ __forceinline__ void ws_dma16(const char* base, const unsigned off, const unsigned lds) {
  unsigned keep;
  uint64_t asm_base;
  asm volatile(RAYEN_ASM_BASE_COPY "s_mov_b32 %[k], m0\n\ts_mov_b32 m0, %[lds]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[off], " RAYEN_ASM_BASE "\n\ts_mov_b32 m0, %[k]"
               : [k] "=&s"(keep), [b] "=&s"(asm_base) : [off] "v"(off), [base] "s"(base), [lds] "s"(lds) : "memory");
}
__device__ __forceinline__ const char* ws_uniform(const void* p) {
  const uint64_t x = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)x), hi = __builtin_amdgcn_readfirstlane((uint32_t)(x >> 32));
  return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
}

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// x of this lane and of lane ^ 32, as (value of the lower half-wave's lane, value of the upper one's) in every lane:
// v_permlane32_swap exchanges the upper half of its first operand with the lower half of its second (gfx950; measured in
// scripts/ubench/permlane32.hip -- through the builtin this hipcc gets it wrong) -- one VALU instruction
// where xhalf() is a round trip through the LDS crossbar, which a kernel at one wave per SIMD has nothing to hide under.
// (lo op hi) equals (x op xhalf(x)) bit for bit for a commutative op.
__device__ __forceinline__ void ws_halves(const float x, float& lo, float& hi) {
  lo = x;
  hi = x;
  // (two wait states: the copies above are VALU writes right in front, and hipcc pads nothing inside a statement)
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
}

}  // namespace

template <int NKK, int TPW, bool TRACK>
__global__ __launch_bounds__(kWsWaves * 64) __attribute__((amdgpu_num_vgpr(kWsNumVgpr))) void mfma_pair_ws_kernel(
    const f16x8* __restrict__ Wh, const WsItem* __restrict__ witems, const int32_t* __restrict__ wtile,
    const MPack* __restrict__ packs, const float* __restrict__ y0,
    const float* __restrict__ v, int64_t B, int64_t ldv, float* __restrict__ y, int64_t ldy,
    float* __restrict__ kappa_out, int32_t* __restrict__ active_out, int32_t* __restrict__ nan_flag,
    const float w_scale, const float w_inv) {
  static_assert(NKK == 2 && !TRACK, "built for n = 64 without the arg-max record so far");
  static_assert(TPW >= 3, "the schedule needs three stages");
  constexpr int NT = 2, NS = NKK * 2;
  constexpr int NSLOT = 6 * NS;         // MFMAs of one tile on the group's two sample tiles
  __shared__ f16x8 bimg[kWsGen][NT][NS][2][64];   // [generation][sample tile][K-step][piece][lane]: B operands
  // rows on their way in (LDS-DMA, whole lines) and out (staged, whole lines): [parity][row][16 slots of 16 bytes];
  // slot s of row r holds piece s ^ (r & 3) -- the quarter-row accesses of a lane group then hit 16 distinct bank groups
  __shared__ __attribute__((aligned(1024))) float rows_lds[2][64][64];
  __shared__ float aux_lds[2][NT][32][32];        // [parity][sample tile][aux row][sample]
  __shared__ float kap_lds[2][kWsWaves][64];      // [parity][wave][row]: the waves' candidates (scaled domain)
  __shared__ float sc_lds[kWsGen][2][64];         // [generation][sv | 1 / sv][row]
  __shared__ __attribute__((aligned(16))) float y0_lds[NKK * 32];
  (void)packs; (void)active_out;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int col = lane & 31;
  const int hi = lane >> 5;
  const int64_t n_groups = (B + 63) / 64;

  // ---- this wave's tiles of W -> accumulator registers, once.  A[tt][sp][0 | 1] = leading | second piece; the first
  // kWsNamedTiles tiles go to a[192..] by name
  constexpr int NCT = TPW - kWsNamedTiles;         // tiles whose operands hipcc allocates
  f16x8 A[NCT > 0 ? NCT : 1][NS][2];
  WsItem its[TPW];
  static_for<TPW>([&](auto TT) {
    constexpr int tt = decltype(TT)::value;
    its[tt] = witems[wave * TPW + tt];
    int tile = wtile[wave * TPW + tt];
    tile = tile < 0 ? 0 : tile;     // (a tile without rows: padding of this wave's list -- its results are never read)
    const char* tb = reinterpret_cast<const char*>(Wh) + (size_t)tile * (NS * 2048);
    static_for<NS * 2>([&](auto C) {
      constexpr int c = decltype(C)::value;        // chunk 2 sp + piece: 1 KiB of the tile
      const char* sb = tb + (c >> 1) * 2048;
      if constexpr (tt < kWsNamedTiles) ws_load_named<192 + 32 * tt + 4 * c, (c & 1) * 1024>(sb, lane * 16);
      else ws_load_chunk<(c & 1) * 1024>(A[tt - kWsNamedTiles][c >> 1][c & 1], sb, lane * 16);
    });
  });
  for (int i = threadIdx.x; i < NKK * 32; i += kWsWaves * 64) y0_lds[i] = y0[i];

  // ---- a lane's two roles in moving rows.
  // QUARTER role (arithmetic): row qr = 16 wave + lane / 4 of the group, pieces 4 q + i (q = lane & 3, i = 0..3), piece p =
  // columns 4p .. 4p+3 = K-step p >> 2 = q, B-operand elements 4 (i >> 1) .. +3 of lane (col, i & 1): pieces {0, 2} make
  // lane (col, 0)'s 16-byte operand, pieces {1, 3} lane (col, 1)'s.
  // LINE role (global memory): instruction j moves rows 16 wave + 4 j .. + 3 as one kilobyte, lane L = row 4 j + L / 16,
  // slot L & 15 -- whole 128-byte lines both ways.
  const int qr = 16 * wave + (lane >> 2), qq = lane & 3, qx = (lane >> 2) & 3;
  const int lr = lane >> 4, lpiece = (lane & 15) ^ (lane >> 4);      // (row 16 wave + 4 j + lr has (row & 3) = lr)
  const unsigned rows_addr = (unsigned)reinterpret_cast<uintptr_t>(&rows_lds[0][0][0]);
  const unsigned line_v = (unsigned)(lr * ldv * 4 + lpiece * 16), line_y = (unsigned)(lr * ldy * 4 + lpiece * 16);   // (ld <= 2^22: host)

  // rows of group `grp` -> rows_lds[buf] (this wave's sixteen): instructions j0 .. j0 + nj - 1.  FULL: every row of the group
  // exists; else rows beyond the batch fetch the batch's last row (their results are never stored)
  auto dma_rows = [&](auto FULL, const int64_t grp, const int buf, const int j0, const int nj) {
    if constexpr ((RAYEN_WS_ABL & 1) != 0) return;
#pragma unroll
    for (int j = j0; j < j0 + nj; ++j) {
      const int64_t s0 = grp * 64 + 16 * wave + 4 * j;
      const unsigned lds = rows_addr + (unsigned)(buf * 16384 + (16 * wave + 4 * j) * 256);
      if constexpr (decltype(FULL)::value) {
        ws_dma16(ws_uniform(v + s0 * ldv), line_v, lds);
      } else {
        int64_t s = s0 + lr;
        s = s < B ? s : B - 1;
        const int64_t rel = (s - s0) * ldv * 4 + lpiece * 16;       // (may be negative: rows behind the batch's last)
        ws_dma16(ws_uniform(reinterpret_cast<const char*>(v + s0 * ldv) - (int64_t(1) << 30)), (unsigned)(rel + (int64_t(1) << 30)), lds);
      }
    }
  };

  // ---- publish: rows(g+1) -> scaled f16 pairs -> image, as sixteen chunks of a few instructions (P0 .. P15).
  // sv = 2^(13 - floor(log2 max|v|)): exponent arithmetic only (rayen_mfma_pair.hip)
  f32x4 raw[4];
  float pub_m = 0.f, pub_sv = 1.f;
  f16x2 pub_h, pub_l;                  // the first two elements of the piece being split
  auto publish_item = [&](auto COUNTED, auto K, const int buf, const int gen) {
    constexpr int k = decltype(K)::value;
    if constexpr (k == 0) {
      // The rows were requested a walk ago.  COUNTED (steady state): the four row stores of the previous iteration's
      // write-out (five with kappa) were issued BEHIND the four requests, and vmcnt retires in order: at most four
      // outstanding = the rows have landed, the stores need not have.  Otherwise: everything.
      if constexpr (decltype(COUNTED)::value) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 4; ++i) raw[i] = *reinterpret_cast<const f32x4*>(&rows_lds[buf][qr][4 * ((4 * qq + i) ^ qx)]);
    } else if constexpr (k == 4 || k == 5) {
      float m = k == 4 ? 0.f : pub_m;    // (fmaxf drops NaNs: a NaN row keeps a finite scale and stays NaN, as in rayen_mfma_pair.hip)
#pragma unroll
      for (int i = 2 * (k - 4); i < 2 * (k - 4) + 2; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) m = fmaxf(m, __builtin_fabsf(raw[i][c]));
      pub_m = m;
      asm volatile("" : : "v"(pub_m));
    } else if constexpr (k == 6) {
      float m = pub_m;
      m = fmaxf(m, dpp_xor1(m));
      m = fmaxf(m, dpp_xor2(m));
      pub_m = m;
      asm volatile("" : : "v"(pub_m));
    } else if constexpr (k == 7) {
      float inv;
      int sv_exp;
      pow2_scale(pub_m, pub_sv, inv, sv_exp);
      sc_lds[gen][0][qr] = pub_sv;       // (the four lanes of a row store the same two words)
      sc_lds[gen][1][qr] = inv;
    } else if constexpr (k >= 8 && k < 16) {
      // pieces in the order 0, 2 (lane (col, 0)'s operand), 1, 3 (lane (col, 1)'s); two elements per chunk
      constexpr int pi = (k - 8) >> 1, i = pi == 0 ? 0 : pi == 1 ? 2 : pi == 2 ? 1 : 3, e = 2 * ((k - 8) & 1);
      f16x2 h, l;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const float xs = raw[i][e + c] * pub_sv;
        const _Float16 p1 = (_Float16)xs;
        const float r1 = xs - (float)p1;
        h[c] = p1;
        l[c] = (_Float16)r1;
      }
      if constexpr (e == 0) {
        pub_h = h; pub_l = l;
        asm volatile("" : : "v"(pub_h), "v"(pub_l));     // (the chunk's arithmetic ends here: it does not sink to the store)
      } else {
        // piece i = elements 4 (i >> 1) .. +3 of lane (col, i & 1)'s operand of K-step q: eight bytes of each image
        f16x4 h4 = {pub_h[0], pub_h[1], h[0], h[1]}, l4 = {pub_l[0], pub_l[1], l[0], l[1]};
        f16x4* dst = reinterpret_cast<f16x4*>(&bimg[gen][qr >> 5][qq][0][(qr & 31) + 32 * (i & 1)]) + (i >> 1);
        dst[0] = h4;
        dst[128] = l4;     // (the second pieces' image: 64 lanes x 16 bytes further)
      }
    }
  };
  constexpr int NP = 16;      // (P1..P3 are empty: the LDS reads of P0 are used three slots later)

  // ---- write-out: y = y0 + v / max(1, kappa) for the rows of the previous group, v rebuilt from the image (22 bits;
  // scaled by sv), staged through rows_lds and stored as whole lines: twenty-eight chunks (W0 .. W27; the four row stores two slots apart)
  float wo_k[kWsWaves], wo_inv = 1.f, wo_scale = 1.f, wo_knat = 0.f, wo_den = 1.f, nan_acc = 0.f;
  f16x8 wo_fh[2], wo_fl[2];
  f32x4 wo_y0[4], wo_o, wo_back[4];
  auto wout_item = [&](auto FULL, auto K, const int64_t grp, const int gen, const int par, const int buf) {
    constexpr int k = decltype(K)::value;
    // operand (col, hh) of K-step q = pieces hh and hh + 2; y0 of piece i
    auto read_frag = [&](auto HH) {
      constexpr int hh = decltype(HH)::value;
      wo_fh[hh] = bimg[gen][qr >> 5][qq][0][(qr & 31) + 32 * hh];
      wo_fl[hh] = bimg[gen][qr >> 5][qq][1][(qr & 31) + 32 * hh];
    };
    auto read_y0 = [&](auto I) { wo_y0[decltype(I)::value] = *reinterpret_cast<const f32x4*>(&y0_lds[4 * (4 * qq + decltype(I)::value)]); };
    auto read_back = [&](auto J) {
      wo_back[decltype(J)::value] = *reinterpret_cast<const f32x4*>(&rows_lds[buf][16 * wave + 4 * decltype(J)::value][4 * lane]);
    };
    auto store_back = [&](auto J) {
      constexpr int j = decltype(J)::value;
      if constexpr ((RAYEN_WS_ABL & 2) != 0) return;
      const int64_t s0 = grp * 64 + 16 * wave + 4 * j;
      char* yb = const_cast<char*>(ws_uniform(y + s0 * ldy));
      if (decltype(FULL)::value || s0 + lr < B) __builtin_nontemporal_store(wo_back[j], reinterpret_cast<f32x4*>(yb + line_y));
    };
    // every LDS read is issued three or more slots (100+ cycles) ahead of its first use: at one wave per SIMD a wait is
    // a hole in the MFMA stream
    if constexpr (k == 0) {
#pragma unroll
      for (int w = 0; w < kWsWaves; ++w) wo_k[w] = kap_lds[par][w][qr];
      wo_inv = sc_lds[gen][1][qr];
    } else if constexpr (k == 1) {
      read_frag(ic<0>{});
      read_y0(ic<0>{});
    } else if constexpr (k == 2) {
      read_frag(ic<1>{});
      read_y0(ic<2>{});
    } else if constexpr (k == 3) {
      read_y0(ic<1>{});
      read_y0(ic<3>{});
    } else if constexpr (k == 4) {
      float kap = wo_k[0];
#pragma unroll
      for (int w = 1; w < kWsWaves; ++w) kap = fmaxf(kap, wo_k[w]);
      wo_knat = (kap * w_inv) * wo_inv;
      wo_den = fmaxf(1.0f, wo_knat);
      asm volatile("" : : "v"(wo_den), "v"(wo_knat));
    } else if constexpr (k == 5) {
      wo_scale = wo_inv * (1.0f / wo_den);
      asm volatile("" : : "v"(wo_scale));
    } else if constexpr (k >= 6 && k < 14) {
      // pieces in the order 0, 2, 1, 3 (as published); two elements per chunk
      constexpr int pi = (k - 6) >> 1, i = pi == 0 ? 0 : pi == 1 ? 2 : pi == 2 ? 1 : 3, e = 2 * ((k - 6) & 1);
      constexpr int hh = i & 1, off = 4 * (i >> 1);
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const float val = (float)wo_fh[hh][off + e + c] + (float)wo_fl[hh][off + e + c];
        wo_o[e + c] = fmaf(val, wo_scale, wo_y0[i][e + c]);
      }
      // NaN anywhere in y raises the flag (rayen/constraint_module.py:531): 0 * NaN accumulates, 0 * finite does not
      // (an infinite y cannot occur without a NaN beside it: y0 + v s with s <= 1 / sv)
      nan_acc = fmaf(wo_o[e], 0.f, fmaf(wo_o[e + 1], 0.f, nan_acc));
      if constexpr (e == 2) *reinterpret_cast<f32x4*>(&rows_lds[buf][qr][4 * ((4 * qq + i) ^ qx)]) = wo_o;
      else asm volatile("" : : "v"(wo_o), "v"(nan_acc));
    } else if constexpr (k == 17) {
      read_back(ic<0>{});
      read_back(ic<1>{});
    } else if constexpr (k == 18) {
      read_back(ic<2>{});
      read_back(ic<3>{});
    } else if constexpr (k == 21) {
      store_back(ic<0>{});
      if (kappa_out != nullptr && !(RAYEN_WS_ABL & 256)) {
        const int64_t s = grp * 64 + qr;
        if (qq == 0 && (decltype(FULL)::value || s < B)) kappa_out[s] = wo_knat;
      }
    } else if constexpr (k == 23) {
      store_back(ic<1>{});
    } else if constexpr (k == 25) {
      store_back(ic<2>{});
    } else if constexpr (k == 27) {
      store_back(ic<3>{});
    }
  };
  constexpr int NW = 28;

  // vb[t][piece][k-step] = 8 f16 = the B operand of one MFMA; element i = column 16 sp + 8 (i >> 2) + 4 hi + (i & 3)
  f16x8 vb[NT][2][NS];

  // ---- epilogues (rayen_mfma_pair.hip's, on this wave's candidates), cut into twelve chunks of a few instructions.
  // The running values live in named registers (kWsKap, kWsS0, kWsS1 + sample tile); `part` (a segment's sum of squares
  // across its tiles) and the closers' inputs are hipcc's.
  float part[NT] = {0.f, 0.f}, ep_oth[NT], ep_a0[NT], ep_a1[NT], ep_vi[NT], ep_vs[NT];
  // (SET = the accumulator set the tile was computed into; acc(t, g) = register kWsAcc0 + 32 SET + 16 t + g)
  auto epilogue = [&](auto KIND, auto C, auto SET, const WsItem& item, const int gen, const int par) {
    constexpr int kind = decltype(KIND)::value, c = decltype(C)::value, abase = kWsAcc0 + 32 * decltype(SET)::value;
    if constexpr ((RAYEN_WS_ABL & 8) != 0) return;
    if constexpr (kind == MI_LIN) {
      if constexpr (c < 8) {
        constexpr int t = c >> 2, r = abase + 16 * t + 4 * (c & 3);
        asm volatile("v_max3_f32 v[%c0], v[%c0], v[%c1], v[%c2]\n\tv_max3_f32 v[%c0], v[%c0], v[%c3], v[%c4]"
                     : : "i"(kWsKap + t), "i"(r), "i"(r + 1), "i"(r + 2), "i"(r + 3));
      }
    } else if constexpr (kind == MI_QFAC || kind == MI_SOC) {
      // a running sum of squares over the segment's tiles (even registers into s0, odd ones into s1: the two lanes of
      // rayen_mfma_pair.hip's packed FMA), closed on its last tile
      if constexpr (c == 0) {
        if (item.flags & MF_LAST) {
          // what the closers need of the aux tile and the samples' scales: read here, used behind chunk 8.
          // (the item's constants are made opaque where they are used: hipcc otherwise hoists what it derives from
          // them -- LDS addresses of the aux rows, 1 / (2 a'), 4 a', w_inv / f_s of EVERY tile of the wave -- out of the
          // loop into ~30 VGPRs this kernel does not have)
          int aux = item.aux();
          asm volatile("" : "+s"(aux));
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            ep_a0[t] = aux_lds[par][t][aux][col];
            if constexpr (kind == MI_SOC) {
              ep_a1[t] = aux_lds[par][t][aux + 1][col];
              ep_vi[t] = sc_lds[gen][1][32 * t + col];
              ep_vs[t] = sc_lds[gen][0][32 * t + col];
            }
          }
        }
      }
      if constexpr (c < 8) {
        constexpr int t = c >> 2, r = abase + 16 * t + 4 * (c & 3);
        if constexpr ((c & 3) == 0) {
          const float start = (item.flags & MF_FIRST) ? 0.f : part[t];
          asm volatile("v_mov_b32 v[%c1], %0\n\tv_mov_b32 v[%c2], 0" : : "v"(start), "i"(kWsS0 + t), "i"(kWsS1 + t));
        }
        asm volatile("v_fma_f32 v[%c0], v[%c2], v[%c2], v[%c0]\n\tv_fma_f32 v[%c1], v[%c3], v[%c3], v[%c1]\n\t"
                     "v_fma_f32 v[%c0], v[%c4], v[%c4], v[%c0]\n\tv_fma_f32 v[%c1], v[%c5], v[%c5], v[%c1]"
                     : : "i"(kWsS0 + t), "i"(kWsS1 + t), "i"(r), "i"(r + 1), "i"(r + 2), "i"(r + 3));
      } else if constexpr (c == 8) {
        asm volatile("v_add_f32 %0, v[%c2], v[%c3]\n\tv_add_f32 %1, v[%c4], v[%c5]"
                     : "=v"(part[0]), "=v"(part[1]) : "i"(kWsS0), "i"(kWsS1), "i"(kWsS0 + 1), "i"(kWsS1 + 1));
        if (item.flags & MF_LAST) {
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            float lo, hi2;
            ws_halves(part[t], lo, hi2);
            ep_oth[t] = lo + hi2;          // = part + xhalf(part), bit for bit
          }
        }
      } else if constexpr (c == 10 || c == 11) {
        // (one sample tile per chunk: a cone's closed form is ~30 instructions)
        constexpr int t = c - 10;
        if (item.flags & MF_LAST) {
          const float total = ep_oth[t];
          float seg_inv = item.seg_inv, f0 = item.f0, f1 = item.f1;
          asm volatile("" : "+s"(seg_inv), "+s"(f0), "+s"(f1));
          float kc;
          if constexpr (kind != MI_SOC) {
            kc = (ep_a0[t] + __builtin_amdgcn_sqrtf(fmaxf(total, 0.f))) * seg_inv;   // (the segment's own power of two undone)
          } else {
            kc = pair_soc_candidate(ep_a0[t], ep_a1[t], total, w_inv * seg_inv, ep_vi[t], f0, f1, ep_vs[t], w_scale);
          }
          // (`if (kc > kappa) kappa = kc` of rayen_mfma_pair.hip: v_max_f32 keeps kappa when kc is NaN, like the comparison)
          asm volatile("v_max_f32 v[%c1], v[%c1], %0" : : "v"(kc), "i"(kWsKap + t));
        }
      }
    } else if constexpr (kind == MI_AUX) {
      if constexpr (c < 8) {
        // register g of sample tile t is aux row (g & 3) + 8 (g >> 2) + 4 hi of sample col
        constexpr int t = c >> 2, g0 = 4 * (c & 3), r = abase + 16 * t + g0;
        const unsigned addr = (unsigned)reinterpret_cast<uintptr_t>(&aux_lds[par][t][4 * hi][col]);
        constexpr int off = 8 * (g0 >> 2) * 32 * 4;
        asm volatile("ds_write_b32 %0, v[%c1] offset:%c5\n\tds_write_b32 %0, v[%c2] offset:%c6\n\t"
                     "ds_write_b32 %0, v[%c3] offset:%c7\n\tds_write_b32 %0, v[%c4] offset:%c8"
                     : : "v"(addr), "i"(r), "i"(r + 1), "i"(r + 2), "i"(r + 3), "i"(off), "i"(off + 128), "i"(off + 256), "i"(off + 384)
                     : "memory");
      }
    }
  };

  // MFMA `slot` of a tile: two passes over the K-steps, by product size (rayen_mfma_pair.hip) -- the same instructions
  // in the same order on the same operands, hence the same bits.  Accumulator set of tile tt: tt & 1.
  auto mfma_slot = [&](auto TT, auto SLOT, auto SET) {
    constexpr int tt = decltype(TT)::value, i = decltype(SLOT)::value, set = decltype(SET)::value;
    // (piece of A, piece of v, K-step, sample tile) of slot i
    constexpr bool pass1 = i < 4 * NS;
    constexpr int sp = pass1 ? (i >> 2) : ((i - 4 * NS) >> 1), t = pass1 ? (i & 1) : ((i - 4 * NS) & 1);
    constexpr int pa = pass1 ? ((i & 3) < 2 ? 1 : 0) : 0, pv = pass1 ? ((i & 3) < 2 ? 0 : 1) : 0;
    constexpr bool first = pass1 && sp == 0 && (i & 3) < 2;
    if constexpr (tt < kWsNamedTiles) ws_mfma<2 * set + t, first, 192 + 32 * tt + 4 * (2 * sp + pa)>(vb[t][pv][sp], vb[t][pv][sp]);
    else ws_mfma<2 * set + t, first, -1>(A[tt - kWsNamedTiles][sp][pa], vb[t][pv][sp]);
  };
  auto reset_kappa = [&]() { asm volatile("v_mov_b32 v[%c0], 0\n\tv_mov_b32 v[%c1], 0" : : "i"(kWsKap), "i"(kWsKap + 1)); };

  // ---- one group: STEADY = a previous group, the next two groups and every row of all of them exist (no branches)
  const int64_t g0 = blockIdx.x, gstride = gridDim.x;
  auto iteration = [&](auto STEADY, const int64_t grp, const int64_t it) {
    constexpr bool steady = decltype(STEADY)::value;
    const int gen = (int)(it & (kWsGen - 1)), par = (int)(it & 1);
    const int gen_next = (gen + 1) & (kWsGen - 1), gen_prev = (gen + kWsGen - 1) & (kWsGen - 1);
    const bool has_prev = steady || it > 0, has_next = steady || grp + gstride < n_groups, has_next2 = steady || grp + 2 * gstride < n_groups;

    auto stamp = [&](const int idx) {
      if constexpr ((RAYEN_WS_ABL & 256) != 0) {
        if (blockIdx.x == 0 && (it == 2 || it == 3) && lane == 0) {
          const uint64_t tnow = __builtin_amdgcn_s_memtime();
          reinterpret_cast<uint32_t*>(kappa_out)[(wave * 2 + (int)(it - 2)) * 16 + idx] = (uint32_t)tnow;
        }
      }
    };
    stamp(0);
    static_for<TPW>([&](auto TT) {
      constexpr int tt = decltype(TT)::value;
      constexpr int set = tt & 1, pset = (tt + 1) & 1;
      const WsItem& prev = its[tt == 0 ? 0 : tt - 1];
      const int ptype = tt == 0 ? (int)MI_NOP : prev.type;      // (stage 0 carries no epilogue)

      // Slot i of this stage = MFMA i of tile tt, then its filler.  Slots [2, 14) carry the twelve chunks of the
      // previous tile's epilogue (behind the stage's third MFMA: see the header) and exist once per epilogue kind -- a
      // wave-uniform branch per STAGE (a branch per slot costs 20-30 scalar and move instructions in hipcc's
      // structurised control flow).  Everything that does not depend on the kind sits in the other slots, OUTSIDE the
      // switch (whatever a switch arm writes meets the other arms' versions behind it), a few instructions per slot:
      //   stage 0, slots 0..13: rows(g+1) -> image(g+1) (P0..P13)
      //   stage 1, slots 14, 17, 20, 23: rows(g+2) requested;  the free slots of the stages >= 2: y(g-1) (W0..W27)
      //   last stage, slots 16..23: image(g+1) -> B registers as the MFMAs release them (second pieces are dead
      //   behind the first pass, a leading piece behind its second-pass MFMAs)
      // (A tile without rows -- padding of a wave's list -- still issues its MFMAs: a second copy of the stage without
      // them costs hipcc ~100 spilled registers, and the waves meet at the barrier of every group anyway.)
      auto free_slot = [&](auto SLOT) {
        constexpr int i = decltype(SLOT)::value;
        if constexpr (tt == 0 && i < NP) { if (has_next) publish_item(STEADY, ic<i>{}, (int)((it + 1) & 1), gen_next); }
        if constexpr (tt == 1 && i >= 14 && (i - 14) % 3 == 0) { if (has_next2) dma_rows(STEADY, grp + 2 * gstride, par, (i - 14) / 3, 1); }
        if constexpr (tt >= 2 && (i < 2 || i >= 14)) {
          // free slots of the stages >= 2 in order: 0, 1, 14..23 of each
          constexpr int fs = 12 * (tt - 2) + (i < 2 ? i : i - 12);
          if constexpr (fs >= 0 && fs < NW) { if (has_prev) wout_item(STEADY, ic<fs>{}, grp - gstride, gen_prev, par ^ 1, (int)((it + 1) & 1)); }
        }
        if constexpr (tt == TPW - 1) {
          if (has_next) {
            if constexpr (i >= 4 * NS && i < 4 * NS + NS) {
              constexpr int sp = i - 4 * NS;
              vb[0][1][sp] = bimg[gen_next][0][sp][1][lane];
              vb[1][1][sp] = bimg[gen_next][1][sp][1][lane];
            }
            if constexpr (i >= 4 * NS && ((i - 4 * NS) & 1)) {
              constexpr int sp = (i - 4 * NS) >> 1;
              vb[0][0][sp] = bimg[gen_next][0][sp][0][lane];
              vb[1][0][sp] = bimg[gen_next][1][sp][0][lane];
            }
          }
        }
      };
      auto arm = [&](auto KIND) {    // slots [2, 14) with the epilogue of kind KIND
        static_for<12>([&](auto D) {
          constexpr int i = 2 + decltype(D)::value;
          mfma_slot(TT, ic<i>{}, ic<set>{});
          __builtin_amdgcn_sched_barrier(0);
          epilogue(KIND, ic<i - 2>{}, ic<pset>{}, prev, gen, par);
          if constexpr (tt == 0) free_slot(ic<i>{});      // (stage 0 has no epilogue: its switch has one arm)
          __builtin_amdgcn_sched_barrier(0);
        });
      };
      static_for<2>([&](auto SLOT) {
        mfma_slot(TT, SLOT, ic<set>{});
        __builtin_amdgcn_sched_barrier(0);
        free_slot(SLOT);
        __builtin_amdgcn_sched_barrier(0);
      });
      if constexpr (tt == 0) arm(ic<MI_NOP>{});
      else {
        switch (ptype) {
          case MI_LIN: arm(ic<MI_LIN>{}); break;
          case MI_QFAC: arm(ic<MI_QFAC>{}); break;
          case MI_SOC: arm(ic<MI_SOC>{}); break;
          case MI_AUX: arm(ic<MI_AUX>{}); break;
          default: arm(ic<MI_NOP>{}); break;
        }
      }
      static_for<NSLOT - 14>([&](auto D) {
        constexpr int i = 14 + decltype(D)::value;
        mfma_slot(TT, ic<i>{}, ic<set>{});
        __builtin_amdgcn_sched_barrier(0);
        free_slot(ic<i>{});
        __builtin_amdgcn_sched_barrier(0);
      });
      if constexpr (tt == 1) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (the aux tile's LDS stores are asm statements: hipcc does not count them)
        stamp(14);
        if constexpr (!(RAYEN_WS_ABL & 16)) __syncthreads();
        stamp(15);
      }
      stamp(1 + tt);
    });
    // ---- what the free slots could not take (instances with few stages)
    {
      constexpr int NFS = 12 * (TPW - 2);      // free slots of the stages >= 2
      static_for<NW - NFS>([&](auto D) { if (has_prev) wout_item(STEADY, ic<NFS + decltype(D)::value>{}, grp - gstride, gen_prev, par ^ 1, (int)((it + 1) & 1)); });
    }
    // ---- the last tile's epilogue, then this wave's candidates of the group -> LDS
    {
      const WsItem& last = its[TPW - 1];
      auto finish = [&](auto KIND) {
        asm volatile("s_nop 13" ::: "memory");      // (the last MFMA's write-back: hipcc pads nothing behind an asm)
        static_for<12>([&](auto E) { epilogue(KIND, E, ic<(TPW - 1) & 1>{}, last, gen, par); });
      };
      switch (last.type) {
        case MI_LIN: finish(ic<MI_LIN>{}); break;
        case MI_QFAC: finish(ic<MI_QFAC>{}); break;
        case MI_SOC: finish(ic<MI_SOC>{}); break;
        default: break;
      }
      stamp(1 + TPW);
      float k0, k1;
      asm volatile("v_mov_b32 %0, v[%c2]\n\tv_mov_b32 %1, v[%c3]" : "=v"(k0), "=v"(k1) : "i"(kWsKap), "i"(kWsKap + 1));
      float k0l, k0h, k1l, k1h;
      ws_halves(k0, k0l, k0h);
      ws_halves(k1, k1l, k1h);
      kap_lds[par][wave][col] = fmaxf(k0l, k0h);            // (both half-waves store the same word)
      kap_lds[par][wave][32 + col] = fmaxf(k1l, k1h);
      reset_kappa();
      stamp(2 + TPW);
    }
  };

  // ---- the persistent loop over this workgroup's groups
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the tiles of W have landed
  if constexpr (RAYEN_WS_STAGGER > 0) {
    for (int i = 0; i < (int)(blockIdx.x & 7); ++i) __builtin_amdgcn_s_sleep(RAYEN_WS_STAGGER);
  }
  if (g0 < n_groups) {
    dma_rows(std::false_type{}, g0, 0, 0, 4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    static_for<NP>([&](auto K) { publish_item(std::false_type{}, K, 0, 0); });
    if (g0 + gstride < n_groups) dma_rows(std::false_type{}, g0 + gstride, 1, 0, 4);
  }
  __syncthreads();
  if (g0 < n_groups) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int sp = 0; sp < NS; ++sp)
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) vb[t][pc][sp] = bimg[0][t][sp][pc][lane];
  }
  reset_kappa();

  int64_t it = 0;     // iteration = index of the group among this workgroup's
  int64_t grp = g0;
  for (; it < 2 && grp < n_groups; grp += gstride, ++it) iteration(std::false_type{}, grp, it);
  // steady state: groups grp - 2 gstride .. grp + 2 gstride all exist and are whole (the previous iteration wrote rows out)
  for (; (grp + 2 * gstride) * 64 + 64 <= B; grp += gstride, ++it) iteration(std::true_type{}, grp, it);
  for (; grp < n_groups; grp += gstride, ++it) iteration(std::false_type{}, grp, it);
  // ---- the last group's rows
  __syncthreads();
  if (it > 0) {
    const int64_t last = g0 + (it - 1) * gstride;
    static_for<NW>([&](auto K) { wout_item(std::false_type{}, K, last, (int)((it - 1) & (kWsGen - 1)), (int)((it - 1) & 1), (int)(it & 1)); });
  }
  if (nan_flag && (nan_acc != nan_acc)) atomicOr(nan_flag, 1);
}

// ---------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------
struct WsImage {
  WsItem* items = nullptr;     // [4][tpw]
  int32_t* tiles = nullptr;    // [4][tpw]: tile of the f16-pair image, -1 = none
  int tpw = 0;                 // compiled instance that serves the pack
  int64_t bytes = 0;
};

void mfma_pair_ws_free(WsImage* ws) {
  if (ws == nullptr) return;
  if (ws->items) (void)hipFree(ws->items);
  if (ws->tiles) (void)hipFree(ws->tiles);
  delete ws;
}

static int ws_instance_for(int nkk, int tiles) {
  if (nkk == 2) return tiles <= 3 ? 3 : (tiles <= 5 ? 5 : 0);
  return 0;
}

// Distribute the item list over the four waves: whole segments (their running sums live in one wave), the longest
// first onto the lightest wave; the aux tile first in its wave; no wave starts with a tile that closes a segment
// (closers read the aux rows, which are published in front of the barrier that follows the second tile).
int mfma_pair_ws_build(const RayenPack* p, const PairImage* img, WsImage** out) {
  *out = nullptr;
  if (img == nullptr || !img->identity || img->host_items.empty()) return RAYEN_OK;
  if (p->n != img->nkk * 32 || p->k != p->n) return RAYEN_OK;
  const std::vector<MItem>& items = img->host_items;
  struct Unit { int first, count; bool aux; };
  std::vector<Unit> units;
  int n_aux = 0;
  for (int i = 0; i < (int)items.size(); ++i) {
    const MItem& it = items[i];
    if (it.type == MI_AUX) { units.push_back({i, 1, true}); ++n_aux; }
    else if (it.type == MI_LIN) units.push_back({i, 1, false});
    else if (it.type == MI_QFAC || it.type == MI_SOC) {
      if (it.flags & MF_FIRST) units.push_back({i, 1, false});
      else if (!units.empty()) ++units.back().count;
    } else return RAYEN_OK;        // (NA_E tiles, packed low-rank forms: not this kernel's)
  }
  if (n_aux > 1) return RAYEN_OK;
  std::vector<int> order(units.size());
  for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
    if (units[a].aux != units[b].aux) return units[a].aux;       // the aux tile is placed first
    return units[a].count > units[b].count;
  });
  std::vector<std::vector<int>> mine(kWsWaves);
  int load[kWsWaves] = {0, 0, 0, 0};
  for (const int u : order) {
    int w = 0;
    for (int c = 1; c < kWsWaves; ++c) if (load[c] < load[w]) w = c;
    mine[w].push_back(u);
    load[w] += units[u].count;
  }
  std::vector<std::vector<int>> seq(kWsWaves);     // item indices per wave, -1 = empty tile
  int tpw = 0;
  for (int w = 0; w < kWsWaves; ++w) {
    std::vector<int>& us = mine[w];
    std::stable_sort(us.begin(), us.end(), [&](int a, int b) {
      if (units[a].aux != units[b].aux) return units[a].aux;
      return units[a].first < units[b].first;
    });
    auto closes_at_once = [&](int u) {
      const MItem& it = items[units[u].first];
      return it.type == MI_PACK || ((it.type == MI_QFAC || it.type == MI_SOC) && units[u].count == 1);
    };
    if (!us.empty() && closes_at_once(us[0])) {
      size_t alt = 0;
      for (size_t i = 1; i < us.size(); ++i) if (!closes_at_once(us[i])) { alt = i; break; }
      if (alt) std::rotate(us.begin(), us.begin() + alt, us.begin() + alt + 1);
      else seq[w].push_back(-1);
    }
    for (const int u : us)
      for (int c = 0; c < units[u].count; ++c) seq[w].push_back(units[u].first + c);
    tpw = std::max(tpw, (int)seq[w].size());
  }
  const int inst = ws_instance_for(img->nkk, tpw);
  if (inst == 0) return RAYEN_OK;
  std::vector<WsItem> wi((size_t)kWsWaves * inst);
  std::vector<int32_t> wt((size_t)kWsWaves * inst, -1);
  for (int w = 0; w < kWsWaves; ++w)
    for (int t = 0; t < inst; ++t) {
      WsItem& o = wi[(size_t)w * inst + t];
      std::memset(&o, 0, sizeof(o));
      o.type = MI_NOP;
      o.seg_inv = 1.f;
      const int idx = t < (int)seq[w].size() ? seq[w][t] : -1;
      if (idx < 0) continue;
      const MItem& it = items[idx];
      o.type = it.type; o.flags = it.flags; o.seg = it.seg; o.row0 = it.row0; o.aux_order = (it.aux & 255) | (idx << 8);
      o.seg_inv = it.seg_inv; o.f0 = it.f0; o.f1 = it.f1;
      wt[(size_t)w * inst + t] = idx;       // item i of the list is tile i of the image
    }
  WsImage* ws = new WsImage();
  ws->tpw = inst;
  const bool ok = hipMalloc(&ws->items, wi.size() * sizeof(WsItem)) == hipSuccess &&
                  hipMemcpy(ws->items, wi.data(), wi.size() * sizeof(WsItem), hipMemcpyHostToDevice) == hipSuccess &&
                  hipMalloc(&ws->tiles, wt.size() * sizeof(int32_t)) == hipSuccess &&
                  hipMemcpy(ws->tiles, wt.data(), wt.size() * sizeof(int32_t), hipMemcpyHostToDevice) == hipSuccess;
  if (!ok) { mfma_pair_ws_free(ws); return RAYEN_E_ALLOC; }
  ws->bytes = (int64_t)(wi.size() * sizeof(WsItem) + wt.size() * sizeof(int32_t));
  *out = ws;
  return RAYEN_OK;
}

bool mfma_pair_ws_serves(const RayenPack* p, const PairImage* img, const WsImage* ws, const float* v, int64_t B,
                         int64_t ldv, const float* y, int64_t ldy, const int32_t* active) {
  if (ws == nullptr || img == nullptr) return false;
  if (active != nullptr) return false;     // (the arg-max record: the instances with it are not built yet)
  if ((reinterpret_cast<uintptr_t>(v) & 15) != 0 || (reinterpret_cast<uintptr_t>(y) & 15) != 0) return false;
  if ((ldv % 4) != 0 || (ldy % 4) != 0 || ldv < p->n || ldy < p->k) return false;
  // every workgroup (one per CU) gets at least two groups: below that there is nothing to overlap
  return (B + 63) / 64 >= (int64_t)(img->n_simd / 4) * 2;
}

template <int NKK, int TPW>
static int launch_ws(const RayenPack* p, const PairImage* img, const WsImage* ws, const float* v, int64_t B, int64_t ldv,
                     float* y, int64_t ldy, float* kappa, int32_t* active, int32_t* nan_flag, hipStream_t stream) {
  const int64_t n_groups = (B + 63) / 64;
  const int64_t cus = launch_simds(img->n_simd) / 4;
  const int64_t rounds = (n_groups + cus - 1) / cus;
  const unsigned grid = (unsigned)((n_groups + rounds - 1) / rounds);
  auto go = [&](auto kern) {
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kWsWaves * 64), 0, stream, static_cast<const f16x8*>(img->Wh), ws->items,
                       ws->tiles, img->packs, img->y0, v, B, ldv, y, ldy, kappa, active, nan_flag, img->w_scale, img->w_inv);
  };
  if (active != nullptr) return RAYEN_E_UNSUPPORTED;
  go(mfma_pair_ws_kernel<NKK, TPW, false>);
  (void)p;
  return hipGetLastError() == hipSuccess ? RAYEN_OK : RAYEN_E_LAUNCH;
}

int mfma_pair_ws_forward(const RayenPack* p, const PairImage* img, const WsImage* ws, const float* v, int64_t B,
                         int64_t ldv, float* y, int64_t ldy, float* kappa, int32_t* active, int32_t* nan_flag,
                         hipStream_t stream) {
  if (B == 0) return RAYEN_OK;
  if (ws == nullptr) return RAYEN_E_UNSUPPORTED;
  if (img->nkk == 2 && ws->tpw == 3) return launch_ws<2, 3>(p, img, ws, v, B, ldv, y, ldy, kappa, active, nan_flag, stream);
  if (img->nkk == 2 && ws->tpw == 5) return launch_ws<2, 5>(p, img, ws, v, B, ldv, y, ldy, kappa, active, nan_flag, stream);
  return RAYEN_E_UNSUPPORTED;
}

}  // namespace rayen
