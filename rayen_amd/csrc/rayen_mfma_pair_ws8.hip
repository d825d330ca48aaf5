// Paired-half forward, W-STATIONARY on eight waves: the tiles of W live in the registers of a workgroup's eight waves
// (two per SIMD) and never move again; what streams is the batch (rayen/constraint_module.py:468-474, 351-458 in one
// launch; the arithmetic is rayen_mfma_pair.hip's, bit for bit).
//
// rayen_mfma_pair.hip / rayen_mfma_pair_io.hip give every wave its own 64 samples and make it walk ALL tiles of W: each
// wave pulls the whole image (config 3: 136 KB) through the vector-memory path once per 64 samples and splits its own
// 64 rows.  Here
//   * wave w keeps ITS tiles of W (an eighth of the item list, whole segments: at most three tiles = 96 VGPRs at n = 64)
//     as MFMA A operands for the life of the kernel: no A stream at all;
//   * a group of 64 samples is split ONCE into scaled f16 pairs by the workgroup (each wave eight rows) and published
//     as a B-operand image in LDS; every wave reads the image into registers (16 ds_read_b128) and runs its own tiles on it;
//   * every wave's candidates of kappa meet in LDS; the rows of y are rebuilt from the image (22 bits, as
//     rayen_mfma_pair.hip does from its B registers), scaled, staged in LDS and stored as whole lines;
//   * rows come in by LDS-DMA (whole lines, no registers), one group ahead.
// Two waves per SIMD: while one wave is in its epilogues, its row movement or at a barrier, the other one's MFMAs run --
// the overlap rayen_mfma_pair.hip has, here without the A stream and with an eighth of the split per wave.  (The same
// idea at ONE wave per SIMD with the whole register file, accumulators and half the tiles in named registers and every
// filler instruction placed by hand between asm MFMAs, is rayen_mfma_pair_ws.hip: bit-exact too, and slower -- a wave
// alone on its SIMD pays ~170 cycles of issue for each of its eight vector-memory instructions per group and every
// dependent chain of its fillers in full)
// Two workgroup barriers per group: A behind every wave's first tile (the aux tile's rows are in LDS: closers may run),
// B behind the walk (every wave's candidates are in LDS: the rows can be written out).  Even waves walk their first
// tile and then publish the next group's rows, odd waves publish first: the two waves of a SIMD start a group in
// complementary phases.
//
// Served: NA_E = I, n = k = 32 NKK, 16-byte aligned rows, one aux tile, at most TPW tiles per wave.
#include "rayen_split_image.h"

#include <algorithm>
#include <type_traits>
#include <utility>

namespace rayen {

namespace {

// developer ablation builds (scripts/ubench/tu_variant.sh rayen_mfma_pair_ws8 <name> -DRAYEN_W8_ABL=<bits>; WRONG RESULTS):
// 1 no row requests | 2 no write-out | 4 no MFMAs | 8 no epilogues | 16 no publish | 32 no image reads | 64 no barriers
#ifndef RAYEN_W8_ABL
#define RAYEN_W8_ABL 0
#endif
constexpr int kW8Waves = 8;
constexpr int kW8Gen = 4;     // generations of the B-operand image in LDS

struct W8Item {    // what an epilogue needs of an MItem (32 bytes)
  int32_t type, flags, seg, row0;
  int32_t aux_order;   // aux row | position in the pack's item list << 8 (ties between waves go to the earlier item)
  float seg_inv, f0, f1;
  __host__ __device__ int aux() const { return aux_order & 255; }
  __host__ __device__ int order() const { return aux_order >> 8; }
};

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// max over the 8 lanes that share a row (DPP: xor 1, xor 2, half-mirror)
__device__ __forceinline__ float w8_row_max8(float m) {
  m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0xB1, 0xF, 0xF, true)));
  m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x4E, 0xF, 0xF, true)));
  m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x141, 0xF, 0xF, true)));
  return m;
}
// ... over the 4 lanes of a quad
__device__ __forceinline__ float w8_row_max4(float m) {
  m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0xB1, 0xF, 0xF, true)));
  m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x4E, 0xF, 0xF, true)));
  return m;
}

// one LDS-DMA: every lane fetches the 16 bytes at base + off; lane L lands at LDS byte lds + 16 L (M0 = the LDS base:
// written in the statement that reads it, restored behind it).  No VGPR destination: the issuing wave counts it in vmcnt.
__device__ __forceinline__ void w8_dma16(const char* base, const unsigned off, const unsigned lds) {
  unsigned keep;
  uint64_t asm_base;
  asm volatile(RAYEN_ASM_BASE_COPY "s_mov_b32 %[k], m0\n\ts_mov_b32 m0, %[lds]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[off], " RAYEN_ASM_BASE "\n\ts_mov_b32 m0, %[k]"
               : [k] "=&s"(keep), [b] "=&s"(asm_base) : [off] "v"(off), [base] "s"(base), [lds] "s"(lds) : "memory");
}
__device__ __forceinline__ const char* w8_uniform(const void* p) {
  const uint64_t x = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)x), hi = __builtin_amdgcn_readfirstlane((uint32_t)(x >> 32));
  return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
}

}  // namespace

template <int NKK, int TPW, bool TRACK>
__global__ __launch_bounds__(kW8Waves * 64, 2) void mfma_pair_ws8_kernel(
    const f16x8* __restrict__ Wh, const W8Item* __restrict__ witems, const W8Item* __restrict__ witems2,
    const int32_t* __restrict__ wtile, const MPack* __restrict__ packs, const float* __restrict__ y0,
    const float* __restrict__ v, int64_t B, int64_t ldv, float* __restrict__ y, int64_t ldy,
    float* __restrict__ kappa_out, int32_t* __restrict__ active_out, int32_t* __restrict__ nan_flag,
    const float w_scale, const float w_inv) {
  constexpr int NT = 2, NS = NKK * 2;
  constexpr int PIECES = NKK * 8;        // 16-byte pieces of a row
  constexpr int NPL = NKK;               // pieces of its row a lane splits (8 lanes per row)
  constexpr int RPI = 64 / PIECES;       // rows one line-role instruction moves (a kilobyte)
  constexpr int NJ = 8 / RPI;            // line-role instructions for a wave's eight rows
  constexpr int SWM = NKK == 2 ? 3 : 1;  // slot swizzle mask: slot = piece ^ (row & SWM)
  __shared__ f16x8 bimg[kW8Gen][NT][NS][2][64];   // [generation][sample tile][K-step][piece][lane]: B operands
  // rows on their way in (LDS-DMA) and out (staged): [parity][row][pieces]; slot s of row r holds piece s ^ (r & SWM)
  __shared__ __attribute__((aligned(1024))) float rows_lds[2][64][NKK * 32];
  __shared__ __attribute__((aligned(1024))) float stage_lds[64][NKK * 32];     // rows of y on their way out (same slots)
  __shared__ float aux_lds[2][NT][32][32];        // [parity][sample tile][aux row][sample]
  __shared__ float kap_lds[2][kW8Waves][64];      // [parity][wave][row]: the waves' candidates (scaled domain)
  __shared__ int code_lds[2][kW8Waves][TRACK ? 64 : 1];   // the arg-max record of a wave's candidate ...
  __shared__ int key_lds[2][kW8Waves][TRACK ? 64 : 1];    // ... and (half-wave it came from << 16) | item position
  __shared__ float sc_lds[kW8Gen][2][64];         // [generation][sv | 1 / sv][row]
  __shared__ __attribute__((aligned(16))) float y0_lds[NKK * 32];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t n_groups = (B + 63) / 64;
  bool bad = false;

  // ---- this wave's tiles of W, once.  A[tt][sp][0 | 1] = leading | second piece
  // (a tile two triangular factors share -- rayen_tiles.h -- is ONE slot: its K-steps 0,1 belong to its[tt], a first half,
  // its K-steps 2,3 to its2[tt], a second half.  its2[tt].type = MI_NOP: the slot is an ordinary tile.)
  f16x8 A[TPW][NS][2];
  W8Item its[TPW], its2[TPW];
#pragma unroll
  for (int tt = 0; tt < TPW; ++tt) {
    its[tt] = witems[wave * TPW + tt];
    its2[tt] = witems2[wave * TPW + tt];
    int tile = wtile[wave * TPW + tt];
    tile = tile < 0 ? 0 : tile;     // (a tile without rows: padding of this wave's list, never walked)
    const f16x8* tb = Wh + (size_t)tile * (NS * 2 * 64) + lane;
#pragma unroll
    for (int sp = 0; sp < NS; ++sp) {
      A[tt][sp][0] = tb[(sp * 2 + 0) * 64];
      A[tt][sp][1] = tb[(sp * 2 + 1) * 64];
    }
  }
  for (int i = threadIdx.x; i < NKK * 32; i += kW8Waves * 64) y0_lds[i] = y0[i];

  // ---- a lane's two roles in moving this wave's eight rows of a group.
  // ARITHMETIC role: row ar = 8 wave + lane / 8, pieces NPL e .. (e = lane & 7); piece p = columns 4p .. 4p+3 = K-step
  // p >> 2, B-operand elements 4 ((p >> 1) & 1) .. +3 of lane (col, p & 1).
  // LINE role (global memory, whole lines): instruction j moves rows 8 wave + RPI j .. as one kilobyte, lane L = row
  // RPI j + L / PIECES, slot L % PIECES.
  // (Recomputed from an opaque copy of the lane number wherever they are used: as loop invariants hipcc keeps ~20 of the
  // addresses derived from them in VGPRs across the walk, spills them, and every scratch reload is a vmcnt(0) that drains
  // the row requests and row stores in flight -- the eight-wave kernel lost a third of its time to that.)
#define RAYEN_W8_ROLES                                                                       \
  int lane_ = lane;                                                                          \
  asm volatile("" : "+v"(lane_));                                                            \
  const int ar = 8 * wave + (lane_ >> 3), ae = lane_ & 7, ax = ar & SWM;                     \
  const int lr = lane_ / PIECES, lslot = lane_ % PIECES;                                     \
  (void)ar; (void)ae; (void)ax; (void)lr; (void)lslot
  const unsigned rows_addr = (unsigned)reinterpret_cast<uintptr_t>(&rows_lds[0][0][0]);

  // rows of group `grp` -> rows_lds[buf] (this wave's eight).  Rows beyond the batch fetch the batch's last row (their
  // results are never stored).
  auto dma_rows = [&](const int64_t grp, const int buf) {
    if constexpr ((RAYEN_W8_ABL & 1) != 0) return;
    RAYEN_W8_ROLES;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int row = 8 * wave + RPI * j + lr;
      const int64_t s0 = grp * 64 + 8 * wave + RPI * j;
      int64_t s = s0 + lr;
      s = s < B ? s : B - 1;
      const int piece = lslot ^ (row & SWM);
      const int64_t rel = (s - s0) * ldv * 4 + piece * 16;       // (may be negative: rows behind the batch's last)
      w8_dma16(w8_uniform(reinterpret_cast<const char*>(v + s0 * ldv) - (int64_t(1) << 30)), (unsigned)(rel + (int64_t(1) << 30)),
               rows_addr + (unsigned)(buf * (64 * NKK * 128) + (8 * wave + RPI * j) * (NKK * 128)));
    }
  };

  // rows -> scaled f16 pairs -> the image of generation `gen`.  sv = 2^(13 - floor(log2 max|v|)): exponent arithmetic
  // only (rayen_mfma_pair.hip)
  auto publish = [&](const int buf, const int gen) {
    if constexpr ((RAYEN_W8_ABL & 16) != 0) return;
    RAYEN_W8_ROLES;
    f32x4 raw[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) raw[i] = *reinterpret_cast<const f32x4*>(&rows_lds[buf][ar][4 * ((NPL * ae + i) ^ ax)]);
    float m = 0.f;    // (fmaxf drops NaNs: a NaN row keeps a finite scale and stays NaN, as in rayen_mfma_pair.hip)
#pragma unroll
    for (int i = 0; i < NPL; ++i)
#pragma unroll
      for (int c = 0; c < 4; ++c) m = fmaxf(m, __builtin_fabsf(raw[i][c]));
    m = w8_row_max8(m);
    float sv, inv;
    int sv_exp;
    pow2_scale(m, sv, inv, sv_exp);
    sc_lds[gen][0][ar] = sv;       // (the eight lanes of a row store the same two words)
    sc_lds[gen][1][ar] = inv;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const int p = NPL * ae + i;
      f16x4 h, l;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float xs = raw[i][c] * sv;
        const _Float16 p1 = (_Float16)xs;
        const float r1 = xs - (float)p1;
        h[c] = p1;
        l[c] = (_Float16)r1;
      }
      f16x4* dst = reinterpret_cast<f16x4*>(&bimg[gen][ar >> 5][p >> 2][0][(ar & 31) + 32 * (p & 1)]) + ((p >> 1) & 1);
      dst[0] = h;
      dst[128] = l;     // (the second pieces' image: 64 lanes x 16 bytes further)
    }
  };

  // y = y0 + v / max(1, kappa) for this wave's rows of group `grp`: v rebuilt from the image (22 bits; scaled by sv),
  // staged in LDS and stored as whole lines
  auto write_out = [&](const int64_t grp, const int gen, const int par) {
    if constexpr ((RAYEN_W8_ABL & 2) != 0) return;
    RAYEN_W8_ROLES;
    float kap = kap_lds[par][0][ar];
    int code = TRACK ? code_lds[par][0][ar] : 0, key = TRACK ? key_lds[par][0][ar] : 0;
#pragma unroll
    for (int w = 1; w < kW8Waves; ++w) {
      const float o = kap_lds[par][w][ar];
      if (TRACK) {
        // the record rayen_mfma_pair.hip's single walk would keep: lower half-wave first, then the earlier item
        const int oc = code_lds[par][w][ar], ok = key_lds[par][w][ar];
        if (o > kap || (o == kap && oc >= 0 && (code < 0 || ok < key))) { code = oc; key = ok; }
      }
      kap = fmaxf(kap, o);
    }
    const float inv = sc_lds[gen][1][ar];
    const float knat = (kap * w_inv) * inv;
    const float scale = inv * (1.0f / fmaxf(1.0f, knat));
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const int p = NPL * ae + i;
      const f16x4* src = reinterpret_cast<const f16x4*>(&bimg[gen][ar >> 5][p >> 2][0][(ar & 31) + 32 * (p & 1)]) + ((p >> 1) & 1);
      const f16x4 h = src[0], l = src[128];
      const f32x4 y4 = *reinterpret_cast<const f32x4*>(&y0_lds[4 * p]);
      f32x4 o;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float val = (float)h[c] + (float)l[c];
        o[c] = fmaf(val, scale, y4[c]);
        bad |= (o[c] != o[c]);        // (rows beyond the batch repeat its last row)
      }
      *reinterpret_cast<f32x4*>(&stage_lds[ar][4 * (p ^ ax)]) = o;
    }
    {
      const int64_t s = grp * 64 + ar;
      if (ae == 0 && s < B) {
        if (kappa_out) kappa_out[s] = knat;
        if (TRACK) { active_out[2 * s] = code >> 20; active_out[2 * s + 1] = code < 0 ? 0 : (code & 0xFFFFF); }
      }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int row = 8 * wave + RPI * j + lr;
      const int64_t s = grp * 64 + row;
      const f32x4 o = *reinterpret_cast<const f32x4*>(&stage_lds[8 * wave + RPI * j][4 * lane_]);
      if (s < B) __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(y + s * ldy + 4 * (lslot ^ (row & SWM))));
    }
  };

  // vb[t][piece][k-step] = 8 f16 = the B operand of one MFMA; element i = column 16 sp + 8 (i >> 2) + 4 hi + (i & 3)
  f16x8 vb[NT][2][NS];
  auto read_image = [&](const int gen) {
    if constexpr ((RAYEN_W8_ABL & 32) != 0) { if (gen != 0) return; }
    int lane_ = lane;
    asm volatile("" : "+v"(lane_));
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int sp = 0; sp < NS; ++sp)
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) vb[t][pc][sp] = bimg[gen][t][sp][pc][lane_];
  };

  // ---- the wave's running state of a group (scaled domain: gW sv times the natural value)
  float kap[NT], part[NT];
  int acode[NT], akey[NT];   // arg-max record (segment << 20 | row, -1 = none) and the item position it came from
  auto reset_state = [&]() {
#pragma unroll
    for (int t = 0; t < NT; ++t) { kap[t] = 0.f; part[t] = 0.f; acode[t] = -1; akey[t] = 0; }
  };

  // one tile: its MFMAs (two passes over the K-steps, by product size -- rayen_mfma_pair.hip's instructions in its order on
  // its operands, hence its bits), then its epilogue on this wave's candidates
  auto tile_part = [&](auto TT, auto SHAPE, const W8Item& item, const int gen, const int par) {
    constexpr int tt = decltype(TT)::value;
    constexpr int shape = decltype(SHAPE)::value;
    constexpr int sp_lo = shape == MS_HALF_B ? NS / 2 : 0, sp_hi = shape == MS_HALF_A ? NS / 2 : NS;
    constexpr int boff = 0;      // (either half of a shared tile runs against the direction's own K-steps, rayen_tiles.h)
    int lane_ = lane;
    asm volatile("" : "+v"(lane_));
    const int col = lane_ & 31, hi = lane_ >> 5;
    f32x16 acc[NT];
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if constexpr ((RAYEN_W8_ABL & 4) != 0) { acc[0] = zero; acc[1] = zero; acc[0][0] = kap[0]; acc[1][3] = part[1]; }
    else {
#pragma unroll
    for (int sp = sp_lo; sp < sp_hi; ++sp) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[tt][sp][1], vb[t][0][sp + boff], sp == sp_lo ? zero : acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[tt][sp][0], vb[t][1][sp + boff], acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int sp = sp_lo; sp < sp_hi; ++sp)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[tt][sp][0], vb[t][0][sp + boff], acc[t], 0, 0, 0);
    }
    if constexpr ((RAYEN_W8_ABL & 8) != 0) { kap[0] = fmaxf(kap[0], acc[0][5] + acc[1][7]); return; }

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
              akey[t] = item.order();
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
        // (the item's constants are made opaque where they are used: hipcc otherwise hoists what it derives from them --
        // LDS addresses of the aux rows, 1 / (2 a'), 4 a', w_inv / f_s of EVERY tile of the wave -- out of the loop into
        // VGPRs this kernel does not have)
        // (read through readfirstlane: where two call sites of this body share code, hipcc selects between the two items'
        // fields in VGPRs, and an "s" operand does not move them back)
        auto uni = [](const float x) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x))); };
        int aux = __builtin_amdgcn_readfirstlane(item.aux());
        float seg_inv = uni(item.seg_inv), f0 = uni(item.f0), f1 = uni(item.f1);
        asm volatile("" : "+s"(aux), "+s"(seg_inv), "+s"(f0), "+s"(f1));
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const float total = part[t] + xhalf(part[t]);
          const float a0 = aux_lds[par][t][aux][col];
          float kc;
          if (item.type != MI_SOC) {
            kc = (a0 + __builtin_amdgcn_sqrtf(fmaxf(total, 0.f))) * seg_inv;   // (the segment's own power of two undone)
          } else {
            kc = pair_soc_candidate(a0, aux_lds[par][t][aux + 1][col], total, w_inv * seg_inv, sc_lds[gen][1][32 * t + col],
                                    f0, f1, sc_lds[gen][0][32 * t + col], w_scale);
          }
          if (kc > kap[t]) { kap[t] = kc; acode[t] = item.seg << 20; akey[t] = item.order(); }
          // (one sample tile after the other: interleaved, the two closed forms need ~30 more registers than this kernel
          // has left beside its tiles of W and its B operands, and a scratch reload is a vmcnt(0) -- which drains the row
          // requests in flight)
          asm volatile("" : "+v"(kap[0]), "+v"(kap[1]));
        }
      }
    } else if (item.type == MI_AUX) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 16; ++g) aux_lds[par][t][(g & 3) + 8 * (g >> 2) + 4 * hi][col] = acc[t][g];
    } else if (item.type == MI_PACK) {
      const MPack pk = packs[item.aux()];
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
          const float kc = (aux_lds[par][t][slot & 31][col] + __builtin_amdgcn_sqrtf(qs)) * (hi ? pk.inv[a][1] : pk.inv[a][0]);
          if (sid >= 0 && kc > kap[t]) { kap[t] = kc; acode[t] = sid << 20; akey[t] = item.order(); }
        }
      }
    }
  };
  auto tile = [&](const int tt_dyn, auto TT, const int gen, const int par) {
    constexpr int tt = decltype(TT)::value;
    (void)tt_dyn;
    if (its[tt].type == MI_NOP) return;
    if constexpr (NS == 4) {
      if (its2[tt].type != MI_NOP) {     // (wave-uniform: a shared tile, first half then second half)
        tile_part(TT, std::integral_constant<int, MS_HALF_A>{}, its[tt], gen, par);
        tile_part(TT, std::integral_constant<int, MS_HALF_B>{}, its2[tt], gen, par);
        return;
      }
    }
    tile_part(TT, std::integral_constant<int, MS_FULL>{}, its[tt], gen, par);
  };

  // the wave's candidates of a finished group -> LDS
  auto post = [&](const int par) {
    int lane_ = lane;
    asm volatile("" : "+v"(lane_));
    const int col = lane_ & 31, hi = lane_ >> 5;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float other = xhalf(kap[t]);
      if (TRACK) {
        // (rayen_mfma_pair.hip: the lower half-wave's record wins a tie)
        akey[t] |= hi << 16;
        const int ocode = __shfl_xor(acode[t], 32), okey = __shfl_xor(akey[t], 32);
        if (other > kap[t] || (other == kap[t] && hi == 1)) { acode[t] = ocode; akey[t] = okey; }
      }
      const float kk = fmaxf(kap[t], other);
      kap_lds[par][wave][32 * t + col] = kk;          // (both half-waves store the same words)
      if (TRACK) { code_lds[par][wave][32 * t + col] = acode[t]; key_lds[par][wave][32 * t + col] = akey[t]; }
    }
    reset_state();
  };

  // ---- the persistent loop over this workgroup's groups
  const int64_t g0 = blockIdx.x, gstride = gridDim.x;
  if (g0 < n_groups) {
    dma_rows(g0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    publish(0, 0);
    if (g0 + gstride < n_groups) dma_rows(g0 + gstride, 1);
  }
  __syncthreads();
  if (g0 < n_groups) read_image(0);
  reset_state();

  // Two TEAMS half a group apart.  The waves a SIMD is likely to hold are w and w + 4 (a workgroup's waves go to the
  // SIMDs in cyclic order, MI355X_MICROARCH.md); team 0 = waves 0..3 leads, team 1 = waves 4..7 runs one barrier behind:
  //   barrier interval      2g+1              2g+2                       2g+3              2g+4
  //   team 0                H1(g)             H2(g) + rows of y(g-1)     H1(g+1)           H2(g+1) + y(g)
  //   team 1                H2(g-1)           y(g-1) + H1(g)             H2(g)             y(g) + H1(g+1)
  // H1 = image(g) -> B registers, rows(g+1) -> image(g+1), rows(g+2) requested, first tile;  H2 = the other tiles with
  // their closers, the wave's candidates -> LDS.  So whenever one wave of a SIMD moves rows, waits for LDS or sits in its
  // closers, the other one is in a stretch of MFMAs (with every wave in the same phase between the same two barriers
  // the eight-wave kernel took the SUM of its MFMA time and its other work: 4.1 us per group against 3.4 us for
  // rayen_mfma_pair_io.hip).  What each phase needs is in LDS a full interval earlier: the aux tile belongs to a wave of
  // team 0 (its rows of group g: interval 2g+1, first closers 2g+2); image(g+1) is complete after 2g+2, first read in
  // 2g+3; the candidates of group g are complete after 2g+3, the rows of y(g) are written in 2g+4.
#ifndef RAYEN_W8_TEAM
#define RAYEN_W8_TEAM 2
#endif
  const int team = (wave >> RAYEN_W8_TEAM) & 1;     // (developer builds: which waves share a SIMD?)
  int64_t it = 0;     // iteration = index of the group among this workgroup's
  if (team == 1) { if constexpr (!(RAYEN_W8_ABL & 64)) __syncthreads(); }
  for (int64_t grp = g0; grp < n_groups; grp += gstride, ++it) {
    const int gen = (int)(it & (kW8Gen - 1)), par = (int)(it & 1);
    const bool has_next = grp + gstride < n_groups, has_next2 = grp + 2 * gstride < n_groups;
    // ---- H1
    if (team == 1 && it > 0) write_out(grp - gstride, (gen + kW8Gen - 1) & (kW8Gen - 1), par ^ 1);
    if (it > 0) read_image(gen);
    // rows(g+1) -> image(g+1).  They were requested an iteration ago; behind them this wave's row stores of one write-out
    // were issued (NJ of them, more with kappa / the record) -- from the third iteration on -- and vmcnt retires in
    // order.  Then rows(g+2) are requested into the buffer rows(g) came in.
    if (has_next) {
      if (it <= 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" : : "n"(NJ) : "memory");
      __builtin_amdgcn_wave_barrier();
      publish((int)((it + 1) & 1), (gen + 1) & (kW8Gen - 1));
    }
    if (has_next2) dma_rows(grp + 2 * gstride, par);
    tile(0, std::integral_constant<int, 0>{}, gen, par);
    if constexpr (!(RAYEN_W8_ABL & 64)) __syncthreads();
    // ---- H2
    if constexpr (TPW > 1) tile(1, std::integral_constant<int, 1>{}, gen, par);
    if constexpr (TPW > 2) tile(2, std::integral_constant<int, 2>{}, gen, par);
    if constexpr (TPW > 3) tile(3, std::integral_constant<int, 3>{}, gen, par);
    if constexpr (TPW > 4) tile(4, std::integral_constant<int, 4>{}, gen, par);
    if constexpr (TPW > 5) tile(5, std::integral_constant<int, 5>{}, gen, par);
    post(par);
    if (team == 0 && it > 0) write_out(grp - gstride, (gen + kW8Gen - 1) & (kW8Gen - 1), par ^ 1);
    if constexpr (!(RAYEN_W8_ABL & 64)) __syncthreads();
  }
  if (team == 0) { if constexpr (!(RAYEN_W8_ABL & 64)) __syncthreads(); }
  if (it > 0) write_out(g0 + (it - 1) * gstride, (int)((it - 1) & (kW8Gen - 1)), (int)((it - 1) & 1));
  if (nan_flag && bad) atomicOr(nan_flag, 1);
}

// ---------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------
struct Ws8Image {
  W8Item* items = nullptr;     // [8][tpw]
  W8Item* items2 = nullptr;    // [8][tpw]: the second half of a slot that holds a shared tile (type MI_NOP: none)
  int32_t* tiles = nullptr;    // [8][tpw]: tile of the f16-pair image, -1 = none
  int tpw = 0;                 // compiled instance that serves the pack
  int64_t bytes = 0;
};

void mfma_pair_ws8_free(Ws8Image* ws) {
  if (ws == nullptr) return;
  if (ws->items) (void)hipFree(ws->items);
  if (ws->items2) (void)hipFree(ws->items2);
  if (ws->tiles) (void)hipFree(ws->tiles);
  delete ws;
}

static int ws8_instance_for(int nkk, int tiles) {
  if (nkk == 2) return tiles <= 1 ? 1 : tiles <= 2 ? 2 : tiles <= 3 ? 3 : 0;
  if (nkk == 1) return tiles <= 2 ? 2 : tiles <= 4 ? 4 : tiles <= 6 ? 6 : 0;
  return 0;
}

// Deal the item list out to the eight waves: whole segments (their running sums live in one wave), the longest first onto
// the lightest wave; the aux tile first in its wave; no wave starts with a tile that closes a segment (closers read the
// aux rows, which are in LDS behind barrier A = behind every wave's first tile).
int mfma_pair_ws8_build(const RayenPack* p, const PairImage* img, Ws8Image** out) {
  *out = nullptr;
  if (img == nullptr || !img->identity || img->host_items.empty()) return RAYEN_OK;
  if (p->n != img->nkk * 32 || p->k != p->n) return RAYEN_OK;
  const std::vector<MItem>& items = img->host_items;
  // a slot = one tile of the image in a wave's registers: an ordinary item, or the two halves of a shared tile
  // (rayen_tiles.h: item `a` its K-steps 0,1, item `b` its K-steps 2,3); a unit = the slots one wave must own together
  // (a segment's running sum lives in one wave; two segments that share a tile go together: three slots)
  struct Slot { int a, b; };
  struct Unit { std::vector<Slot> slots; bool aux; int first; int count() const { return (int)slots.size(); } };
  std::vector<Unit> units;
  int n_aux = 0;
  for (int i = 0; i < (int)items.size(); ++i) {
    const MItem& it = items[i];
    if (it.type == MI_AUX) { units.push_back({{{i, -1}}, true, i}); ++n_aux; }
    else if (it.type == MI_LIN || it.type == MI_PACK) units.push_back({{{i, -1}}, false, i});
    else if (it.type == MI_QFAC || it.type == MI_SOC) {
      if (it.shape() == MS_HALF_B) {
        if (units.empty() || units.back().slots.back().b >= 0 || items[units.back().slots.back().a].shape() != MS_HALF_A) return RAYEN_OK;
        units.back().slots.back().b = i;
      } else if (it.shape() != MS_HALF_A && (it.flags & MF_FIRST)) {
        units.push_back({{{i, -1}}, false, i});
      } else if (!units.empty()) {
        units.back().slots.push_back({i, -1});
      } else {
        return RAYEN_OK;
      }
    } else return RAYEN_OK;        // (NA_E tiles: not this kernel's)
  }
  if (n_aux > 1) return RAYEN_OK;
  std::vector<int> order(units.size());
  for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
    if (units[a].aux != units[b].aux) return units[a].aux;       // the aux tile is placed first
    return units[a].count() > units[b].count();
  });
  std::vector<std::vector<int>> mine(kW8Waves);
  int load[kW8Waves] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (const int u : order) {
    int w = 0;
    for (int c = 1; c < kW8Waves; ++c) if (load[c] < load[w]) w = c;
    mine[w].push_back(u);
    load[w] += units[u].count();
  }
  // the SIMD of wave w is not ours to choose, but waves s and s + 4 are the likeliest pair: heavy beside light, and the
  // wave with the aux tile in the leading team (waves 0..3: the kernel's closers count on it)
  {
    std::vector<int> by_load(kW8Waves);
    for (int w = 0; w < kW8Waves; ++w) by_load[w] = w;
    std::stable_sort(by_load.begin(), by_load.end(), [&](int a, int b) { return load[a] > load[b]; });
    std::vector<std::vector<int>> paired(kW8Waves);
    for (int s = 0; s < kW8Waves / 2; ++s) {
      std::vector<int> heavy = mine[by_load[s]], light = mine[by_load[kW8Waves - 1 - s]];
      bool light_has_aux = false;
      for (const int u : light) light_has_aux = light_has_aux || units[u].aux;
      if (light_has_aux) std::swap(heavy, light);
      paired[s] = heavy;
      paired[s + kW8Waves / 2] = light;
    }
    mine = paired;
  }
  const Slot empty_slot = {-1, -1};
  std::vector<std::vector<Slot>> seq(kW8Waves);     // slots per wave, a = -1: an empty one
  int tpw = 0;
  for (int w = 0; w < kW8Waves; ++w) {
    std::vector<int>& us = mine[w];
    std::stable_sort(us.begin(), us.end(), [&](int a, int b) {
      if (units[a].aux != units[b].aux) return units[a].aux;
      return units[a].first < units[b].first;
    });
    auto closes_at_once = [&](int u) {
      const MItem& it = items[units[u].first];
      return it.type == MI_PACK || ((it.type == MI_QFAC || it.type == MI_SOC) && units[u].count() == 1);
    };
    if (!us.empty() && closes_at_once(us[0])) {
      size_t alt = 0;
      for (size_t i = 1; i < us.size(); ++i) if (!closes_at_once(us[i])) { alt = i; break; }
      if (alt) std::rotate(us.begin(), us.begin() + alt, us.begin() + alt + 1);
      else seq[w].push_back(empty_slot);
    }
    for (const int u : us)
      for (const Slot& sl : units[u].slots) seq[w].push_back(sl);
    tpw = std::max(tpw, (int)seq[w].size());
  }
  const int inst = ws8_instance_for(img->nkk, tpw);
  if (inst == 0) return RAYEN_OK;
  std::vector<W8Item> wi((size_t)kW8Waves * inst), wi2((size_t)kW8Waves * inst);
  std::vector<int32_t> wt((size_t)kW8Waves * inst, -1);
  auto fill = [&](W8Item& o, int idx) {
    std::memset(&o, 0, sizeof(o));
    o.type = MI_NOP;
    o.seg_inv = 1.f;
    if (idx < 0) return;
    const MItem& it = items[idx];
    o.type = it.type; o.flags = it.flags; o.seg = it.seg; o.row0 = it.row0;
    o.aux_order = (it.aux & 255) | (idx << 8);
    o.seg_inv = it.seg_inv; o.f0 = it.f0; o.f1 = it.f1;
  };
  for (int w = 0; w < kW8Waves; ++w)
    for (int t = 0; t < inst; ++t) {
      const Slot sl = t < (int)seq[w].size() ? seq[w][t] : empty_slot;
      fill(wi[(size_t)w * inst + t], sl.a);
      fill(wi2[(size_t)w * inst + t], sl.b);
      if (sl.a >= 0) wt[(size_t)w * inst + t] = items[sl.a].tile();
    }
  Ws8Image* ws = new Ws8Image();
  ws->tpw = inst;
  const bool ok = hipMalloc(&ws->items, wi.size() * sizeof(W8Item)) == hipSuccess &&
                  hipMemcpy(ws->items, wi.data(), wi.size() * sizeof(W8Item), hipMemcpyHostToDevice) == hipSuccess &&
                  hipMalloc(&ws->items2, wi2.size() * sizeof(W8Item)) == hipSuccess &&
                  hipMemcpy(ws->items2, wi2.data(), wi2.size() * sizeof(W8Item), hipMemcpyHostToDevice) == hipSuccess &&
                  hipMalloc(&ws->tiles, wt.size() * sizeof(int32_t)) == hipSuccess &&
                  hipMemcpy(ws->tiles, wt.data(), wt.size() * sizeof(int32_t), hipMemcpyHostToDevice) == hipSuccess;
  if (!ok) { mfma_pair_ws8_free(ws); return RAYEN_E_ALLOC; }
  ws->bytes = (int64_t)(2 * wi.size() * sizeof(W8Item) + wt.size() * sizeof(int32_t));
  *out = ws;
  return RAYEN_OK;
}

bool mfma_pair_ws8_serves(const RayenPack* p, const PairImage* img, const Ws8Image* ws, const float* v, int64_t B,
                          int64_t ldv, const float* y, int64_t ldy) {
  if (ws == nullptr || img == nullptr) return false;
  if ((reinterpret_cast<uintptr_t>(v) & 15) != 0 || (reinterpret_cast<uintptr_t>(y) & 15) != 0) return false;
  if ((ldv % 4) != 0 || (ldy % 4) != 0 || ldv < p->n || ldy < p->k || ldv > (1 << 22) || ldy > (1 << 22)) return false;
  // every workgroup (one per CU) gets at least two groups: below that there is nothing to overlap
  return (B + 63) / 64 >= (int64_t)(img->n_simd / 4) * 2;
}

template <int NKK, int TPW>
static int launch_ws8(const RayenPack* p, const PairImage* img, const Ws8Image* ws, const float* v, int64_t B, int64_t ldv,
                      float* y, int64_t ldy, float* kappa, int32_t* active, int32_t* nan_flag, hipStream_t stream) {
  const int64_t n_groups = (B + 63) / 64;
  const int64_t cus = launch_simds(img->n_simd) / 4;
  const int64_t rounds = (n_groups + cus - 1) / cus;
  const unsigned grid = (unsigned)((n_groups + rounds - 1) / rounds);
  auto go = [&](auto kern) {
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kW8Waves * 64), 0, stream, static_cast<const f16x8*>(img->Wh), ws->items,
                       ws->items2, ws->tiles, img->packs, img->y0, v, B, ldv, y, ldy, kappa, active, nan_flag, img->w_scale, img->w_inv);
  };
  if (active != nullptr) go(mfma_pair_ws8_kernel<NKK, TPW, true>);
  else go(mfma_pair_ws8_kernel<NKK, TPW, false>);
  (void)p;
  return hipGetLastError() == hipSuccess ? RAYEN_OK : RAYEN_E_LAUNCH;
}

int mfma_pair_ws8_forward(const RayenPack* p, const PairImage* img, const Ws8Image* ws, const float* v, int64_t B,
                          int64_t ldv, float* y, int64_t ldy, float* kappa, int32_t* active, int32_t* nan_flag,
                          hipStream_t stream) {
  if (B == 0) return RAYEN_OK;
  if (ws == nullptr) return RAYEN_E_UNSUPPORTED;
#define RAYEN_W8_CASE(NKK_, TPW_) \
  if (img->nkk == NKK_ && ws->tpw == TPW_) return launch_ws8<NKK_, TPW_>(p, img, ws, v, B, ldv, y, ldy, kappa, active, nan_flag, stream);
  RAYEN_W8_CASE(2, 1) RAYEN_W8_CASE(2, 2) RAYEN_W8_CASE(2, 3)
  RAYEN_W8_CASE(1, 2) RAYEN_W8_CASE(1, 4) RAYEN_W8_CASE(1, 6)
#undef RAYEN_W8_CASE
  return RAYEN_E_UNSUPPORTED;
}

}  // namespace rayen
