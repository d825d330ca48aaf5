// Tile layout shared by the MFMA kernels (private to librayen_hip.so).
//
// The constant set (rows of W, see include/rayen_hip.h) is laid out as a sequence of 32-row tiles,
// each described by one MItem.  How a tile's 32 x n_pad block is stored on the device (fragment
// order of a particular MFMA instruction, element type) is up to the kernel family.
#pragma once

#include <cmath>
#include <cstring>
#include <vector>

#include "rayen_internal.h"

namespace rayen {

enum : int32_t { MI_AUX = 0, MI_LIN = 1, MI_QSYM = 2, MI_QFAC = 3, MI_SOC = 4, MI_OUT = 5, MI_NOP = 6, MI_PACK = 7 };
enum : int32_t { MF_FIRST = 1, MF_LAST = 2, MF_SYM = 4 };  // SYM: rows of a symmetric form, summed as acc . v

// One work item of the tile walk = one 32-row tile of W.
struct MItem {
  int32_t type;
  int32_t flags;
  int32_t seg;     // caller's segment index (reported in `active`)
  int32_t row0;    // LIN: logical W row of the tile's first row | QSYM: tile index | OUT: first output row
  int32_t aux;     // row of phi | c (b is aux+1) inside the aux tile
  int32_t qbegin;  // first k-group this tile needs (QSYM: the block-lower-triangular part is folded away)
  float f0, f1;    // SOC: tau, a'
  float seg_inv;   // f16-pair image: 1 / (the power of two the segment's rows were boosted by), see rayen_mfma_pair.hip
  float pad_;
  double f0d, f1d; // the same in full precision (fp64 kernel)
};

// A packed tile holds up to eight small factor segments (rank <= 4: one quad of rows = the four
// registers 4a..4a+3 of one half-wave; rank 5..8: the same quad in both halves).  One record per
// tile, indexed by MItem::aux; row0 of the item carries the "pair" bits.
struct MPack {
  int32_t aux[4][2];  // [quad a][half]: aux row of phi for the segment sitting there
  int32_t seg[4][2];  // caller's segment index, -1 = empty
  float inv[4][2];    // f16-pair image: 1 / (the power of two that segment's rows were boosted by)
};


inline int n_pad_of(int n) { return (n + 31) / 32 * 32; }

struct TileLayout {
  int n, n_pad;
  std::vector<double> raw;  // [tile][32][n_pad], zero padded
  std::vector<MItem> items;
  std::vector<MPack> packs;
  int64_t useful_rows = 0;
  std::vector<std::vector<double>> owned;  // factor rows computed here (layouts without symmetric forms)

  explicit TileLayout(int n_) : n(n_), n_pad(n_pad_of(n_)) {}
  int nq() const { return n_pad / 8; }
  int n_tiles() const { return (int)(raw.size() / ((size_t)32 * n_pad)); }

  // rows: pointers to up to 32 source rows (nullptr = zero row), each with `ncols` valid columns
  void add_tile(const std::vector<const double*>& rows, int ncols) {
    const size_t base = raw.size();
    raw.resize(base + (size_t)32 * n_pad, 0.0);
    for (size_t r = 0; r < rows.size() && r < 32; ++r) {
      if (rows[r] == nullptr) continue;
      for (int c = 0; c < ncols && c < n_pad; ++c) raw[base + r * n_pad + c] = rows[r][c];
      ++useful_rows;
    }
  }

  // fragment order of v_mfma_f32_32x32x2_f32 with the K order of rayen_mfma.hip:
  // [tile][k-group q][lane l] float4 = row l&31, columns 8q + 4(l>>5) .. +3
  std::vector<float> fragments_f32() const {
    const int nt = n_tiles(), q_n = nq();
    std::vector<float> frag((size_t)nt * q_n * 64 * 4, 0.f);
    for (int t = 0; t < nt; ++t)
      for (int q = 0; q < q_n; ++q)
        for (int l = 0; l < 64; ++l)
          for (int c = 0; c < 4; ++c)
            frag[(((size_t)t * q_n + q) * 64 + l) * 4 + c] =
                (float)raw[((size_t)t * 32 + (l & 31)) * n_pad + 8 * q + 4 * (l >> 5) + c];
    return frag;
  }
};

inline int aux_rows_of(const RayenSegment& g) {
  if (g.type == RAYEN_SEG_QUAD_SYM || g.type == RAYEN_SEG_QUAD_FAC) return 1;
  if (g.type == RAYEN_SEG_SOC) return 2;
  return 0;
}

inline bool is_small_factor(const RayenSegment& g) { return g.type == RAYEN_SEG_QUAD_FAC && g.nrows <= 8; }

// Lay the whole constant set out as a sequence of 32-row tiles:
//   segments are taken in order, in batches whose aux rows (phi | c, b) fit one aux tile; each
//   batch = [AUX tile] [own tiles of the large segments] [packed tiles of the small factor ones];
//   the rows of NA_E (if it is not the identity) come last.
// Factor of a positive semi-definite G (n x n, row-major): rows u_j with sum_j u_j u_j' = G.
// Through the eigen-decomposition (cyclic Jacobi, fp64): u_j = sqrt(lambda_j) q_j' for the eigenvalues above
// 1e-13 lambda_max.  A module built in fp32 hands over forms that carry the rounding noise of its buffers (eigenvalues
// of -1e-8 lambda_max where the exact form is rank deficient): the negative part is dropped, everything else is
// reproduced to fp64 rounding.  (Round 1 used a Cholesky with diagonal pivoting; on such forms it stops at the first
// non-positive pivot with a residual of 1e-6 |G| -- the reason fuzz set 971 was 2e-5 off on the split-operand kernels.)
inline std::vector<std::vector<double>> psd_factor_rows(const double* G, int n) {
  std::vector<double> A((size_t)n * n), Q((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i) {
    Q[(size_t)i * n + i] = 1.0;
    for (int j = 0; j < n; ++j) A[(size_t)i * n + j] = 0.5 * (G[(size_t)i * n + j] + G[(size_t)j * n + i]);
  }
  double scale = 0.0;
  for (const double x : A) scale = std::fabs(x) > scale ? std::fabs(x) : scale;
  std::vector<std::vector<double>> rows;
  if (!(scale > 0.0) || !std::isfinite(scale)) return rows;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0;
    for (int p = 0; p < n; ++p)
      for (int q = p + 1; q < n; ++q) off += A[(size_t)p * n + q] * A[(size_t)p * n + q];
    if (!(std::sqrt(off) > 1e-17 * scale * n)) break;
    for (int p = 0; p < n; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = A[(size_t)p * n + q];
        if (std::fabs(apq) <= 1e-300) continue;
        const double theta = (A[(size_t)q * n + q] - A[(size_t)p * n + p]) / (2.0 * apq);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
        for (int k = 0; k < n; ++k) {   // columns p, q of A
          const double akp = A[(size_t)k * n + p], akq = A[(size_t)k * n + q];
          A[(size_t)k * n + p] = c * akp - sn * akq;
          A[(size_t)k * n + q] = sn * akp + c * akq;
        }
        for (int k = 0; k < n; ++k) {   // rows p, q of A
          const double apk = A[(size_t)p * n + k], aqk = A[(size_t)q * n + k];
          A[(size_t)p * n + k] = c * apk - sn * aqk;
          A[(size_t)q * n + k] = sn * apk + c * aqk;
        }
        for (int k = 0; k < n; ++k) {   // eigenvectors: columns of Q
          const double qkp = Q[(size_t)k * n + p], qkq = Q[(size_t)k * n + q];
          Q[(size_t)k * n + p] = c * qkp - sn * qkq;
          Q[(size_t)k * n + q] = sn * qkp + c * qkq;
        }
      }
  }
  double lmax = 0.0;
  for (int i = 0; i < n; ++i) lmax = A[(size_t)i * n + i] > lmax ? A[(size_t)i * n + i] : lmax;
  // largest eigenvalues first (the order only decides which rows share a tile)
  std::vector<int> order(n);
  for (int i = 0; i < n; ++i) order[i] = i;
  for (int i = 0; i < n; ++i)
    for (int j = i + 1; j < n; ++j)
      if (A[(size_t)order[j] * n + order[j]] > A[(size_t)order[i] * n + order[i]]) { const int t = order[i]; order[i] = order[j]; order[j] = t; }
  for (int idx = 0; idx < n; ++idx) {
    const int i = order[idx];
    const double lam = A[(size_t)i * n + i];
    if (!(lam > 1e-13 * lmax) || !(lam > 0.0)) continue;
    const double root = std::sqrt(lam);
    std::vector<double> u(n);
    for (int c = 0; c < n; ++c) u[c] = root * Q[(size_t)c * n + i];
    rows.push_back(u);
  }
  return rows;
}

// allow_sym = false: every quadratic / cone is laid out through a factor (rows u with ||U v||^2 = v'Gv), so
// that no epilogue needs the direction itself (the split-operand kernel keeps v only as bf16 pieces).
inline int layout_tiles(const RayenPack* p, TileLayout& b, bool allow_pack, bool allow_sym = true) {
  const double* W = p->W.data();
  auto wrow = [&](int r) { return W + (size_t)r * p->n; };
  auto blank = [](int type) { MItem it; std::memset(&it, 0, sizeof(it)); it.type = type; it.seg_inv = 1.f; return it; };
  const size_t nseg = p->segs.size();
  size_t s0 = 0;
  while (s0 < nseg) {
    // ---- batch [s0, s1): as many segments as one aux tile can serve
    size_t s1 = s0;
    int aux_used = 0;
    while (s1 < nseg && aux_used + aux_rows_of(p->segs[s1]) <= 32) aux_used += aux_rows_of(p->segs[s1++]);
    if (s1 == s0) return RAYEN_E_UNSUPPORTED;
    std::vector<int> aux_slot(nseg, -1);
    if (aux_used > 0) {
      std::vector<const double*> rows;
      for (size_t s = s0; s < s1; ++s) {
        const RayenSegment& g = p->segs[s];
        if (aux_rows_of(g) == 0) continue;
        aux_slot[s] = (int)rows.size();
        for (int r = 0; r < aux_rows_of(g); ++r) rows.push_back(wrow(g.aux_row + r));
      }
      b.items.push_back(blank(MI_AUX));
      b.add_tile(rows, p->n);
    }
    // ---- large segments: their own tiles
    for (size_t s = s0; s < s1; ++s) {
      const RayenSegment& g = p->segs[s];
      if (allow_pack && is_small_factor(g)) continue;
      // a symmetric form v'Gv costs (NKK+1)/2 tiles per 32 rows thanks to the block-triangular fold;
      // an SOC block M (rows x n) is turned into G = M'M when that is cheaper than its own rows
      const int sym_cost = (b.n_pad / 32) * (b.n_pad / 32 + 1) / 2;        // both in 32 x 32 blocks of MFMA work
      const int fac_cost = ((g.nrows + 31) / 32) * (b.n_pad / 32);
      const bool sym = allow_sym && (g.type == RAYEN_SEG_QUAD_SYM || (g.type == RAYEN_SEG_SOC && sym_cost < fac_cost));
      const bool refactor = !allow_sym && g.type == RAYEN_SEG_QUAD_SYM;
      size_t fac0 = 0;
      int fac_rows = g.nrows;
      if (refactor) {
        fac0 = b.owned.size();
        for (auto& u : psd_factor_rows(wrow(g.row0), p->n)) b.owned.push_back(std::move(u));
        fac_rows = (int)(b.owned.size() - fac0);
        if (fac_rows == 0) { b.owned.push_back(std::vector<double>(p->n, 0.0)); fac_rows = 1; }  // G = 0
      }
      std::vector<double> gram;  // [n][n] for an SOC in symmetric form
      if (sym && g.type == RAYEN_SEG_SOC) {
        gram.assign((size_t)p->n * p->n, 0.0);
        for (int r = 0; r < g.nrows; ++r)
          for (int i = 0; i < p->n; ++i) {
            const double mi = wrow(g.row0 + r)[i];
            if (mi == 0.0) continue;
            for (int j = 0; j < p->n; ++j) gram[(size_t)i * p->n + j] += mi * wrow(g.row0 + r)[j];
          }
      }
      auto srow = [&](int r) { return gram.empty() ? wrow(g.row0 + r) : gram.data() + (size_t)r * p->n; };
      const int srows = sym ? p->n : fac_rows;
      const int total = sym ? b.n_pad : fac_rows;
      const int ntiles = (total + 31) / 32;
      for (int t = 0; t < ntiles; ++t) {
        std::vector<const double*> rows;
        std::vector<std::vector<double>> folded;  // symmetric form: row tile t keeps column blocks >= t, off-diagonal ones doubled
        if (sym) {
          for (int r = 32 * t; r < 32 * t + 32 && r < srows; ++r) {
            std::vector<double> row(p->n, 0.0);
            for (int c = 32 * t; c < p->n; ++c) row[c] = srow(r)[c] * (c >= 32 * (t + 1) ? 2.0 : 1.0);
            folded.push_back(row);
          }
          for (auto& row : folded) rows.push_back(row.data());
        } else {
          for (int r = 32 * t; r < 32 * t + 32 && r < fac_rows; ++r)
            rows.push_back(refactor ? b.owned[fac0 + r].data() : wrow(g.row0 + r));
        }
        b.add_tile(rows, p->n);
        MItem it = blank(0);
        it.seg = (int32_t)s;
        it.aux = aux_slot[s];
        it.f0 = (float)g.f0;
        it.f1 = (float)g.f1;
        it.flags = (t == 0 ? MF_FIRST : 0) | (t == ntiles - 1 ? MF_LAST : 0) | (sym ? MF_SYM : 0);
        if (sym) { it.row0 = t; it.qbegin = 4 * t; }
        switch (g.type) {
          case RAYEN_SEG_LIN: it.type = MI_LIN; it.row0 = g.row0 + 32 * t; break;
          case RAYEN_SEG_QUAD_SYM: it.type = refactor ? MI_QFAC : MI_QSYM; break;
          case RAYEN_SEG_QUAD_FAC: it.type = MI_QFAC; break;
          case RAYEN_SEG_SOC: it.type = MI_SOC; break;
          default: return RAYEN_E_UNSUPPORTED;
        }
        b.items.push_back(it);
      }
    }
    // ---- small factor segments: eight quads of rows per packed tile
    {
      std::vector<const double*> rows(32, nullptr);
      MPack pk;
      int pair_bits = 0, used = 0;  // `used` counts half-quads handed out: slot = a * 2 + half
      auto reset = [&]() {
        std::fill(rows.begin(), rows.end(), nullptr);
        for (int a = 0; a < 4; ++a) for (int h = 0; h < 2; ++h) { pk.aux[a][h] = 0; pk.seg[a][h] = -1; pk.inv[a][h] = 1.f; }
        pair_bits = 0;
        used = 0;
      };
      auto flush = [&]() {
        if (used == 0) return;
        MItem it = blank(MI_PACK);
        it.aux = (int32_t)b.packs.size();
        it.row0 = pair_bits;
        b.packs.push_back(pk);
        b.items.push_back(it);
        b.add_tile(rows, p->n);
        reset();
      };
      reset();
      for (size_t s = s0; s < s1; ++s) {
        const RayenSegment& g = p->segs[s];
        if (!allow_pack || !is_small_factor(g)) continue;
        const bool pair = g.nrows > 4;
        if (pair && (used & 1)) ++used;          // a pair starts on an even half-quad
        if (used + (pair ? 2 : 1) > 8) flush();
        const int a = used / 2, h = used & 1;
        for (int r = 0; r < g.nrows; ++r) rows[8 * a + 4 * h + r] = wrow(g.row0 + r);  // rows 8a+4h.. are contiguous
        pk.aux[a][h] = aux_slot[s];
        pk.seg[a][h] = (int32_t)s;
        if (pair) { pk.aux[a][1] = aux_slot[s]; pk.seg[a][1] = (int32_t)s; pair_bits |= 1 << a; }
        used += pair ? 2 : 1;
      }
      flush();
    }
    s0 = s1;
  }
  if (!p->out_identity) {
    const int k_tiles = (p->k + 31) / 32;
    for (int t = 0; t < k_tiles; ++t) {
      std::vector<const double*> rows;
      for (int r = 32 * t; r < 32 * t + 32 && r < p->k; ++r) rows.push_back(p->NA_E.data() + (size_t)r * p->n);
      b.add_tile(rows, p->n);
      MItem it = blank(MI_OUT);
      it.row0 = 32 * t;
      it.flags = (t == 0 ? MF_FIRST : 0) | (t == k_tiles - 1 ? MF_LAST : 0);
      b.items.push_back(it);
    }
  }
  return RAYEN_OK;
}


}  // namespace rayen
