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
  int32_t tile_shape;  // image tile this item reads | MS_* << 24 (which K-steps of that tile are the item's: f16-pair image)
  double f0d, f1d; // the same in full precision (fp64 kernel)
  __host__ __device__ int tile() const { return tile_shape & 0xFFFFFF; }
  __host__ __device__ int shape() const { return (tile_shape >> 24) & 3; }
  __host__ __device__ bool last_tile() const { return (tile_shape >> 30) & 1; }   // the walk wraps to tile 0 behind it
};

// An item reads a whole tile, or -- f16-pair image, n_pad = 64 -- the half of a tile that two triangular factors share.
// A factor U of more than 32 rows only ever enters through ||U v|| (rayen/constraint_module.py:360-399), so it is
// replaced by a triangular R with R'R = U'U (upper_triangular_factor): 32 of its rows are dense (a full tile), the
// others are zero over one 32-column half of the direction and keep a block of at most 32 x 32.  Two neighbouring
// factors X, Y are eliminated in opposite column orders, so that X's block sits over columns 0..31 and Y's over columns
// 32..63, and the two blocks share ONE tile: K-steps 0,1 of it are X's (MS_HALF_A), K-steps 2,3 are Y's (MS_HALF_B),
// each against the direction's own K-steps -- the operand pairing of a full tile, only the accumulator starts afresh at
// K-step 2.  The walk is [X dense][shared: X's block][shared: Y's block][Y dense]: four items on three tiles.
// A triangular factor without such a neighbour keeps two full tiles (the zero block of its second one is multiplied).
enum : int32_t { MS_FULL = 0, MS_HALF_A = 1, MS_HALF_B = 2 };

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

// Upper-triangular factor of a set of rows: R (min(m, n) x n, zero below the diagonal) with R'R = U'U, by Householder
// reflections in fp64 -- ||R v|| = ||U v|| for every v (rayen/constraint_module.py:360-399 only ever asks for that norm).
// Rows that come out all zero (rank-deficient U) are dropped from the end.
// `order`: the column elimination order (order[j] = original column eliminated j-th; empty = 0, 1, 2, ...): row i of the
// result is zero in the columns order[0..i-1].  Rows come back in ORIGINAL column positions.
inline std::vector<std::vector<double>> upper_triangular_factor(const std::vector<const double*>& rows, int n,
                                                                const std::vector<int>& order = std::vector<int>()) {
  const int m = (int)rows.size();
  std::vector<double> A((size_t)m * n);
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < n; ++j) A[(size_t)i * n + j] = rows[i][order.empty() ? j : order[j]];
  const int steps = m < n ? m : n;
  std::vector<double> w(m);
  for (int j = 0; j < steps; ++j) {
    double norm = 0.0;
    for (int i = j; i < m; ++i) norm += A[(size_t)i * n + j] * A[(size_t)i * n + j];
    norm = std::sqrt(norm);
    if (!(norm > 0.0)) continue;
    const double alpha = A[(size_t)j * n + j] > 0.0 ? -norm : norm;
    for (int i = j; i < m; ++i) w[i] = A[(size_t)i * n + j];
    w[j] -= alpha;
    double wn = 0.0;
    for (int i = j; i < m; ++i) wn += w[i] * w[i];
    if (!(wn > 0.0)) continue;
    for (int c = j; c < n; ++c) {
      double dot = 0.0;
      for (int i = j; i < m; ++i) dot += w[i] * A[(size_t)i * n + c];
      const double f = 2.0 * dot / wn;
      for (int i = j; i < m; ++i) A[(size_t)i * n + c] -= f * w[i];
    }
    A[(size_t)j * n + j] = alpha;
    for (int i = j + 1; i < m; ++i) A[(size_t)i * n + j] = 0.0;
  }
  std::vector<std::vector<double>> R;
  for (int i = 0; i < steps; ++i) {
    std::vector<double> r(n, 0.0);
    for (int c = i; c < n; ++c) r[order.empty() ? c : order[c]] = A[(size_t)i * n + c];
    R.push_back(r);
  }
  while (!R.empty()) {
    bool zero = true;
    for (const double x : R.back()) zero = zero && x == 0.0;
    if (!zero) break;
    R.pop_back();
  }
  return R;
}

// allow_sym = false: every quadratic / cone is laid out through a factor (rows u with ||U v||^2 = v'Gv), so
// that no epilogue needs the direction itself (the split-operand kernel keeps v only as bf16 pieces).
// tri (f16-pair image, with allow_sym = false, n_pad = 64): factors of more than 32 rows are made upper triangular
// (above) and two neighbours share ONE tile for their second halves: [X rows 0..31] [X rows 32.. | Y rows 32..] [Y rows
// 0..31] -- three tiles of matrix work and of A stream where four were (config 3: 14 tiles instead of 17).
inline int layout_tiles(const RayenPack* p, TileLayout& b, bool allow_pack, bool allow_sym = true, bool tri = false) {
  const double* W = p->W.data();
  auto wrow = [&](int r) { return W + (size_t)r * p->n; };
  auto blank = [](int type) { MItem it; std::memset(&it, 0, sizeof(it)); it.type = type; it.seg_inv = 1.f; return it; };
  tri = tri && !allow_sym && b.n_pad == 64;
  // the factor rows a segment is laid out through when `tri` applies to it (empty: not triangular)
  // swapped: columns 32.. are eliminated first, so the rows from n - 32 on keep columns 0..31 only (the X of a pair);
  // otherwise the rows from 32 on keep columns 32.. only (the Y of a pair, or a factor on its own)
  auto tri_rows = [&](const RayenSegment& g, bool swapped) {
    std::vector<std::vector<double>> none;
    if (!tri || (allow_pack && is_small_factor(g))) return none;
    std::vector<const double*> src;
    std::vector<std::vector<double>> eig;
    if (g.type == RAYEN_SEG_QUAD_SYM) {
      eig = psd_factor_rows(wrow(g.row0), p->n);
      for (const auto& u : eig) src.push_back(u.data());
    } else if (g.type == RAYEN_SEG_QUAD_FAC || g.type == RAYEN_SEG_SOC) {
      for (int r = 0; r < g.nrows; ++r) src.push_back(wrow(g.row0 + r));
    } else {
      return none;
    }
    if ((int)src.size() <= 32) return none;
    for (const double* r : src)
      for (int c = 0; c < p->n; ++c) if (!std::isfinite(r[c])) return none;
    std::vector<int> order;
    if (swapped) {
      for (int c = 32; c < p->n; ++c) order.push_back(c);
      for (int c = 0; c < 32; ++c) order.push_back(c);
    }
    std::vector<std::vector<double>> R = upper_triangular_factor(src, p->n, order);
    if ((int)R.size() <= 32) return none;
    return R;
  };
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
      b.items.back().tile_shape = b.n_tiles();
      b.add_tile(rows, p->n);
    }
    // ---- large segments: their own tiles
    for (size_t s = s0; s < s1; ++s) {
      const RayenSegment& g = p->segs[s];
      if (allow_pack && is_small_factor(g)) continue;
      if (tri) {
        // the next large segment of the batch, if it is triangular too: the two share a tile
        size_t sy = s + 1;
        while (sy < s1 && allow_pack && is_small_factor(p->segs[sy])) ++sy;
        std::vector<std::vector<double>> RY;
        if (sy < s1) RY = tri_rows(p->segs[sy], /*swapped=*/false);
        std::vector<std::vector<double>> RX = tri_rows(g, /*swapped=*/!RY.empty());
        if (RX.empty()) RY.clear();
        if (!RX.empty()) {
          auto seg_item = [&](size_t sg, int flags, int shape) {
            const RayenSegment& gg = p->segs[sg];
            MItem it = blank(gg.type == RAYEN_SEG_SOC ? MI_SOC : MI_QFAC);
            it.seg = (int32_t)sg;
            it.aux = aux_slot[sg];
            it.f0 = (float)gg.f0;
            it.f1 = (float)gg.f1;
            it.flags = flags;
            it.tile_shape = b.n_tiles() | (shape << 24);
            return it;
          };
          auto own = [&](std::vector<std::vector<double>>& R) {
            const size_t first = b.owned.size();
            for (auto& r : R) b.owned.push_back(std::move(r));
            return first;
          };
          const int nx = (int)RX.size();
          const size_t x0 = own(RX);
          std::vector<const double*> rows;
          if (!RY.empty()) {
            // X, columns 32.. eliminated first: rows 0 .. n-33 dense, rows n-32 .. keep columns 0..31
            const int ny = (int)RY.size(), xh = p->n - 32;
            const size_t y0 = own(RY);
            for (int r = 0; r < xh; ++r) rows.push_back(b.owned[x0 + r].data());
            b.items.push_back(seg_item(s, MF_FIRST, MS_FULL));
            b.add_tile(rows, p->n);
            // the shared tile: columns 0..31 <- X's rows n-32 .., columns 32.. <- Y's rows 32 ..
            std::vector<std::vector<double>> shared(32, std::vector<double>(p->n, 0.0));
            for (int r = xh; r < nx; ++r)
              for (int c = 0; c < 32; ++c) shared[r - xh][c] = b.owned[x0 + r][c];
            for (int r = 32; r < ny; ++r)
              for (int c = 32; c < p->n; ++c) shared[r - 32][c] = b.owned[y0 + r][c];
            rows.clear();
            for (auto& r : shared) rows.push_back(r.data());
            b.items.push_back(seg_item(s, MF_LAST, MS_HALF_A));
            b.items.push_back(seg_item(sy, MF_FIRST, MS_HALF_B));
            b.add_tile(rows, p->n);
            rows.clear();
            for (int r = 0; r < 32; ++r) rows.push_back(b.owned[y0 + r].data());
            b.items.push_back(seg_item(sy, MF_LAST, MS_FULL));
            b.add_tile(rows, p->n);
            s = sy;       // (the small factor segments in between are laid out below, like all of them)
            continue;
          }
          // no partner (natural order: rows 32.. keep columns 32.. only): two ordinary tiles
          for (int r = 0; r < 32; ++r) rows.push_back(b.owned[x0 + r].data());
          b.items.push_back(seg_item(s, MF_FIRST, MS_FULL));
          b.add_tile(rows, p->n);
          rows.clear();
          for (int r = 32; r < nx; ++r) rows.push_back(b.owned[x0 + r].data());
          b.items.push_back(seg_item(s, MF_LAST, MS_FULL));
          b.add_tile(rows, p->n);
          continue;
        }
      }
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
        MItem it = blank(0);
        it.tile_shape = b.n_tiles();
        b.add_tile(rows, p->n);
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
        it.tile_shape = b.n_tiles();
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
      MItem it = blank(MI_OUT);
      it.tile_shape = b.n_tiles();
      b.add_tile(rows, p->n);
      it.row0 = 32 * t;
      it.flags = (t == 0 ? MF_FIRST : 0) | (t == k_tiles - 1 ? MF_LAST : 0);
      b.items.push_back(it);
    }
  }
  const int last = b.n_tiles() - 1;
  for (MItem& it : b.items)
    if (it.tile() == last) it.tile_shape |= 1 << 30;
  return RAYEN_OK;
}


}  // namespace rayen
