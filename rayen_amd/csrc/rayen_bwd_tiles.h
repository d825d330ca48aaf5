// Tile list of the matrix-core backward kernels (private to librayen_hip.so): one dense n_pad x n_pad
// symmetric form S per quadratic / cone segment (S = G, U'U or M'M), as full-width 32-row tiles.
#pragma once

#include <cstring>
#include <vector>

#include "rayen_tiles.h"

namespace rayen {

enum : int32_t { BI_NOP = 0, BI_QUAD = 1, BI_SOC = 2 };

struct BItem {
  int32_t type;
  int32_t flags;    // MF_FIRST | MF_LAST of the segment's row tiles
  int32_t seg;      // caller's segment index (what `active` holds)
  int32_t tp;       // row tile of S (rows 32 tp .. 32 tp + 31 = elements of v)
  int32_t aux_row;  // W row of phi | c (M'beta is the next row)
  int32_t reserved;
  float f0, f1;     // SOC: tau, a'
  double f0d, f1d;  // the same in full precision (fp64 kernel)
};

inline bool bwd_quad_like(const RayenSegment& g) {
  return g.type == RAYEN_SEG_QUAD_SYM || g.type == RAYEN_SEG_QUAD_FAC || g.type == RAYEN_SEG_SOC;
}

// Shape test shared by the fp32 and fp64 backward: NA_E = I, n <= 64, no LMI, and not too many
// segments -- the walk is dense (one n x n form per quadratic / cone), so sets made of very many small
// low-rank quadratics are cheaper on the per-lane generic backward.
inline bool bwd_tiles_eligible(const RayenPack* p) {
  if (!p->out_identity || p->n > 64) return false;
  int64_t tiles = 0;
  for (const RayenSegment& g : p->segs) {
    if (g.type == RAYEN_SEG_LMI) return false;
    if (bwd_quad_like(g)) tiles += n_pad_of(p->n) / 32;
  }
  return tiles <= 64;
}

// Fills `b` (raw fp64 tiles: the item tiles, a no-op tile if their count is odd, one spare tile for
// the prefetch) and `items` (never empty); returns the number of items the kernel walks (even).
inline int layout_bwd_tiles(const RayenPack* p, TileLayout& b, std::vector<BItem>& items) {
  const int n = p->n, nkk = n_pad_of(n) / 32;
  const double* W = p->W.data();
  for (size_t s = 0; s < p->segs.size(); ++s) {
    const RayenSegment& g = p->segs[s];
    if (!bwd_quad_like(g)) continue;
    std::vector<double> S((size_t)n * n, 0.0);
    if (g.type == RAYEN_SEG_QUAD_SYM) {
      for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) S[(size_t)i * n + j] = W[(size_t)(g.row0 + i) * n + j];
    } else {  // U'U or M'M
      for (int r = 0; r < g.nrows; ++r) {
        const double* row = W + (size_t)(g.row0 + r) * n;
        for (int i = 0; i < n; ++i) {
          if (row[i] == 0.0) continue;
          for (int j = 0; j < n; ++j) S[(size_t)i * n + j] += row[i] * row[j];
        }
      }
    }
    for (int tp = 0; tp < nkk; ++tp) {
      std::vector<const double*> rows;
      for (int r = 32 * tp; r < 32 * tp + 32 && r < n; ++r) rows.push_back(S.data() + (size_t)r * n);
      b.add_tile(rows, n);
      BItem it;
      std::memset(&it, 0, sizeof(it));
      it.type = g.type == RAYEN_SEG_SOC ? BI_SOC : BI_QUAD;
      it.flags = (tp == 0 ? MF_FIRST : 0) | (tp == nkk - 1 ? MF_LAST : 0);
      it.seg = (int32_t)s;
      it.tp = tp;
      it.aux_row = g.aux_row;
      it.f0 = (float)g.f0;
      it.f1 = (float)g.f1;
      it.f0d = g.f0;
      it.f1d = g.f1;
      items.push_back(it);
    }
  }
  if (items.size() % 2) {
    BItem it;
    std::memset(&it, 0, sizeof(it));
    it.type = BI_NOP;
    items.push_back(it);
    b.add_tile({}, n);
  }
  b.add_tile({}, n);  // spare tile: the prefetch runs one tile past the end
  const int n_real = (int)items.size();
  if (items.empty()) {
    BItem it;
    std::memset(&it, 0, sizeof(it));
    items.push_back(it);  // never read (n_items = 0), keeps the allocation non-empty
  }
  return n_real;
}

}  // namespace rayen
