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

// Fills `b` (raw fp64 tiles: the item tiles, a no-op tile if their count is odd, two spare tiles for
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
  // two spare tiles: the bucketed walk of a form that ends the list fetches [it_lo, it_hi) two tiles ahead, so its last
  // iteration touches it_hi and it_hi + 1 (never used; a read past the allocation could still fault)
  b.add_tile({}, n);
  b.add_tile({}, n);
  const int n_real = (int)items.size();
  if (items.empty()) {
    BItem it;
    std::memset(&it, 0, sizeof(it));
    items.push_back(it);  // never read (n_items = 0), keeps the allocation non-empty
  }
  return n_real;
}

// ---------------------------------------------------------------------------------------------
// General shapes (rayen_mfma_bwdg.hip / rayen_mfma_bwdg64.hip): dense forms for the large segments,
// packed tiles (each followed by its transpose) for the small factor segments.
// ---------------------------------------------------------------------------------------------
enum : int32_t { BI_PACK1 = 3, BI_PACK2 = 4 };

struct BPack {
  int32_t seg[4][2];   // [quad a][half]: caller's segment index sitting there, -1 = empty
  int32_t pair_bits;   // bit a: the segment of quad a spans both halves (rank 5..8)
  int32_t reserved;
};


// Fills `b` (item tiles, a no-op tile if their count is odd, two spare tiles), `items`, `packs` (never
// empty) and `seg_aux` ([n_segments + 1]: W row of phi for factor segments, -1 otherwise); returns the
// number of items the kernel walks (even).
inline int layout_bwdg_tiles(const RayenPack* p, TileLayout& b, std::vector<BItem>& items,
                             std::vector<BPack>& packs, std::vector<int32_t>& seg_aux,
                             std::vector<int32_t>* seg_group = nullptr, std::vector<int32_t>* group_items = nullptr) {
  // seg_group[s]: index of the item group that serves segment s (-1: none: linear rows) and group_items[2 g], [2 g + 1]:
  // that group's item range -- a dense form's row tiles, or the PACK1 / PACK2 pair of a packed tile (the bucketed walk)
  if (seg_group) seg_group->assign(p->segs.size() + 1, -1);
  if (group_items) group_items->clear();
  std::vector<size_t> in_pack;
  const int n = p->n, np = n_pad_of(n), nkk = np / 32;
  const double* W = p->W.data();
  seg_aux.assign(p->segs.size() + 1, -1);
  auto blank = [](int type) { BItem it; std::memset(&it, 0, sizeof(it)); it.type = type; return it; };

  // ---- dense forms for everything that is not a small factor (same as rayen_bwd_tiles.h)
  for (size_t s = 0; s < p->segs.size(); ++s) {
    const RayenSegment& g = p->segs[s];
    if (g.type == RAYEN_SEG_QUAD_FAC) seg_aux[s] = g.aux_row;
    if (!bwd_quad_like(g) || is_small_factor(g)) continue;
    std::vector<double> S((size_t)n * n, 0.0);
    if (g.type == RAYEN_SEG_QUAD_SYM) {
      for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) S[(size_t)i * n + j] = W[(size_t)(g.row0 + i) * n + j];
    } else {
      for (int r = 0; r < g.nrows; ++r) {
        const double* row = W + (size_t)(g.row0 + r) * n;
        for (int i = 0; i < n; ++i) {
          if (row[i] == 0.0) continue;
          for (int j = 0; j < n; ++j) S[(size_t)i * n + j] += row[i] * row[j];
        }
      }
    }
    if (seg_group && group_items) {
      (*seg_group)[s] = (int32_t)(group_items->size() / 2);
      group_items->push_back((int32_t)items.size());
      group_items->push_back((int32_t)items.size() + nkk);
    }
    for (int tp = 0; tp < nkk; ++tp) {
      std::vector<const double*> rows;
      for (int r = 32 * tp; r < 32 * tp + 32 && r < n; ++r) rows.push_back(S.data() + (size_t)r * n);
      b.add_tile(rows, n);
      BItem it = blank(g.type == RAYEN_SEG_SOC ? BI_SOC : BI_QUAD);
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
  // ---- small factors: eight half-quads of rows per tile (placement of rayen_tiles.h), each tile followed
  // by its transpose
  {
    std::vector<const double*> rows(32, nullptr);
    BPack pk;
    int used = 0;
    auto reset = [&]() {
      std::fill(rows.begin(), rows.end(), nullptr);
      std::memset(&pk, 0, sizeof(pk));
      for (int a = 0; a < 4; ++a) for (int h = 0; h < 2; ++h) pk.seg[a][h] = -1;
      used = 0;
    };
    auto flush = [&]() {
      if (used == 0) return;
      if (seg_group && group_items) {
        for (size_t sidx : in_pack) (*seg_group)[sidx] = (int32_t)(group_items->size() / 2);
        group_items->push_back((int32_t)items.size());
        group_items->push_back((int32_t)items.size() + 2);
      }
      in_pack.clear();
      BItem it1 = blank(BI_PACK1);
      it1.aux_row = (int32_t)packs.size();
      items.push_back(it1);
      b.add_tile(rows, n);
      // transposed: raw2[r][32 tp + kk] = tile[kk][32 tp + r]
      std::vector<std::vector<double>> tr(32, std::vector<double>(np, 0.0));
      for (int kk = 0; kk < 32; ++kk) {
        if (rows[kk] == nullptr) continue;
        for (int e = 0; e < n; ++e) tr[e % 32][32 * (e / 32) + kk] = rows[kk][e];
      }
      std::vector<const double*> trp;
      for (auto& r : tr) trp.push_back(r.data());
      BItem it2 = blank(BI_PACK2);
      items.push_back(it2);
      b.add_tile(trp, np);
      packs.push_back(pk);
      reset();
    };
    reset();
    for (size_t s = 0; s < p->segs.size(); ++s) {
      const RayenSegment& g = p->segs[s];
      if (!is_small_factor(g)) continue;
      const bool pair = g.nrows > 4;
      if (pair && (used & 1)) ++used;
      if (used + (pair ? 2 : 1) > 8) flush();
      const int a = used / 2, h = used & 1;
      for (int r = 0; r < g.nrows; ++r) rows[8 * a + 4 * h + r] = W + (size_t)(g.row0 + r) * n;
      pk.seg[a][h] = (int32_t)s;
      if (pair) { pk.seg[a][1] = (int32_t)s; pk.pair_bits |= 1 << a; }
      used += pair ? 2 : 1;
      in_pack.push_back(s);
    }
    flush();
  }
  // a PACK1 / PACK2 pair must not straddle the kernel's (it, it+1) unrolling in a way that matters: the
  // kernel keeps wreg across process() calls, so any order works; only the count must be even
  if (items.size() % 2) {
    items.push_back(blank(BI_NOP));
    b.add_tile({}, n);
  }
  b.add_tile({}, n);  // two spare tiles for the prefetch (it looks two tiles ahead of the last item)
  b.add_tile({}, n);
  const int n_real = (int)items.size();
  if (items.empty()) items.push_back(blank(BI_NOP));
  if (packs.empty()) { BPack pk; std::memset(&pk, 0, sizeof(pk)); packs.push_back(pk); }
  return n_real;

}

constexpr int kBwdgMaxN = 64;

// Shape test of the general backward kernels: n, k <= 64, no LMI (the fp64 twin adds n <= 32).
inline bool bwdg_tiles_eligible(const RayenPack* p) {
  if (p->n > kBwdgMaxN || p->k > 64) return false;
  int64_t tiles = 0;
  int small = 0;
  for (const RayenSegment& g : p->segs) {
    if (g.type == RAYEN_SEG_LMI) return false;
    if (!bwd_quad_like(g)) continue;
    if (is_small_factor(g)) ++small;
    else tiles += n_pad_of(p->n) / 32;
  }
  tiles += 2 * ((small + 3) / 4);  // at least four (rank 5..8) and at most eight segments per packed tile pair
  return tiles <= 96;
}

}  // namespace rayen
