// fp32 MFMA path, host side: eligibility, image construction, dispatch.  The forward kernel itself lives in
// rayen_mfma_kernel.h; the split-operand kernel that serves most packs instead is rayen_mfma_split.hip.
#include "rayen_mfma_kernel.h"

#include <cstring>
#include <vector>

namespace rayen {

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
  if (img->y0) (void)hipFree(img->y0);
  delete img;
}

int mfma_forward(const RayenPack* p, const MfmaImage* img, const float* v, int64_t B, int64_t ldv, float* y,
                 int64_t ldy, float* kappa, int32_t* active, int32_t* nan_flag, int old_mode,
                 hipStream_t stream) {
  if (B == 0) return RAYEN_OK;
  switch (img->nkk) {
    case 1: return launch_mfma<1, 0>(p, img, v, B, ldv, y, ldy, kappa, active, nan_flag, old_mode, MapperArgs(), stream);
    case 2: return launch_mfma<2, 0>(p, img, v, B, ldv, y, ldy, kappa, active, nan_flag, old_mode, MapperArgs(), stream);
    case 3: return launch_mfma<3, 0>(p, img, v, B, ldv, y, ldy, kappa, active, nan_flag, old_mode, MapperArgs(), stream);
    case 4: return launch_mfma<4, 0>(p, img, v, B, ldv, y, ldy, kappa, active, nan_flag, old_mode, MapperArgs(), stream);
    default: return RAYEN_E_UNSUPPORTED;
  }
}

}  // namespace rayen
