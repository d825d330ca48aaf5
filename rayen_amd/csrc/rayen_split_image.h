// Shared by the two split-operand forward kernels (rayen_mfma_split.hip, rayen_mfma_split4.hip): vector types and the
// device image of the constants (three bf16 pieces of every entry in MFMA fragment order + the item list).
#pragma once

#include "rayen_mfma_kernel.h"

namespace rayen {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct SplitImage {
  void* Wb = nullptr;      // [n_tiles][NS][3][64] x 8 bf16
  MItem* items = nullptr;
  MPack* packs = nullptr;
  float* y0 = nullptr;
  int n_items = 0;
  int nkk = 0;
  int identity = 0;
  int n_simd = 1024;
  int wave1 = 1;           // large batches of NA_E = I packs run one wave per SIMD (rayen_mfma_split4.hip)
  int64_t bytes = 0;
};

bool mfma_split4_serves(const RayenPack* p, const SplitImage* img);
int mfma_split4_forward(const RayenPack* p, const SplitImage* img, const float* v, int64_t B, int64_t ldv, float* y,
                        int64_t ldy, float* kappa, int32_t* active, int32_t* nan_flag, hipStream_t stream);

}  // namespace rayen
