// fp32 MFMA path (placeholder until the fragment-ordered kernel lands; the
// dispatcher in rayen_abi.hip falls through to the generic kernels).
#include "rayen_internal.h"

namespace rayen {

struct MfmaImage {};

bool mfma_eligible(const RayenPack*) { return false; }
int mfma_build(const RayenPack*, MfmaImage**, int64_t*) { return RAYEN_E_UNSUPPORTED; }
void mfma_free(MfmaImage* img) { delete img; }
int mfma_forward(const RayenPack*, const MfmaImage*, const float*, int64_t, int64_t, float*, int64_t,
                 float*, int32_t*, int32_t*, hipStream_t) {
  return RAYEN_E_UNSUPPORTED;
}

}  // namespace rayen
