// fp32 MFMA path with the module's mapper fused in front (NKX > 0 instances of the forward kernel):
//   y = project(Wm x + b)   -- rayen/constraint_module.py:525 (mapper) + :468-474 (forwardForRAYEN)
// in one launch; v = Wm x + b stays in registers (written out only when the caller wants it for the
// backward).  Kept in its own translation unit so that it compiles next to rayen_mfma.hip.
#include "rayen_mfma_kernel.h"

namespace rayen {

// in_dim up to 64 (two 32-column blocks of x per sample tile in registers next to v)
bool mfma_mapper_fusable(const RayenPack* p, const MfmaImage* img, int in_dim) {
  (void)p;
  return img != nullptr && img->nkk >= 1 && img->nkk <= 4 && in_dim >= 4 && in_dim <= 64 && in_dim % 4 == 0;
}

int mfma_forward_mapped(const RayenPack* p, const MfmaImage* img, const float* x, int64_t B, int64_t ldx,
                        int in_dim, const float* w, int64_t ldw, const float* bias, float* v_out,
                        int64_t ldvo, float* y, int64_t ldy, float* kappa, int32_t* active,
                        int32_t* nan_flag, hipStream_t stream) {
  if (!mfma_mapper_fusable(p, img, in_dim) || ldw % 4 != 0 || (reinterpret_cast<uintptr_t>(w) & 15) != 0)
    return RAYEN_E_UNSUPPORTED;
  if (B == 0) return RAYEN_OK;
  MapperArgs mp;
  mp.w = w;
  mp.ldw = ldw;
  mp.bias = bias;
  mp.in_dim = in_dim;
  mp.v_out = v_out;
  mp.ldvo = ldvo;
  const int nkx = (in_dim + 31) / 32;
#define RAYEN_MAPPED_CASE(NKK, NKX) \
  if (img->nkk == NKK && nkx == NKX) \
    return launch_mfma<NKK, NKX>(p, img, x, B, ldx, y, ldy, kappa, active, nan_flag, 0, mp, stream);
  RAYEN_MAPPED_CASE(1, 1)
  RAYEN_MAPPED_CASE(1, 2)
  RAYEN_MAPPED_CASE(2, 1)
  RAYEN_MAPPED_CASE(2, 2)
  RAYEN_MAPPED_CASE(3, 1)
  RAYEN_MAPPED_CASE(3, 2)
  RAYEN_MAPPED_CASE(4, 1)
  RAYEN_MAPPED_CASE(4, 2)
#undef RAYEN_MAPPED_CASE
  return RAYEN_E_UNSUPPORTED;
}

}  // namespace rayen
