// fp32 instances of the four-lanes-per-sample LMI kernel (see rayen_lmi_quad.h).
#include "rayen_lmi_quad.h"

namespace rayen {

bool lmi_quad_eligible_f32(const RayenPack* p) { return lq::lmi_quad_eligible_t<float>(p); }
int lmi_quad_build_f32(const RayenPack* p, LmiQuadImage** out, int64_t* bytes) {
  return lq::lmi_quad_build_t<float>(p, out, bytes);
}
int lmi_quad_forward_f32(const RayenPack* p, const LmiQuadImage* img, const float* v, int64_t B, int64_t ldv,
                         float* y, int64_t ldy, float* kappa, int32_t* active, int32_t* nan_flag,
                         hipStream_t stream) {
  return lq::lmi_quad_forward_t<float>(p, img, v, B, ldv, y, ldy, kappa, active, nan_flag, stream);
}
void lmi_quad_free(LmiQuadImage* img) {
  if (img == nullptr) return;
  if (img->data) (void)hipFree(img->data);
  if (img->wrow) (void)hipFree(img->wrow);
  if (img->wm) (void)hipFree(img->wm);
  if (img->lin_id) (void)hipFree(img->lin_id);
  delete img;
}

bool lmi_quad_bwd_serves_f32(const RayenPack* p, const LmiQuadImage* img) { return lq::lmi_quad_bwd_serves<float>(p, img); }
int lmi_quad_backward_f32(const RayenPack* p, const LmiQuadImage* img, const float* v, int64_t B, int64_t ldv,
                          const float* kappa, const int32_t* active, const float* grad_y, int64_t ldg, float* grad_v,
                          int64_t ldgv, hipStream_t stream) {
  return lq::lmi_quad_backward_t<float>(p, img, v, B, ldv, kappa, active, grad_y, ldg, grad_v, ldgv, stream);
}

}  // namespace rayen
