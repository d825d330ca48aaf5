// fp64 instances of the four-lanes-per-sample LMI kernel (see rayen_lmi_quad.h).
#include "rayen_lmi_quad.h"

namespace rayen {

bool lmi_quad_eligible_f64(const RayenPack* p) { return lq::lmi_quad_eligible_t<double>(p); }
int lmi_quad_build_f64(const RayenPack* p, LmiQuadImage** out, int64_t* bytes) {
  return lq::lmi_quad_build_t<double>(p, out, bytes);
}
int lmi_quad_forward_f64(const RayenPack* p, const LmiQuadImage* img, const double* v, int64_t B, int64_t ldv,
                         double* y, int64_t ldy, double* kappa, int32_t* active, int32_t* nan_flag,
                         hipStream_t stream) {
  return lq::lmi_quad_forward_t<double>(p, img, v, B, ldv, y, ldy, kappa, active, nan_flag, stream);
}

bool lmi_quad_bwd_serves_f64(const RayenPack* p, const LmiQuadImage* img) { return lq::lmi_quad_bwd_serves<double>(p, img); }
int lmi_quad_backward_f64(const RayenPack* p, const LmiQuadImage* img, const double* v, int64_t B, int64_t ldv,
                          const double* kappa, const int32_t* active, const double* grad_y, int64_t ldg, double* grad_v,
                          int64_t ldgv, hipStream_t stream) {
  return lq::lmi_quad_backward_t<double>(p, img, v, B, ldv, kappa, active, grad_y, ldg, grad_v, ldgv, stream);
}

}  // namespace rayen
