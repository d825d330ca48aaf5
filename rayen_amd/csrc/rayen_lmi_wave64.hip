// fp64 instances of the wave-per-sample LMI kernels (see rayen_lmi_wave.h).
#include "rayen_lmi_wave.h"

namespace rayen {

bool lmi_wave_eligible_f64(const RayenPack* p) { return lw::lmi_wave_eligible_t<double>(p); }
int lmi_wave_build_f64(const RayenPack* p, LmiWaveImage** out, int64_t* bytes) { return lw::lmi_wave_build_t<double>(p, out, bytes); }
bool lmi_wave_serves_f64(const LmiWaveImage* img) { return lw::lmi_wave_serves_t<double>(img); }
int lmi_wave_forward_f64(const RayenPack* p, const LmiWaveImage* img, const double* v, int64_t B, int64_t ldv, double* y,
                         int64_t ldy, double* kappa, int32_t* active, int32_t* nan_flag, hipStream_t stream) {
  return lw::lmi_wave_forward_t<double>(p, img, v, B, ldv, y, ldy, kappa, active, nan_flag, stream);
}
int lmi_wave_backward_f64(const RayenPack* p, const LmiWaveImage* img, const double* v, int64_t B, int64_t ldv,
                          const double* kappa, const int32_t* active, const double* grad_y, int64_t ldg, double* grad_v,
                          int64_t ldgv, hipStream_t stream) {
  return lw::lmi_wave_backward_t<double>(p, img, v, B, ldv, kappa, active, grad_y, ldg, grad_v, ldgv, stream);
}

}  // namespace rayen
