// fp32 instances of the wave-per-sample LMI kernels (see rayen_lmi_wave.h).
#include "rayen_lmi_wave.h"

namespace rayen {

bool lmi_wave_eligible_f32(const RayenPack* p) { return lw::lmi_wave_eligible_t<float>(p); }
int lmi_wave_build_f32(const RayenPack* p, LmiWaveImage** out, int64_t* bytes) { return lw::lmi_wave_build_t<float>(p, out, bytes); }
bool lmi_wave_serves_f32(const LmiWaveImage* img) { return lw::lmi_wave_serves_t<float>(img); }
void lmi_wave_free(LmiWaveImage* img) { lw::lmi_wave_free_image(img); }
int lmi_wave_forward_f32(const RayenPack* p, const LmiWaveImage* img, const float* v, int64_t B, int64_t ldv, float* y,
                         int64_t ldy, float* kappa, int32_t* active, int32_t* nan_flag, hipStream_t stream) {
  return lw::lmi_wave_forward_t<float>(p, img, v, B, ldv, y, ldy, kappa, active, nan_flag, stream);
}
int lmi_wave_backward_f32(const RayenPack* p, const LmiWaveImage* img, const float* v, int64_t B, int64_t ldv,
                          const float* kappa, const int32_t* active, const float* grad_y, int64_t ldg, float* grad_v,
                          int64_t ldgv, hipStream_t stream) {
  return lw::lmi_wave_backward_t<float>(p, img, v, B, ldv, kappa, active, grad_y, ldg, grad_v, ldgv, stream);
}

}  // namespace rayen
