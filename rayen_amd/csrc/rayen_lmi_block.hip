// Instances of the workgroup-per-sample LMI forward (see rayen_lmi_block.h).
#include "rayen_lmi_block.h"

namespace rayen {

// [linear rows] + one LMI whose packed triangle fits the LDS
template <typename T>
static bool eligible(const RayenPack* p) {
  int n_lmi = 0, r = 0;
  for (const RayenSegment& g : p->segs) {
    if (g.type == RAYEN_SEG_LMI) { ++n_lmi; r = g.dim; }
    else if (g.type != RAYEN_SEG_LIN) return false;
  }
  return n_lmi == 1 && r >= 2 && (lb::head_cols_fwd<T>(r, p->n) >= 0 || lb::head_cols_fwd<T>(r, 0) >= 0);   // (fused | products)
}
// the same with quadratics / cones next to the LMI (another kernel's: rayen_abi.hip::mixed_forward)
template <typename T>
static bool eligible_mixed(const RayenPack* p) {
  int n_lmi = 0, n_other = 0, r = 0;
  for (const RayenSegment& g : p->segs) {
    if (g.type == RAYEN_SEG_LMI) { ++n_lmi; r = g.dim; }
    else if (g.type != RAYEN_SEG_LIN) ++n_other;
  }
  return n_lmi == 1 && n_other > 0 && r >= 2 && lb::head_cols_fwd<T>(r, p->n) >= 0 && lb::head_cols_bwd<T>(r, p->n) >= 0;
}
bool lmi_block_eligible_mixed_f32(const RayenPack* p) { return eligible_mixed<float>(p); }
bool lmi_block_eligible_mixed_f64(const RayenPack* p) { return eligible_mixed<double>(p); }
bool lmi_block_eligible_f32(const RayenPack* p) { return eligible<float>(p); }
bool lmi_block_eligible_f64(const RayenPack* p) { return eligible<double>(p); }
bool lmi_block_serves_f32(const LmiWaveImage* img) { return lb::lmi_block_serves_t<float>(img); }
bool lmi_block_serves_f64(const LmiWaveImage* img) { return lb::lmi_block_serves_t<double>(img); }
int lmi_block_prepare_f32(const LmiWaveImage* img) { return lb::lmi_block_prepare_t<float>(img); }
int lmi_block_prepare_f64(const LmiWaveImage* img) { return lb::lmi_block_prepare_t<double>(img); }
int lmi_block_forward_f32(const RayenPack* p, const LmiWaveImage* img, const float* v, int64_t B, int64_t ldv, float* y,
                          int64_t ldy, float* kappa, int32_t* active, int32_t* nan_flag, hipStream_t stream,
                          const float* kappa_in, int64_t ldk_in, int old_mode) {
  return lb::lmi_block_forward_t<float>(p, img, v, B, ldv, y, ldy, kappa, active, nan_flag, stream, kappa_in, ldk_in, nullptr, 0, old_mode);
}
int lmi_block_forward_f64(const RayenPack* p, const LmiWaveImage* img, const double* v, int64_t B, int64_t ldv, double* y,
                          int64_t ldy, double* kappa, int32_t* active, int32_t* nan_flag, hipStream_t stream,
                          const double* kappa_in, int64_t ldk_in, int old_mode) {
  return lb::lmi_block_forward_t<double>(p, img, v, B, ldv, y, ldy, kappa, active, nan_flag, stream, kappa_in, ldk_in, nullptr, 0, old_mode);
}

bool lmi_block_bwd_serves_f32(const LmiWaveImage* img) { return lb::lmi_block_bwd_serves_t<float>(img); }
bool lmi_block_bwd_serves_f64(const LmiWaveImage* img) { return lb::lmi_block_bwd_serves_t<double>(img); }
int lmi_block_backward_f32(const RayenPack* p, const LmiWaveImage* img, const float* v, int64_t B, int64_t ldv,
                           const float* kappa, const int32_t* active, const float* grad_y, int64_t ldg, float* grad_v,
                           int64_t ldgv, hipStream_t stream, int only_lmi, int old_mode) {
  return lb::lmi_block_backward_t<float>(p, img, v, B, ldv, kappa, active, grad_y, ldg, grad_v, ldgv, stream, only_lmi, nullptr, 0, nullptr, 0,
                                         nullptr, old_mode);
}
int lmi_block_backward_f64(const RayenPack* p, const LmiWaveImage* img, const double* v, int64_t B, int64_t ldv,
                           const double* kappa, const int32_t* active, const double* grad_y, int64_t ldg, double* grad_v,
                           int64_t ldgv, hipStream_t stream, int only_lmi, int old_mode) {
  return lb::lmi_block_backward_t<double>(p, img, v, B, ldv, kappa, active, grad_y, ldg, grad_v, ldgv, stream, only_lmi, nullptr, 0, nullptr, 0,
                                         nullptr, old_mode);
}
// the products route (sets with many generators): T = v W_ext' comes from a library GEMM, see rayen_abi.hip
bool lmi_block_products_serves_f32(const LmiWaveImage* img) {
  return img != nullptr && lb::plan_for<float>(img->r, 0, false).nth != 0 && lb::plan_for<float>(img->r, 0, true).nth != 0;
}
bool lmi_block_products_serves_f64(const LmiWaveImage* img) {
  return img != nullptr && lb::plan_for<double>(img->r, 0, false).nth != 0 && lb::plan_for<double>(img->r, 0, true).nth != 0;
}
int lmi_block_forward_products_f32(const RayenPack* p, const LmiWaveImage* img, const float* prods, int64_t ldt, const float* v,
                                   int64_t B, int64_t ldv, float* y, int64_t ldy, float* kappa, int32_t* active,
                                   int32_t* nan_flag, hipStream_t stream) {
  return lb::lmi_block_forward_t<float>(p, img, v, B, ldv, y, ldy, kappa, active, nan_flag, stream, nullptr, 1, prods, ldt);
}
int lmi_block_forward_products_f64(const RayenPack* p, const LmiWaveImage* img, const double* prods, int64_t ldt, const double* v,
                                   int64_t B, int64_t ldv, double* y, int64_t ldy, double* kappa, int32_t* active,
                                   int32_t* nan_flag, hipStream_t stream) {
  return lb::lmi_block_forward_t<double>(p, img, v, B, ldv, y, ldy, kappa, active, nan_flag, stream, nullptr, 1, prods, ldt);
}
int lmi_block_bwd_coefficients_f32(const RayenPack* p, const LmiWaveImage* img, const float* prods, int64_t ldt, const float* v,
                                   int64_t B, int64_t ldv, const float* kappa, const int32_t* active, const float* grad_y,
                                   int64_t ldg, float* C, int64_t ldc, float* gs, hipStream_t stream) {
  return lb::lmi_block_backward_t<float>(p, img, v, B, ldv, kappa, active, grad_y, ldg, nullptr, 0, stream, 0, prods, ldt, C, ldc, gs);
}
int lmi_block_bwd_coefficients_f64(const RayenPack* p, const LmiWaveImage* img, const double* prods, int64_t ldt, const double* v,
                                   int64_t B, int64_t ldv, const double* kappa, const int32_t* active, const double* grad_y,
                                   int64_t ldg, double* C, int64_t ldc, double* gs, hipStream_t stream) {
  return lb::lmi_block_backward_t<double>(p, img, v, B, ldv, kappa, active, grad_y, ldg, nullptr, 0, stream, 0, prods, ldt, C, ldc, gs);
}
}  // namespace rayen

#ifdef RAYEN_LB_PROFILE
extern "C" void rayen_debug_lb_prof(unsigned long long* out8) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out8, HIP_SYMBOL(rayen::lb::g_lb_prof), 8 * sizeof(unsigned long long));
  unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(rayen::lb::g_lb_prof), zero, sizeof(zero));
}
#endif
