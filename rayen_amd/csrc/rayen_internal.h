// Private definitions shared by the translation units of librayen_hip.so.
// Nothing here is part of the C ABI (include/rayen_hip.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "rayen_hip.h"

namespace rayen {

constexpr int kRowBlock = 8;  // rows handled together by the generic (lane = sample) path

// Device-side segment record of the generic path.  Row blocks are groups of
// kRowBlock rows stored as Wg[(rb * n + j) * kRowBlock + r] so that the eight
// multipliers of column j are one aligned, wave-uniform (scalar) load.
struct GSeg {
  int32_t type;
  int32_t aux_rb;  // row block holding the aux rows (phi | c, b) or -1
  int32_t rb0;     // first main row block
  int32_t nrb;     // number of main row blocks
  int32_t nrows;   // true number of main rows
  int32_t dim;     // LMI: r
  int32_t seg;     // index into the caller's segment table (reported in `active`)
  int32_t row0;    // first logical W row (reported for LIN)
  int32_t frb0;    // QUAD_SYM, forward: row blocks of a factor U (U'U = G, rayen_tiles.h::psd_factor_rows): ||U v||^2
  int32_t fnrb;    // instead of v'(G v) -- no cancellation between the terms of the form; 0 = none
  double f0, f1;
};

template <typename T>
struct GenericImage {
  bool built = false;
  T* Wg = nullptr;        // [n_rb, n, 8]
  T* Ng = nullptr;        // output map NA_E, row-blocked the same way: [out_nrb, n, 8] (null if identity)
  T* NTg = nullptr;       // NA_E' (rows = n, columns = k) for the backward: [ceil(n/8), k, 8] (null if identity)
  T* y0 = nullptr;        // [k]
  GSeg* segs = nullptr;   // [n_gseg]
  int n_gseg = 0;
  int n_rb = 0;
  int out_nrb = 0;
  int lmi_words = 0;      // per-sample LDS words the largest LMI needs
  bool skip_lmi = false;  // set BEFORE generic_build: LMI segments keep their index but have no rows and no type
                          // (rayen_abi.hip::mixed_forward: the workgroup-per-sample kernel evaluates the LMI)
  int64_t bytes = 0;
};

struct MfmaImage;    // rayen_mfma.hip
struct Mfma64Image;  // rayen_mfma_f64.hip
struct MfmaBwdImage; // rayen_mfma_bwd.hip
struct Mfma64BwdImage;  // rayen_mfma_bwd64.hip
struct MfmaBwdgImage;   // rayen_mfma_bwdg.hip
struct MfmaBwdpImage;   // rayen_mfma_bwdp.hip
struct MfmaBwddImage;   // rayen_mfma_bwdd.hip
struct Mfma64BwdgImage; // rayen_mfma_bwdg64.hip
struct LmiQuadImage;    // rayen_lmi_quad.h
struct LmiWaveImage;    // rayen_lmi_wave.h
struct SplitImage;      // rayen_mfma_split.hip
struct PairImage;       // rayen_mfma_pair.hip
struct Ws8Image;        // rayen_mfma_pair_ws8.hip
struct WideImage;       // rayen_wide.hip

}  // namespace rayen

// Immutable after rayen_pack_create returns (every device image is built there), so it is shared by threads and
// streams without a lock.
struct RayenPack {
  int device = -1;
  int k = 0, n = 0, n_rows = 0;
  int out_identity = 0;
  int fp32_mode = 0;             // RayenPackDesc.fp32_mode (after the environment override)
  int prepared = 0;              // RAYEN_PREPARE_F32 | RAYEN_PREPARE_F64 | 4 (backward images)
  double inward_bias = 0.0;      // eps of RAYEN_PREPARE_INWARD_BIAS: the fp32 images are built from (1 + eps) W
  std::vector<double> W;         // host copy [n_rows, n]
  std::vector<double> NA_E;      // host copy [k, n] (identity materialised)
  std::vector<double> y0;        // host copy [k]
  std::vector<RayenSegment> segs;
  mutable rayen::GenericImage<float> g32;
  mutable rayen::GenericImage<double> g64;
  rayen::MfmaImage* m32 = nullptr;
  rayen::Mfma64Image* m64 = nullptr;
  rayen::MfmaBwdImage* mb32 = nullptr;
  rayen::Mfma64BwdImage* mb64 = nullptr;
  rayen::MfmaBwdgImage* mbg32 = nullptr;
  rayen::MfmaBwdpImage* mbp32 = nullptr;   // f16-pair backward (packed low-rank quadratics, n <= 32)
  int mbp32_state = 0;           // 1: may serve the pack | 2: rejected by bwd32_selfcheck
  rayen::MfmaBwddImage* mbd32 = nullptr;   // f16-pair backward of dense forms at n = k = 64 (next to mb32, which it is measured against)
  int mbd32_state = 0;           // the same
  double check_bwd_pair = -1.0, check_bwd_exact = -1.0;  // worst gradient-row errors against the fp64 lane backward
  rayen::Mfma64BwdgImage* mbg64 = nullptr;
  rayen::LmiQuadImage* q32 = nullptr;
  rayen::LmiQuadImage* q64 = nullptr;
  rayen::LmiWaveImage* w32 = nullptr;   // wave-per-sample LMI kernels: matrices beyond the other LMI kernels' sizes
  rayen::LmiWaveImage* w64 = nullptr;
  rayen::SplitImage* sp32 = nullptr;
  int sp32_state = 0;            // 1: the bf16-triple kernel may serve this pack | 2: rejected by fp32_selfcheck
  rayen::PairImage* pr32 = nullptr;
  rayen::PairImage* pr32m = nullptr;   // the f16-pair image without shared tiles, for the instances behind the fused mapper
                                       // (null: pr32 has none either and serves them)
  rayen::Ws8Image* ws8_32 = nullptr;   // the same image dealt out to eight W-stationary waves (null: not served)
  int pr32_state = 0;            // the same for the f16-pair kernel (which is preferred when both are accepted)
  rayen::WideImage* wide = nullptr;    // segment tables of the products epilogue (any n; packs without an LMI)
  double check_split = -1.0, check_exact = -1.0, check_pair = -1.0;  // worst row errors against fp64 (fp32_selfcheck)
  // [linear rows, quadratics, cones] + ONE LMI the lane kernels do not hold: the lane kernel evaluates everything but the
  // LMI (generic image built with skip_lmi), the workgroup-per-sample kernel the LMI on top of that (rayen_abi.hip)
  bool mixed32 = false, mixed64 = false;
  int64_t device_bytes = 0;
};

namespace rayen {

// products epilogue for wide sets (rayen_wide.hip)
int wide_build(const RayenPack* p, WideImage** out, int64_t* bytes);
void wide_free(WideImage* img);
template <typename T>
int wide_epilogue(const RayenPack* p, const WideImage* img, const T* Tm, int64_t ldt, const T* v, int64_t B, int64_t ldv,
                  T* y, int64_t ldy, T* kappa, int32_t* active, int32_t* nan_flag, hipStream_t stream);

template <typename T>
int wide_bwd_coefficients(const RayenPack* p, const WideImage* img, const T* Tm, int64_t ldt, const T* v, int64_t B,
                          int64_t ldv, const T* kappa, const int32_t* active, const T* gy, int64_t ldg, T* C, int64_t ldc,
                          T* gs, hipStream_t stream);

// SIMDs the persistent grids may fill: all of the device's minus rayen_reserve_cus() compute units (a collective that
// runs beside the projection -- RCCL's all-gather kernels in the multi-GPU step -- needs CUs of its own)
int launch_simds(int n_simd);

// generic path (rayen_generic.hip)
template <typename T>
int generic_build(const RayenPack* p, GenericImage<T>* img);
template <typename T>
void generic_free(GenericImage<T>* img);
template <typename T>
int generic_block_for(const RayenPack* p, const GenericImage<T>& img);
template <typename T>
int generic_forward(const RayenPack* p, const GenericImage<T>& img, const T* v, int64_t B,
                    int64_t ldv, T* y, int64_t ldy, T* kappa, int32_t* active, int32_t* nan_flag,
                    int old_mode, hipStream_t stream, int64_t ldk = 1);
template <typename T>
bool generic_holds_lmis(const RayenPack* p);
template <typename T>
bool generic_backward_serves(const RayenPack* p, const GenericImage<T>& img);
template <typename T>
int generic_backward(const RayenPack* p, const GenericImage<T>& img, const T* v, int64_t B,
                     int64_t ldv, const T* kappa, const int32_t* active, const T* grad_y,
                     int64_t ldg, T* grad_v, int64_t ldgv, int old_mode, hipStream_t stream);

// fp32 MFMA path (rayen_mfma.hip)
bool mfma_eligible(const RayenPack* p);
int mfma_build(const RayenPack* p, MfmaImage** out, int64_t* bytes);
void mfma_free(MfmaImage* img);
int mfma_forward(const RayenPack* p, const MfmaImage* img, const float* v, int64_t B, int64_t ldv,
                 float* y, int64_t ldy, float* kappa, int32_t* active, int32_t* nan_flag,
                 int old_mode, hipStream_t stream);

// fp32 results on split bf16 operands (rayen_mfma_split.hip)
bool mfma_split_eligible(const RayenPack* p);
int mfma_split_build(const RayenPack* p, SplitImage** out, int64_t* bytes);
void mfma_split_free(SplitImage* img);
int mfma_split_forward(const RayenPack* p, const SplitImage* img, const float* v, int64_t B, int64_t ldv,
                       float* y, int64_t ldy, float* kappa, int32_t* active, int32_t* nan_flag,
                       hipStream_t stream);

// fp32 results on pairs of f16 operands (rayen_mfma_pair.hip); eligibility is mfma_split_eligible's
int mfma_pair_build(const RayenPack* p, PairImage** out, int64_t* bytes);
int mfma_pair_build_dense(const RayenPack* p, PairImage** out, int64_t* bytes);   // without shared tiles (fused mapper)
bool mfma_pair_has_halves(const PairImage* img);
void mfma_pair_free(PairImage* img);
int mfma_pair_forward(const RayenPack* p, const PairImage* img, const float* v, int64_t B, int64_t ldv,
                      float* y, int64_t ldy, float* kappa, int32_t* active, int32_t* nan_flag,
                      hipStream_t stream);

// the same arithmetic with the rows of v and y trickled through LDS under the tile walk (rayen_mfma_pair_io.hip)
bool mfma_pair_io_serves(const RayenPack* p, const PairImage* img, const float* v, int64_t B, int64_t ldv,
                         const float* y, int64_t ldy);
int mfma_pair_io_prepare(const RayenPack* p, const PairImage* img);   // function attributes (pack creation only)
int mfma_pair_io_forward(const RayenPack* p, const PairImage* img, const float* v, int64_t B, int64_t ldv,
                         float* y, int64_t ldy, float* kappa, int32_t* active, int32_t* nan_flag,
                         hipStream_t stream);

// the same arithmetic with the image of W resident in LDS, rows straight from / to memory one group ahead, three waves per
// SIMD on groups of 32 samples (rayen_mfma_pair_wl.hip, round 6)
bool mfma_pair_wl_serves(const RayenPack* p, const PairImage* img, const float* v, int64_t B, int64_t ldv,
                         const float* y, int64_t ldy);
int mfma_pair_wl_prepare(const RayenPack* p, PairImage* img);   // function attributes (pack creation only)
bool mfma_pair_wl_serves_mapped(const RayenPack* p, const PairImage* img, const float* x, int64_t B, int64_t ldx, int in_dim,
                                const float* v_out, int64_t ldvo, const float* y, int64_t ldy);
int mfma_pair_wl_forward_mapped(const RayenPack* p, const PairImage* img, const float* x, int64_t B, int64_t ldx, int in_dim,
                                const void* image, float* v_out, int64_t ldvo, float* y, int64_t ldy, float* kappa,
                                int32_t* active, int32_t* nan_flag, hipStream_t stream);
int mfma_pair_wl_forward(const RayenPack* p, const PairImage* img, const float* v, int64_t B, int64_t ldv,
                         float* y, int64_t ldy, float* kappa, int32_t* active, int32_t* nan_flag,
                         hipStream_t stream);

// the same arithmetic, W-stationary: the tiles of W in the registers of a workgroup's eight waves, the batch through an
// LDS image (rayen_mfma_pair_ws8.hip)
int mfma_pair_ws8_build(const RayenPack* p, const PairImage* img, Ws8Image** out);   // *out = null: not served
void mfma_pair_ws8_free(Ws8Image* ws);
bool mfma_pair_ws8_serves(const RayenPack* p, const PairImage* img, const Ws8Image* ws, const float* v, int64_t B,
                          int64_t ldv, const float* y, int64_t ldy);
int mfma_pair_ws8_forward(const RayenPack* p, const PairImage* img, const Ws8Image* ws, const float* v, int64_t B,
                          int64_t ldv, float* y, int64_t ldy, float* kappa, int32_t* active, int32_t* nan_flag,
                          hipStream_t stream);

int64_t mfma_pair_mapper_image_bytes(const RayenPack* p, const PairImage* img, int in_dim);
int mfma_pair_mapper_prepare(const RayenPack* p, const PairImage* img, const float* w, int64_t ldw, int in_dim,
                             const float* bias, void* image, hipStream_t stream);
int mfma_pair_forward_mapped(const RayenPack* p, const PairImage* img, const float* x, int64_t B, int64_t ldx,
                             int in_dim, const void* image, float* v_out, int64_t ldvo, float* y, int64_t ldy,
                             float* kappa, int32_t* active, int32_t* nan_flag, hipStream_t stream);

// the bf16-triple kernel behind the module's mapper v = Wm x + b; Wm as a caller-owned split-operand image
int64_t mfma_split_mapper_image_bytes(const RayenPack* p, const SplitImage* img, int in_dim);
int mfma_split_mapper_prepare(const RayenPack* p, const SplitImage* img, const float* w, int64_t ldw, int in_dim,
                              const float* bias, void* image, hipStream_t stream);
int mfma_split_forward_mapped(const RayenPack* p, const SplitImage* img, const float* x, int64_t B, int64_t ldx,
                              int in_dim, const void* image, float* v_out, int64_t ldvo, float* y, int64_t ldy,
                              float* kappa, int32_t* active, int32_t* nan_flag, hipStream_t stream);

// fp32 MFMA path with the mapper v = Wm x + b fused in front (rayen_mfma_mapped.hip)
bool mfma_mapper_fusable(const RayenPack* p, const MfmaImage* img, int in_dim);
int mfma_forward_mapped(const RayenPack* p, const MfmaImage* img, const float* x, int64_t B, int64_t ldx,
                        int in_dim, const float* w, int64_t ldw, const float* bias, float* v_out,
                        int64_t ldvo, float* y, int64_t ldy, float* kappa, int32_t* active,
                        int32_t* nan_flag, hipStream_t stream);

// fp32 MFMA backward (rayen_mfma_bwd.hip)
bool mfma_bwd_eligible(const RayenPack* p);
int mfma_bwd_build(const RayenPack* p, MfmaBwdImage** out, int64_t* bytes);
void mfma_bwd_free(MfmaBwdImage* img);
int64_t mfma_bwd_workspace_bytes(const RayenPack* p, const MfmaBwdImage* img, int64_t B);
int mfma_backward(const RayenPack* p, const MfmaBwdImage* img, const float* v, int64_t B, int64_t ldv,
                  const float* kappa, const int32_t* active, const float* grad_y, int64_t ldg, float* grad_v,
                  int64_t ldgv, int old_mode, void* workspace, int64_t workspace_bytes, hipStream_t stream);

// four-lanes-per-sample LMI kernels (rayen_lmi_quad32.hip / rayen_lmi_quad64.hip)
bool lmi_quad_eligible_f32(const RayenPack* p);
bool lmi_quad_eligible_f64(const RayenPack* p);
int lmi_quad_build_f32(const RayenPack* p, LmiQuadImage** out, int64_t* bytes);
int lmi_quad_build_f64(const RayenPack* p, LmiQuadImage** out, int64_t* bytes);
void lmi_quad_free(LmiQuadImage* img);
int lmi_quad_forward_f32(const RayenPack* p, const LmiQuadImage* img, const float* v, int64_t B, int64_t ldv,
                         float* y, int64_t ldy, float* kappa, int32_t* active, int32_t* nan_flag,
                         hipStream_t stream);
int lmi_quad_forward_f64(const RayenPack* p, const LmiQuadImage* img, const double* v, int64_t B, int64_t ldv,
                         double* y, int64_t ldy, double* kappa, int32_t* active, int32_t* nan_flag,
                         hipStream_t stream);

// fp32 MFMA backward for sets with equalities / packed low-rank quadratics (rayen_mfma_bwdg.hip)
// the same shapes with every quadratic in packed tiles, n <= 32: backward on f16 pairs (rayen_mfma_bwdp.hip)
// fp32 backward of dense-form sets at n = k = 64 on f16 pairs, forms resident in LDS, unbucketed (rayen_mfma_bwdd.hip, round 6)
bool mfma_bwdd_eligible(const RayenPack* p);
int mfma_bwdd_build(const RayenPack* p, MfmaBwddImage** out, int64_t* bytes);
void mfma_bwdd_free(MfmaBwddImage* img);
bool mfma_bwdd_serves(const RayenPack* p, const MfmaBwddImage* img, const float* v, int64_t B, int64_t ldv,
                      const float* gy, int64_t ldg, const float* gv, int64_t ldgv);
int mfma_bwdd_backward(const RayenPack* p, const MfmaBwddImage* img, const float* v, int64_t B, int64_t ldv,
                       const float* kappa, const int32_t* active, const float* grad_y, int64_t ldg, float* grad_v,
                       int64_t ldgv, hipStream_t stream);
bool mfma_bwdp_eligible(const RayenPack* p);
int mfma_bwdp_build(const RayenPack* p, MfmaBwdpImage** out, int64_t* bytes);
void mfma_bwdp_free(MfmaBwdpImage* img);
int mfma_bwdp_backward(const RayenPack* p, const MfmaBwdpImage* img, const float* v, int64_t B, int64_t ldv,
                       const float* kappa, const int32_t* active, const float* grad_y, int64_t ldg,
                       float* grad_v, int64_t ldgv, hipStream_t stream);
bool mfma_bwdg_eligible(const RayenPack* p);
int mfma_bwdg_build(const RayenPack* p, MfmaBwdgImage** out, int64_t* bytes);
void mfma_bwdg_free(MfmaBwdgImage* img);
int64_t mfma_bwdg_workspace_bytes(const RayenPack* p, const MfmaBwdgImage* img, int64_t B);
int mfma_bwdg_backward(const RayenPack* p, const MfmaBwdgImage* img, const float* v, int64_t B, int64_t ldv,
                       const float* kappa, const int32_t* active, const float* grad_y, int64_t ldg,
                       float* grad_v, int64_t ldgv, int old_mode, void* workspace, int64_t workspace_bytes,
                       hipStream_t stream);

// its fp64 twin (rayen_mfma_bwdg64.hip)
bool mfma64_bwdg_eligible(const RayenPack* p);
int mfma64_bwdg_build(const RayenPack* p, Mfma64BwdgImage** out, int64_t* bytes);
void mfma64_bwdg_free(Mfma64BwdgImage* img);
int mfma64_bwdg_backward(const RayenPack* p, const Mfma64BwdgImage* img, const double* v, int64_t B, int64_t ldv,
                         const double* kappa, const int32_t* active, const double* grad_y, int64_t ldg,
                         double* grad_v, int64_t ldgv, int old_mode, hipStream_t stream);

// one wave per sample, the matrix in LDS: sets = [linear rows] + one LMI of any size the LDS holds (rayen_lmi_wave.h)
bool lmi_wave_eligible_f32(const RayenPack* p);
bool lmi_wave_eligible_f64(const RayenPack* p);
int lmi_wave_build_f32(const RayenPack* p, LmiWaveImage** out, int64_t* bytes);
int lmi_wave_build_f64(const RayenPack* p, LmiWaveImage** out, int64_t* bytes);
void lmi_wave_free(LmiWaveImage* img);
int lmi_wave_forward_f32(const RayenPack* p, const LmiWaveImage* img, const float* v, int64_t B, int64_t ldv, float* y,
                         int64_t ldy, float* kappa, int32_t* active, int32_t* nan_flag, hipStream_t stream);
int lmi_wave_forward_f64(const RayenPack* p, const LmiWaveImage* img, const double* v, int64_t B, int64_t ldv, double* y,
                         int64_t ldy, double* kappa, int32_t* active, int32_t* nan_flag, hipStream_t stream);
int lmi_wave_backward_f32(const RayenPack* p, const LmiWaveImage* img, const float* v, int64_t B, int64_t ldv,
                          const float* kappa, const int32_t* active, const float* grad_y, int64_t ldg, float* grad_v,
                          int64_t ldgv, hipStream_t stream);
int lmi_wave_backward_f64(const RayenPack* p, const LmiWaveImage* img, const double* v, int64_t B, int64_t ldv,
                          const double* kappa, const int32_t* active, const double* grad_y, int64_t ldg, double* grad_v,
                          int64_t ldgv, hipStream_t stream);

// one workgroup per sample, the packed lower triangle in LDS (rayen_lmi_block.h): forward and backward for r up to ~280 (fp32) / ~197 (fp64)
// on the wave kernel's image
bool lmi_block_eligible_f32(const RayenPack* p);
bool lmi_block_eligible_f64(const RayenPack* p);
bool lmi_block_serves_f32(const LmiWaveImage* img);
bool lmi_block_serves_f64(const LmiWaveImage* img);
int lmi_block_prepare_f32(const LmiWaveImage* img);
int lmi_block_prepare_f64(const LmiWaveImage* img);
bool lmi_block_eligible_mixed_f32(const RayenPack* p);     // ... with quadratics / cones next to the LMI
bool lmi_block_eligible_mixed_f64(const RayenPack* p);
int lmi_block_forward_f32(const RayenPack* p, const LmiWaveImage* img, const float* v, int64_t B, int64_t ldv, float* y,
                          int64_t ldy, float* kappa, int32_t* active, int32_t* nan_flag, hipStream_t stream,
                          const float* kappa_in = nullptr, int64_t ldk_in = 1, int old_mode = 0);
int lmi_block_forward_f64(const RayenPack* p, const LmiWaveImage* img, const double* v, int64_t B, int64_t ldv, double* y,
                          int64_t ldy, double* kappa, int32_t* active, int32_t* nan_flag, hipStream_t stream,
                          const double* kappa_in = nullptr, int64_t ldk_in = 1, int old_mode = 0);
bool lmi_block_bwd_serves_f32(const LmiWaveImage* img);
bool lmi_block_bwd_serves_f64(const LmiWaveImage* img);
int lmi_block_backward_f32(const RayenPack* p, const LmiWaveImage* img, const float* v, int64_t B, int64_t ldv,
                           const float* kappa, const int32_t* active, const float* grad_y, int64_t ldg, float* grad_v,
                           int64_t ldgv, hipStream_t stream, int only_lmi = 0, int old_mode = 0);
int lmi_block_backward_f64(const RayenPack* p, const LmiWaveImage* img, const double* v, int64_t B, int64_t ldv,
                           const double* kappa, const int32_t* active, const double* grad_y, int64_t ldg, double* grad_v,
                           int64_t ldgv, hipStream_t stream, int only_lmi = 0, int old_mode = 0);
// ... from the products T = v W_ext' of a library GEMM (sets with many generators): forward, and the backward's coefficients
bool lmi_block_products_serves_f32(const LmiWaveImage* img);
bool lmi_block_products_serves_f64(const LmiWaveImage* img);
int lmi_block_forward_products_f32(const RayenPack* p, const LmiWaveImage* img, const float* prods, int64_t ldt, const float* v,
                                   int64_t B, int64_t ldv, float* y, int64_t ldy, float* kappa, int32_t* active,
                                   int32_t* nan_flag, hipStream_t stream);
int lmi_block_forward_products_f64(const RayenPack* p, const LmiWaveImage* img, const double* prods, int64_t ldt, const double* v,
                                   int64_t B, int64_t ldv, double* y, int64_t ldy, double* kappa, int32_t* active,
                                   int32_t* nan_flag, hipStream_t stream);
int lmi_block_bwd_coefficients_f32(const RayenPack* p, const LmiWaveImage* img, const float* prods, int64_t ldt, const float* v,
                                   int64_t B, int64_t ldv, const float* kappa, const int32_t* active, const float* grad_y,
                                   int64_t ldg, float* C, int64_t ldc, float* gs, hipStream_t stream);
int lmi_block_bwd_coefficients_f64(const RayenPack* p, const LmiWaveImage* img, const double* prods, int64_t ldt, const double* v,
                                   int64_t B, int64_t ldv, const double* kappa, const int32_t* active, const double* grad_y,
                                   int64_t ldg, double* C, int64_t ldc, double* gs, hipStream_t stream);
bool lmi_wave_serves_f32(const LmiWaveImage* img);      // (the image may exist for the block kernel alone)
bool lmi_wave_serves_f64(const LmiWaveImage* img);

bool lmi_quad_bwd_serves_f32(const RayenPack* p, const LmiQuadImage* img);
bool lmi_quad_bwd_serves_f64(const RayenPack* p, const LmiQuadImage* img);
int lmi_quad_backward_f32(const RayenPack* p, const LmiQuadImage* img, const float* v, int64_t B, int64_t ldv,
                          const float* kappa, const int32_t* active, const float* grad_y, int64_t ldg,
                          float* grad_v, int64_t ldgv, hipStream_t stream);
int lmi_quad_backward_f64(const RayenPack* p, const LmiQuadImage* img, const double* v, int64_t B, int64_t ldv,
                          const double* kappa, const int32_t* active, const double* grad_y, int64_t ldg,
                          double* grad_v, int64_t ldgv, hipStream_t stream);

// fp64 MFMA backward (rayen_mfma_bwd64.hip)
bool mfma64_bwd_eligible(const RayenPack* p);
int mfma64_bwd_build(const RayenPack* p, Mfma64BwdImage** out, int64_t* bytes);
void mfma64_bwd_free(Mfma64BwdImage* img);
int64_t mfma64_bwd_workspace_bytes(const RayenPack* p, const Mfma64BwdImage* img, int64_t B);
int mfma64_backward(const RayenPack* p, const Mfma64BwdImage* img, const double* v, int64_t B, int64_t ldv,
                    const double* kappa, const int32_t* active, const double* grad_y, int64_t ldg,
                    double* grad_v, int64_t ldgv, int old_mode, void* workspace, int64_t workspace_bytes,
                    hipStream_t stream);

// fp64 MFMA path (rayen_mfma_f64.hip)
bool mfma64_eligible(const RayenPack* p);
int mfma64_build(const RayenPack* p, Mfma64Image** out, int64_t* bytes);
void mfma64_free(Mfma64Image* img);
int mfma64_forward(const RayenPack* p, const Mfma64Image* img, const double* v, int64_t B, int64_t ldv,
                   double* y, int64_t ldy, double* kappa, int32_t* active, int32_t* nan_flag,
                   int old_mode, hipStream_t stream);

}  // namespace rayen
