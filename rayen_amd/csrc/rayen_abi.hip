// C-ABI entry points of librayen_hip.so (declared in include/rayen_hip.h).
#include "rayen_internal.h"

#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <vector>

using namespace rayen;

namespace {

bool device_is_gfx950(int dev) {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
  return std::strncmp(prop.gcnArchName, "gfx950", 6) == 0;
}

int check_table(const RayenPackDesc* d) {
  if (d->k <= 0 || d->n <= 0 || d->n > d->k || d->n_rows < 0 || d->n_segments < 0) return RAYEN_E_BAD_ARG;
  if (d->n_rows > 0 && d->W == nullptr) return RAYEN_E_BAD_ARG;
  if (d->n_segments > 0 && d->segments == nullptr) return RAYEN_E_BAD_ARG;
  if (d->y0 == nullptr) return RAYEN_E_BAD_ARG;
  if (!d->out_identity && d->NA_E == nullptr) return RAYEN_E_BAD_ARG;
  if (d->out_identity && d->k != d->n) return RAYEN_E_BAD_ARG;
  for (int s = 0; s < d->n_segments; ++s) {
    const RayenSegment& g = d->segments[s];
    if (g.type < RAYEN_SEG_LIN || g.type > RAYEN_SEG_LMI) return RAYEN_E_BAD_ARG;
    if (g.row0 < 0 || g.nrows < 0 || g.row0 + g.nrows > d->n_rows) return RAYEN_E_BAD_ARG;
    const int aux = (g.type == RAYEN_SEG_QUAD_SYM || g.type == RAYEN_SEG_QUAD_FAC) ? 1
                    : (g.type == RAYEN_SEG_SOC) ? 2 : 0;
    if (aux && (g.aux_row < 0 || g.aux_row + aux > d->n_rows)) return RAYEN_E_BAD_ARG;
    if (g.type == RAYEN_SEG_QUAD_SYM && g.nrows != d->n) return RAYEN_E_BAD_ARG;
    if (g.type == RAYEN_SEG_LMI && (g.dim <= 0 || g.nrows != g.dim * (g.dim + 1) / 2)) return RAYEN_E_BAD_ARG;
  }
  return RAYEN_OK;
}

template <typename T> GenericImage<T>& image_of(const RayenPack* p);
template <> GenericImage<float>& image_of<float>(const RayenPack* p) { return p->g32; }
template <> GenericImage<double>& image_of<double>(const RayenPack* p) { return p->g64; }

// which kernel family served this thread's most recent forward call (rayen_last_forward_kernel)
thread_local int g_last_forward = RAYEN_KERNEL_NONE;

// Schedules of the f16-pair forward (same arithmetic): 3 (default, round 6) = the image of W resident in LDS
// (rayen_mfma_pair_wl.hip) where the pack and the call allow it, else as 1 | 1 (the default of rounds 3-5) = rows of v and y
// trickled through LDS under the tile walk (rayen_mfma_pair_io.hip) where the call's shape allows it, the W-stationary
// kernel for mid-size batches | 0 = rayen_mfma_pair.hip always | 2 = W-stationary (rayen_mfma_pair_ws8.hip) where the pack
// and the call allow it, else as 1.  All bit-identical.  RAYEN_PAIR_IO / rayen_pair_schedule select (A/B runs).
std::atomic<int>& pair_schedule_cell() {
  static std::atomic<int> mode([] {
    const char* e = std::getenv("RAYEN_PAIR_IO");
    return (e != nullptr && e[0] >= '0' && e[0] <= '3') ? e[0] - '0' : 3;
  }());
  return mode;
}
int pair_schedule() { return pair_schedule_cell().load(std::memory_order_relaxed); }

// compute units left free by the persistent grids (rayen_reserve_cus; RAYEN_RESERVE_CUS sets the initial value)
std::atomic<int>& reserved_cus_cell() {
  static std::atomic<int> cus([] {
    const char* e = std::getenv("RAYEN_RESERVE_CUS");
    const int v = e != nullptr ? std::atoi(e) : 0;
    return v < 0 ? 0 : (v > 128 ? 128 : v);
  }());
  return cus;
}

int check_device(const RayenPack* p) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) return RAYEN_E_NO_DEVICE;
  return dev == p->device ? RAYEN_OK : RAYEN_E_DEVICE_MISMATCH;
}

// Every device image a pack may ever need is built by rayen_pack_create (build_images below), so the entry points
// only read the pack: no lazy state, no lock, no allocation, nothing on the null stream after creation.
template <typename T>
int check_ready(const RayenPack* p, bool backward) {
  const int rc = check_device(p);
  if (rc) return rc;
  const int want = (sizeof(T) == 4 ? RAYEN_PREPARE_F32 : RAYEN_PREPARE_F64) | (backward ? 4 : 0);
  return (p->prepared & want) == want ? RAYEN_OK : RAYEN_E_NOT_PREPARED;
}

template <typename T>
int build_generic(RayenPack* p) {
  GenericImage<T>& img = image_of<T>(p);
  const int rc = generic_build<T>(p, &img);
  if (rc != RAYEN_OK) { generic_free<T>(&img); return rc; }
  p->device_bytes += img.bytes;
  return RAYEN_OK;
}

template <typename Image, typename Build>
int build_one(RayenPack* p, bool eligible, Image** slot, Build build) {
  if (!eligible) return RAYEN_OK;
  int64_t bytes = 0;
  Image* img = nullptr;
  const int rc = build(p, &img, &bytes);
  if (rc != RAYEN_OK) return rc;
  *slot = img;
  p->device_bytes += bytes;
  return RAYEN_OK;
}

// Which LMI kernel takes a matrix neither the quad nor the lane kernels hold: the workgroup-per-sample kernels
// (rayen_lmi_block.h) wherever they serve in fp32, beyond 44 x 44 in fp64, and wherever the wave kernel's full storage does
// not fit.  Measured with warm clocks, B = 2 000, forward / backward ms, block against wave (profiles/bench/r05_lmi_cross.txt):
// fp32 r = 33 0.086 / 0.114 against 0.105 / 0.129, 64 0.24 / 0.30 against 0.29 / 0.35, 80 0.43 / 0.51 against 1.01 / 1.21;
// fp64 r = 40 0.193 / 0.190 against 0.178 / 0.196, 50 0.27 / 0.27 against 0.45 / 0.50.
bool lmi_block_preferred(bool block_serves, bool wave_serves, int r, bool f64) {
  if (!block_serves) return false;
  if (!wave_serves) return true;
  const char* env = std::getenv("RAYEN_LMI_BLOCK");       // 0 / 1 pin a kernel (developer A/B, tests)
  if (env != nullptr && (env[0] == '0' || env[0] == '1')) return env[0] == '1';
  return f64 ? r > 44 : true;
}

int lmi_dim(const RayenPack* p) {
  int r = 0;
  for (const RayenSegment& g : p->segs)
    if (g.type == RAYEN_SEG_LMI && g.dim > r) r = g.dim;
  return r;
}

int build_images(RayenPack* p, int prepare) {
  // (no precision bit = both precisions: RAYEN_PREPARE_FWD_ONLY alone means "forward only, fp32 and fp64")
  const bool f32 = (prepare & (RAYEN_PREPARE_F32 | RAYEN_PREPARE_F64)) == 0 || (prepare & RAYEN_PREPARE_F32);
  const bool f64 = (prepare & (RAYEN_PREPARE_F32 | RAYEN_PREPARE_F64)) == 0 || (prepare & RAYEN_PREPARE_F64);
  const bool bwd = !(prepare & RAYEN_PREPARE_FWD_ONLY);
  int rc = RAYEN_OK;
  if ((rc = build_one(p, true, &p->wide, wide_build))) return rc;
  // The inward bias (RAYEN_PREPARE_INWARD_BIAS / RAYEN_INWARD_BIAS, SURVEY.md section 7 "fp32 feasibility"): every term of
  // kappa is positively homogeneous of degree 1 in the rows of W, so the fp32 images built from (1 + eps) W evaluate
  // (1 + eps) kappa -- a step 1 / max(1, (1 + eps) kappa) that stops eps short of the boundary the fp32 arithmetic would
  // otherwise land ON (half of its roundings outside).  No kernel changes, the backward differentiates the same biased
  // function, interior samples (kappa <= 1 / (1 + eps)) are untouched, and the fp64 images keep the exact W.
  struct Restore {
    RayenPack* p; std::vector<double> keep;
    ~Restore() { if (!keep.empty()) p->W.swap(keep); }
  } restore{p, {}};
  if (f32 && p->inward_bias > 0.0) {
    restore.keep = p->W;
    for (double& w : p->W) w *= 1.0 + p->inward_bias;
    // (a symmetric form enters as sqrt(v'G v): ITS rows take the factor twice, or phi . v + sqrt(..) with phi . v < 0 --
    // most directions -- would move the other way; factors U, cone rows, LMI generators and linear rows enter linearly)
    for (const RayenSegment& g : p->segs)
      if (g.type == RAYEN_SEG_QUAD_SYM)
        for (size_t i = (size_t)g.row0 * p->n; i < (size_t)(g.row0 + g.nrows) * p->n; ++i) p->W[i] *= 1.0 + p->inward_bias;
  }
  if (f32) {
    p->prepared |= RAYEN_PREPARE_F32;
    p->mixed32 = lmi_block_eligible_mixed_f32(p) && !generic_holds_lmis<float>(p);
    p->g32.skip_lmi = p->mixed32;
    if ((rc = build_generic<float>(p))) return rc;
    if ((rc = build_one(p, mfma_eligible(p), &p->m32, mfma_build))) return rc;
    {
      const int mode = p->fp32_mode;
      const bool split_ok = mode != 1 && mfma_split_eligible(p);
      if ((rc = build_one(p, split_ok && mode != 3, &p->sp32, mfma_split_build))) return rc;
      if ((rc = build_one(p, split_ok && (mode == 0 || mode == 3), &p->pr32, mfma_pair_build))) return rc;
      if (p->pr32 != nullptr && (rc = mfma_pair_io_prepare(p, p->pr32))) return rc;
      if (p->pr32 != nullptr && (rc = mfma_pair_wl_prepare(p, p->pr32))) return rc;
      if (p->pr32 != nullptr && (rc = mfma_pair_ws8_build(p, p->pr32, &p->ws8_32))) return rc;
      // (the instances behind the fused mapper walk an image without shared tiles, rayen_mfma_pair.hip)
      if (p->pr32 != nullptr && mfma_pair_has_halves(p->pr32) && (rc = build_one(p, true, &p->pr32m, mfma_pair_build_dense))) return rc;
    }
    if ((rc = build_one(p, lmi_quad_eligible_f32(p), &p->q32, lmi_quad_build_f32))) return rc;
    // (the wave-per-sample LMI kernels take what neither the quad kernel nor the lane kernels hold: matrices beyond ~30 x 30)
    if ((rc = build_one(p, (lmi_wave_eligible_f32(p) || lmi_block_eligible_f32(p) || p->mixed32) && (p->q32 == nullptr || lmi_dim(p) > 28), &p->w32, lmi_wave_build_f32))) return rc;
    if (p->w32 != nullptr && (rc = lmi_block_prepare_f32(p->w32))) return rc;
    if (bwd) {
      if ((rc = build_one(p, mfma_bwd_eligible(p), &p->mb32, mfma_bwd_build))) return rc;
      // (f16 pairs in the backward follow the forward's switch: fp32_mode 0 measured, 3 unmeasured, 1 / 2 / 4 never)
      if (p->mb32 != nullptr && (p->fp32_mode == 0 || p->fp32_mode == 3) &&
          (rc = build_one(p, mfma_bwdd_eligible(p), &p->mbd32, mfma_bwdd_build)))
        return rc;
      if (p->mb32 == nullptr && (rc = build_one(p, mfma_bwdg_eligible(p), &p->mbg32, mfma_bwdg_build))) return rc;
      // (f16 pairs in the backward follow the forward's switch: fp32_mode 0 measured, 3 unmeasured, 1 / 2 / 4 never)
      if (p->mbg32 != nullptr && (p->fp32_mode == 0 || p->fp32_mode == 3) &&
          (rc = build_one(p, mfma_bwdp_eligible(p), &p->mbp32, mfma_bwdp_build)))
        return rc;
    }
  }
  if (!restore.keep.empty()) { p->W.swap(restore.keep); restore.keep.clear(); }     // (the exact rows again)
  if (f64) {
    p->prepared |= RAYEN_PREPARE_F64;
    p->mixed64 = lmi_block_eligible_mixed_f64(p) && !generic_holds_lmis<double>(p);
    p->g64.skip_lmi = p->mixed64;
    if ((rc = build_generic<double>(p))) return rc;
    if ((rc = build_one(p, mfma64_eligible(p), &p->m64, mfma64_build))) return rc;
    if ((rc = build_one(p, lmi_quad_eligible_f64(p), &p->q64, lmi_quad_build_f64))) return rc;
    if ((rc = build_one(p, (lmi_wave_eligible_f64(p) || lmi_block_eligible_f64(p) || p->mixed64) && (p->q64 == nullptr || lmi_dim(p) > 20), &p->w64, lmi_wave_build_f64))) return rc;
    if (p->w64 != nullptr && (rc = lmi_block_prepare_f64(p->w64))) return rc;
    if (bwd) {
      if ((rc = build_one(p, mfma64_bwd_eligible(p), &p->mb64, mfma64_bwd_build))) return rc;
      if (p->mb64 == nullptr && (rc = build_one(p, mfma64_bwdg_eligible(p), &p->mbg64, mfma64_bwdg_build))) return rc;
    }
  }
  if (bwd) p->prepared |= 4;
  return RAYEN_OK;
}

// Sets with quadratics / cones NEXT TO an LMI the lane kernels do not hold (beyond ~30 x 30; until round 5 these left the C
// ABI for the device's libraries): two launches.  The lane-per-sample kernel walks an image whose LMI segment is empty
// (GenericImage::skip_lmi) and leaves, per sample, the maximum over everything else -- in `kappa`, or in column 0 of `y`
// when the caller wants no kappa -- and its row in `active`; the workgroup-per-sample kernel (rayen_lmi_block.h) evaluates
// the LMI (and the linear rows once more), takes the larger of the two and writes y.  The backward is the lane kernel's for
// every sample (a sample whose active row is the LMI gets s N'g from it) followed by the workgroup kernel on the samples
// the LMI clipped.
template <typename T> bool is_mixed(const RayenPack* p) { return sizeof(T) == 4 ? p->mixed32 : p->mixed64; }

template <typename T>
int mixed_forward(const RayenPack* p, const T* v, int64_t B, int64_t ldv, T* y, int64_t ldy, T* kappa, int32_t* active,
                  int32_t* nan_flag, int old_mode, hipStream_t stream) {
  if (y == nullptr) return RAYEN_E_UNSUPPORTED;
  // (RAYEN_old: kappa is the same, so the lane kernel runs WITHOUT the head -- it writes no y here -- and the workgroup
  // kernel takes the step 1 / (||v|| e^beta + kappa))
  T* between = kappa != nullptr ? kappa : y;
  const int64_t ldk = kappa != nullptr ? 1 : ldy;
  const int rc = generic_forward<T>(p, image_of<T>(p), v, B, ldv, static_cast<T*>(nullptr), 0, between, active, nullptr, 0,
                                    stream, ldk);
  if (rc) return rc;
  g_last_forward = RAYEN_KERNEL_LMI_BLOCK;
  if constexpr (sizeof(T) == 4)
    return lmi_block_forward_f32(p, p->w32, v, B, ldv, y, ldy, kappa, active, nan_flag, stream, between, ldk, old_mode);
  else
    return lmi_block_forward_f64(p, p->w64, v, B, ldv, y, ldy, kappa, active, nan_flag, stream, between, ldk, old_mode);
}

template <typename T>
int project_generic(const RayenPack* p, const T* v, int64_t B, int64_t ldv, T* y, int64_t ldy, T* kappa,
                    int32_t* active, int32_t* nan_flag, void* stream, int old_mode = 0) {
  if (p == nullptr || B < 0 || (B > 0 && v == nullptr) || ldv < p->n + old_mode ||
      (y != nullptr && ldy < p->k))
    return RAYEN_E_BAD_ARG;
  int rc = check_ready<T>(p, false);
  if (rc) return rc;
  if (is_mixed<T>(p)) return mixed_forward<T>(p, v, B, ldv, y, ldy, kappa, active, nan_flag, old_mode, static_cast<hipStream_t>(stream));
  return generic_forward<T>(p, image_of<T>(p), v, B, ldv, y, ldy, kappa, active, nan_flag, old_mode,
                            static_cast<hipStream_t>(stream));
}

// developer A/B: RAYEN_BWD_DENSE_PAIRS=0 keeps config-3-like packs on the bucketed exact-fp32 backward (rayen_mfma_bwd.hip)
static bool dense_pairs_backward_enabled() {
  static const bool on = [] { const char* e = std::getenv("RAYEN_BWD_DENSE_PAIRS"); return !(e != nullptr && e[0] == '0'); }();
  return on;
}

template <typename T>
int project_bwd(const RayenPack* p, const T* v, int64_t B, int64_t ldv, const T* kappa,
                const int32_t* active, const T* grad_y, int64_t ldg, T* grad_v, int64_t ldgv, void* stream,
                int old_mode = 0, bool force_generic = false, void* workspace = nullptr, int64_t workspace_bytes = 0) {
  if (p == nullptr || B < 0 || ldv < p->n + old_mode || ldg < p->k || ldgv < p->n + old_mode)
    return RAYEN_E_BAD_ARG;
  if (B > 0 && (!v || !kappa || !active || !grad_y || !grad_v)) return RAYEN_E_BAD_ARG;
  int rc = check_ready<T>(p, true);
  if (rc) return rc;
  if constexpr (sizeof(T) == 4) {
    if (!force_generic && !old_mode) {
      if (p->q32 != nullptr && lmi_quad_bwd_serves_f32(p, p->q32))
        return lmi_quad_backward_f32(p, p->q32, v, B, ldv, kappa, active, grad_y, ldg, grad_v, ldgv,
                                     static_cast<hipStream_t>(stream));
    }
    if (!force_generic) {
      if (p->mbd32 != nullptr && p->mbd32_state == 1 && !old_mode && dense_pairs_backward_enabled() &&
          mfma_bwdd_serves(p, p->mbd32, v, B, ldv, grad_y, ldg, grad_v, ldgv))
        return mfma_bwdd_backward(p, p->mbd32, v, B, ldv, kappa, active, grad_y, ldg, grad_v, ldgv,
                                  static_cast<hipStream_t>(stream));
      if (p->mb32 != nullptr)
        return mfma_backward(p, p->mb32, v, B, ldv, kappa, active, grad_y, ldg, grad_v, ldgv, old_mode, workspace,
                             workspace_bytes, static_cast<hipStream_t>(stream));
      if (p->mbp32 != nullptr && p->mbp32_state == 1 && !old_mode)
        return mfma_bwdp_backward(p, p->mbp32, v, B, ldv, kappa, active, grad_y, ldg, grad_v, ldgv,
                                  static_cast<hipStream_t>(stream));
      if (p->mbg32 != nullptr)
        return mfma_bwdg_backward(p, p->mbg32, v, B, ldv, kappa, active, grad_y, ldg, grad_v, ldgv, old_mode, workspace,
                                  workspace_bytes, static_cast<hipStream_t>(stream));
    }
  }
  if constexpr (sizeof(T) == 8) {
    if (!force_generic && !old_mode) {
      if (p->q64 != nullptr && lmi_quad_bwd_serves_f64(p, p->q64))
        return lmi_quad_backward_f64(p, p->q64, v, B, ldv, kappa, active, grad_y, ldg, grad_v, ldgv,
                                     static_cast<hipStream_t>(stream));
    }
    if (!force_generic) {
      if (p->mb64 != nullptr)
        return mfma64_backward(p, p->mb64, v, B, ldv, kappa, active, grad_y, ldg, grad_v, ldgv, old_mode, workspace,
                               workspace_bytes, static_cast<hipStream_t>(stream));
      if (p->mbg64 != nullptr)
        return mfma64_bwdg_backward(p, p->mbg64, v, B, ldv, kappa, active, grad_y, ldg, grad_v, ldgv, old_mode,
                                    static_cast<hipStream_t>(stream));
    }
  }
  const int rcg = generic_backward<T>(p, image_of<T>(p), v, B, ldv, kappa, active, grad_y, ldg, grad_v, ldgv,
                                      old_mode, static_cast<hipStream_t>(stream));
  if (rcg == RAYEN_OK && is_mixed<T>(p)) {        // (the lane kernel has left out the LMI's term: see mixed_forward)
    if constexpr (sizeof(T) == 4)
      return lmi_block_backward_f32(p, p->w32, v, B, ldv, kappa, active, grad_y, ldg, grad_v, ldgv,
                                    static_cast<hipStream_t>(stream), 1, old_mode);
    else
      return lmi_block_backward_f64(p, p->w64, v, B, ldv, kappa, active, grad_y, ldg, grad_v, ldgv,
                                    static_cast<hipStream_t>(stream), 1, old_mode);
  }
  if (rcg == RAYEN_E_UNSUPPORTED && old_mode) {    // (the RAYEN_old head: the workgroup-per-sample kernels have it, the wave kernels do not)
    if constexpr (sizeof(T) == 4) {
      if (p->w32 != nullptr && lmi_block_bwd_serves_f32(p->w32))
        return lmi_block_backward_f32(p, p->w32, v, B, ldv, kappa, active, grad_y, ldg, grad_v, ldgv,
                                      static_cast<hipStream_t>(stream), 0, 1);
    } else {
      if (p->w64 != nullptr && lmi_block_bwd_serves_f64(p->w64))
        return lmi_block_backward_f64(p, p->w64, v, B, ldv, kappa, active, grad_y, ldg, grad_v, ldgv,
                                      static_cast<hipStream_t>(stream), 0, 1);
    }
  }
  if (rcg == RAYEN_E_UNSUPPORTED && !old_mode) {   // (nothing was launched)
    if constexpr (sizeof(T) == 4) {
      if (p->w32 != nullptr) {
        if (lmi_block_preferred(lmi_block_bwd_serves_f32(p->w32), lmi_wave_serves_f32(p->w32), lmi_dim(p), false))
          return lmi_block_backward_f32(p, p->w32, v, B, ldv, kappa, active, grad_y, ldg, grad_v, ldgv,
                                        static_cast<hipStream_t>(stream));
        return lmi_wave_backward_f32(p, p->w32, v, B, ldv, kappa, active, grad_y, ldg, grad_v, ldgv,
                                     static_cast<hipStream_t>(stream));
      }
    } else {
      if (p->w64 != nullptr) {
        if (lmi_block_preferred(lmi_block_bwd_serves_f64(p->w64), lmi_wave_serves_f64(p->w64), lmi_dim(p), true))
          return lmi_block_backward_f64(p, p->w64, v, B, ldv, kappa, active, grad_y, ldg, grad_v, ldgv,
                                        static_cast<hipStream_t>(stream));
        return lmi_wave_backward_f64(p, p->w64, v, B, ldv, kappa, active, grad_y, ldg, grad_v, ldgv,
                                     static_cast<hipStream_t>(stream));
      }
    }
  }
  return rcg;
}

// [linear rows] + ONE LMI on the workgroup-per-sample kernels, with nothing else in the set: such a pack also takes the
// products route (the GEMM forms S(v) for the whole batch; the kernels then read / write rows of T and C)
bool lmi_products_pack(const RayenPack* p) {
  if (p->mixed32 || p->mixed64 || p->q32 != nullptr || p->q64 != nullptr) return false;   // (the four-lane kernel's sizes stay with it)
  for (const RayenSegment& g : p->segs)
    if (g.type != RAYEN_SEG_LIN && g.type != RAYEN_SEG_LMI) return false;
  return (p->w32 != nullptr && lmi_block_products_serves_f32(p->w32)) || (p->w64 != nullptr && lmi_block_products_serves_f64(p->w64));
}

template <typename T>
int project_from_products(const RayenPack* p, const T* Tm, int64_t ldt, const T* v, int64_t B, int64_t ldv, T* y,
                                 int64_t ldy, T* kappa, int32_t* active, int32_t* nan_flag, void* stream) {
  if (p == nullptr || B < 0 || (B > 0 && (v == nullptr || Tm == nullptr)) || ldv < p->n || (y != nullptr && ldy < p->k) ||
      ldt < (int64_t)p->n_rows + (p->out_identity ? 0 : p->k))
    return RAYEN_E_BAD_ARG;
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) return RAYEN_E_NO_DEVICE;
  if (dev != p->device) return RAYEN_E_DEVICE_MISMATCH;
  if (p->wide == nullptr && lmi_products_pack(p)) {      // [linear rows] + one LMI: the workgroup-per-sample kernel reads S(v) from T
    if (y == nullptr) return RAYEN_E_UNSUPPORTED;
    g_last_forward = RAYEN_KERNEL_LMI_BLOCK;
    if constexpr (sizeof(T) == 4)
      return lmi_block_forward_products_f32(p, p->w32, Tm, ldt, v, B, ldv, y, ldy, kappa, active, nan_flag, static_cast<hipStream_t>(stream));
    else
      return lmi_block_forward_products_f64(p, p->w64, Tm, ldt, v, B, ldv, y, ldy, kappa, active, nan_flag, static_cast<hipStream_t>(stream));
  }
  return wide_epilogue<T>(p, p->wide, Tm, ldt, v, B, ldv, y, ldy, kappa, active, nan_flag,
                          static_cast<hipStream_t>(stream));
}

}  // namespace

namespace rayen {
int launch_simds(int n_simd) {
  const int left = n_simd - 4 * reserved_cus_cell().load(std::memory_order_relaxed);
  return left < 4 ? 4 : left;
}
}  // namespace rayen

extern "C" {

int rayen_abi_version(void) { return RAYEN_ABI_VERSION; }

int rayen_last_forward_kernel(void) { return g_last_forward; }

int rayen_reserve_cus(int cus) {
  if (cus >= 0) return reserved_cus_cell().exchange(cus > 128 ? 128 : cus, std::memory_order_relaxed);
  return reserved_cus_cell().load(std::memory_order_relaxed);
}

int rayen_pair_schedule(int mode) {
  if (mode >= 0 && mode <= 3) return pair_schedule_cell().exchange(mode, std::memory_order_relaxed);
  return pair_schedule();
}

const char* rayen_strerror(int code) {
  switch (code) {
    case RAYEN_OK: return "ok";
    case RAYEN_E_BAD_ARG: return "bad argument (null pointer, negative size or inconsistent segment table)";
    case RAYEN_E_ABI: return "ABI version mismatch";
    case RAYEN_E_NO_DEVICE: return "no usable HIP device (this library targets gfx950 / MI355X)";
    case RAYEN_E_ALLOC: return "device allocation or upload failed";
    case RAYEN_E_LAUNCH: return "kernel launch failed";
    case RAYEN_E_UNSUPPORTED: return "shape or operation not supported by the kernels";
    case RAYEN_E_DEVICE_MISMATCH: return "the pack lives on another device than the current one";
    case RAYEN_E_NOT_PREPARED: return "this precision / direction was excluded by RayenPackDesc.prepare at pack creation";
    default: return "unknown error code";
  }
}

// The split-operand kernels (bf16 triples, f16 pairs) are fp32-grade on well-conditioned sums; where a constraint set
// makes the sums cancel heavily their error constants are larger than an fp32 FMA chain's.  So
// every pack they could serve is measured ONCE, inside rayen_pack_create: probe directions (random at two
// magnitudes, +-every row of W -- the worst cancellation is along constraint normals --, directions outside the span
// of every quadratic's / cone's factor rows, near-ties of neighbouring linear rows) go through each of them,
// the exact-fp32 MFMA kernel and the fp64 lane kernel (the yardstick); a split-operand kernel may serve the pack only
// if its worst row error against fp64 is below 4e-6 of the row's size or within 1.5x of the exact-fp32 kernel's own.
// The f16-pair kernel is preferred when both pass.  Private stream, buffers freed before pack_create returns.
static int fp32_selfcheck(RayenPack* p) {
  if (p->sp32 == nullptr && p->pr32 == nullptr) return RAYEN_OK;
  if (p->fp32_mode == 2 || p->fp32_mode == 3 || p->m32 == nullptr) {
    p->sp32_state = p->sp32 != nullptr ? 1 : 0;
    p->pr32_state = p->pr32 != nullptr ? 1 : 0;
    return RAYEN_OK;
  }
  const int n = p->n, k = p->k;
  std::vector<float> hv;
  uint32_t state = 0x9E3779B9u;
  auto uniform = [&state]() {
    state = state * 1664525u + 1013904223u;
    return (float)(state >> 8) * (1.0f / 8388608.0f) - 1.0f;  // (-1, 1)
  };
  for (int b = 0; b < 512; ++b) for (int j = 0; j < n; ++j) hv.push_back(1.5f * uniform());
  for (int b = 0; b < 256; ++b) for (int j = 0; j < n; ++j) hv.push_back(96.0f * uniform());
  // (the f16-pair kernel scales every row by its own power of two: a row with components spread over many binades
  // is its worst case)
  for (int b = 0; b < 128; ++b)
    for (int j = 0; j < n; ++j) hv.push_back(std::ldexp(uniform(), -(int)((state >> 3) % 20u)));
  const int rows = p->n_rows, take = rows < 384 ? rows : 384;
  for (int t = 0; t < take; ++t) {
    const double* w = &p->W[(size_t)((int64_t)t * rows / take) * n];
    double big = 0.0;
    for (int j = 0; j < n; ++j) big = std::fmax(big, std::fabs(w[j]));
    if (!(big > 0.0) || !std::isfinite(big)) continue;
    for (int sign = -1; sign <= 1; sign += 2)
      for (int j = 0; j < n; ++j) hv.push_back((float)(sign * 1.5 * w[j] / big));
  }
  // Adversarial directions (round 3), built from where the error analysis of the pair scheme says a
  // candidate of kappa is decided by cancellation:
  //  * for every quadratic / cone, a direction (nearly) orthogonal to its factor rows -- ||U v|| is then a small
  //    difference of large products and the candidate rests on phi . v alone: the component of a random direction
  //    outside the rows' span (Gram-Schmidt), or, for a full-rank factor, its least represented direction;
  //  * near-ties of two linear rows: d_i . v = d_j . v, both beyond the boundary.
  {
    std::vector<double> Q, r(n), u(n);
    int quads = 0;
    for (const RayenSegment& g : p->segs) {
      if (g.type != RAYEN_SEG_QUAD_FAC && g.type != RAYEN_SEG_QUAD_SYM && g.type != RAYEN_SEG_SOC) continue;
      if (++quads > 96) break;
      Q.clear();
      int rank = 0;
      double weakest = 2.0;
      std::vector<double> weak_dir(n, 0.0);
      for (int t = 0; t < g.nrows && rank < n; ++t) {
        const double* w = &p->W[(size_t)(g.row0 + t) * n];
        double norm0 = 0.0;
        for (int j = 0; j < n; ++j) { u[j] = w[j]; norm0 += w[j] * w[j]; }
        if (!(norm0 > 0.0) || !std::isfinite(norm0)) continue;
        for (int pass = 0; pass < 2; ++pass)
          for (int q = 0; q < rank; ++q) {
            double dot = 0.0;
            for (int j = 0; j < n; ++j) dot += Q[(size_t)q * n + j] * u[j];
            for (int j = 0; j < n; ++j) u[j] -= dot * Q[(size_t)q * n + j];
          }
        double norm1 = 0.0;
        for (int j = 0; j < n; ++j) norm1 += u[j] * u[j];
        const double ratio = std::sqrt(norm1 / norm0);
        if (ratio < 1e-9) continue;
        if (ratio < weakest) { weakest = ratio; for (int j = 0; j < n; ++j) weak_dir[j] = u[j] / std::sqrt(norm1); }
        Q.resize((size_t)(rank + 1) * n);
        for (int j = 0; j < n; ++j) Q[(size_t)rank * n + j] = u[j] / std::sqrt(norm1);
        ++rank;
      }
      if (rank == 0) continue;
      if (rank < n) {
        for (int j = 0; j < n; ++j) r[j] = uniform();
        for (int q = 0; q < rank; ++q) {
          double dot = 0.0;
          for (int j = 0; j < n; ++j) dot += Q[(size_t)q * n + j] * r[j];
          for (int j = 0; j < n; ++j) r[j] -= dot * Q[(size_t)q * n + j];
        }
      } else {
        r = weak_dir;
      }
      double big = 0.0;
      for (int j = 0; j < n; ++j) big = std::fmax(big, std::fabs(r[j]));
      if (!(big > 1e-12)) continue;
      for (int sign = -1; sign <= 1; sign += 2)
        for (int j = 0; j < n; ++j) hv.push_back((float)(sign * 1.5 * r[j] / big));
    }
    for (const RayenSegment& g : p->segs) {
      if (g.type != RAYEN_SEG_LIN || g.nrows < 2) continue;
      const int pairs = g.nrows - 1 < 32 ? g.nrows - 1 : 32;
      for (int t = 0; t < pairs; ++t) {
        const int i = g.row0 + (int)((int64_t)t * (g.nrows - 1) / pairs), j2 = i + 1;
        const double* di = &p->W[(size_t)i * n];
        const double* dj = &p->W[(size_t)j2 * n];
        double ii = 0.0, ij = 0.0, jj = 0.0;
        for (int j = 0; j < n; ++j) { ii += di[j] * di[j]; ij += di[j] * dj[j]; jj += dj[j] * dj[j]; }
        const double det = ii * jj - ij * ij;
        if (!(det > 1e-9 * ii * jj) || !std::isfinite(det)) continue;
        const double c = 1.5, a = c * (jj - ij) / det, b = c * (ii - ij) / det;    // d_i . v = d_j . v = c
        for (int j = 0; j < n; ++j) hv.push_back((float)(a * di[j] + b * dj[j]));
      }
    }
  }
  const int64_t B0 = (int64_t)(hv.size() / (size_t)n);
  const size_t ny = (size_t)B0 * k;
  std::vector<double> hvd(hv.begin(), hv.end());
  std::vector<float> yf(3 * ny);      // triple | exact | pair
  std::vector<double> yt(ny);
  // the yardstick needs the fp64 lane image even when the caller asked for fp32 only
  bool own_g64 = false;
  int rc = RAYEN_OK;
  if (!p->g64.built) {
    rc = generic_build<double>(p, &p->g64);
    if (rc != RAYEN_OK) { generic_free<double>(&p->g64); return rc; }
    own_g64 = !(p->prepared & RAYEN_PREPARE_F64);
    if (!own_g64) p->device_bytes += p->g64.bytes;
  }
  hipStream_t st = nullptr;
  float *dv = nullptr, *dyf = nullptr;
  double *dvd = nullptr, *dyt = nullptr;
  bool ok = hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess &&
            hipMalloc(&dv, hv.size() * sizeof(float)) == hipSuccess &&
            hipMalloc(&dvd, hvd.size() * sizeof(double)) == hipSuccess &&
            hipMalloc(&dyf, yf.size() * sizeof(float)) == hipSuccess &&
            hipMalloc(&dyt, yt.size() * sizeof(double)) == hipSuccess &&
            hipMemsetAsync(dyf, 0, yf.size() * sizeof(float), st) == hipSuccess &&
            hipMemcpyAsync(dv, hv.data(), hv.size() * sizeof(float), hipMemcpyHostToDevice, st) == hipSuccess &&
            hipMemcpyAsync(dvd, hvd.data(), hvd.size() * sizeof(double), hipMemcpyHostToDevice, st) == hipSuccess;
  if (ok) {
    if (p->sp32 != nullptr) rc = mfma_split_forward(p, p->sp32, dv, B0, n, dyf, k, nullptr, nullptr, nullptr, st);
    if (rc == RAYEN_OK) rc = mfma_forward(p, p->m32, dv, B0, n, dyf + ny, k, nullptr, nullptr, nullptr, 0, st);
    if (rc == RAYEN_OK && p->pr32 != nullptr)
      rc = mfma_pair_forward(p, p->pr32, dv, B0, n, dyf + 2 * ny, k, nullptr, nullptr, nullptr, st);
    if (rc == RAYEN_OK)
      rc = generic_forward<double>(p, p->g64, dvd, B0, n, dyt, k, nullptr, nullptr, nullptr, 0, st);
    ok = rc == RAYEN_OK &&
         hipMemcpyAsync(yf.data(), dyf, yf.size() * sizeof(float), hipMemcpyDeviceToHost, st) == hipSuccess &&
         hipMemcpyAsync(yt.data(), dyt, yt.size() * sizeof(double), hipMemcpyDeviceToHost, st) == hipSuccess &&
         hipStreamSynchronize(st) == hipSuccess;
  }
  if (dv) (void)hipFree(dv);
  if (dvd) (void)hipFree(dvd);
  if (dyf) (void)hipFree(dyf);
  if (dyt) (void)hipFree(dyt);
  if (st) (void)hipStreamDestroy(st);
  if (own_g64) generic_free<double>(&p->g64);
  if (!ok) return rc != RAYEN_OK ? rc : RAYEN_E_ALLOC;
  double worst[3] = {0.0, 0.0, 0.0};
  bool broken[3] = {false, false, false};   // a NaN / inf where the yardstick is finite: sticky, the family is out
  for (int64_t b = 0; b < B0; ++b) {
    double d[3] = {0.0, 0.0, 0.0}, size = 1e-30;
    bool row_bad[3] = {false, false, false};
    for (int i = 0; i < k; ++i) {
      const double t = yt[(size_t)b * k + i];
      for (int f = 0; f < 3; ++f) {
        const double e = std::fabs((double)yf[f * ny + (size_t)b * k + i] - t);
        if (std::isfinite(e)) d[f] = std::fmax(d[f], e);     // (fmax would drop a NaN silently)
        else if (std::isfinite(t)) row_bad[f] = true;
      }
      if (std::isfinite(t)) size = std::fmax(size, std::fabs(t));
    }
    for (int f = 0; f < 3; ++f) {
      broken[f] = broken[f] || row_bad[f];
      worst[f] = std::fmax(worst[f], d[f] / size);
    }
  }
  for (int f = 0; f < 3; ++f)
    if (broken[f]) worst[f] = std::numeric_limits<double>::infinity();
  p->check_exact = worst[1];
  auto accept = [&](double w) { return (w <= 4e-6 || (std::isfinite(worst[1]) && w <= 1.5 * worst[1])) ? 1 : 2; };
  if (p->sp32 != nullptr) { p->check_split = worst[0]; p->sp32_state = accept(worst[0]); }
  if (p->pr32 != nullptr) { p->check_pair = worst[2]; p->pr32_state = accept(worst[2]); }
  return RAYEN_OK;
}

// The f16-pair backward (rayen_mfma_bwdp.hip) is accepted per pack the way the forward families are: probe directions
// (inside, moderately and far outside the set) and random incoming gradients go through it, through the exact-fp32
// kernel it would replace and through the fp64 lane-per-sample backward -- all three on the SAME kappa / arg-max record
// (the fp64 forward's), so they differentiate the same branch -- and it may serve the pack if its worst gradient row is
// within 4e-6 of the row's size or within 1.5 x the exact-fp32 kernel's own error.
static int bwd32_selfcheck(RayenPack* p) {
  // (two f16-pair backwards exist, never for the same pack: packed low-rank quadratics at n <= 32 -- mbp32, measured next to
  // mbg32 -- and dense forms at n = k = 64 -- mbd32, measured next to mb32)
  const bool dense = p->mbd32 != nullptr;
  if (p->mbp32 == nullptr && !dense) return RAYEN_OK;
  if (dense && p->fp32_mode == 3) { p->mbd32_state = 1; return RAYEN_OK; }
  if (!dense && (p->fp32_mode == 3 || p->mbg32 == nullptr)) { p->mbp32_state = 1; return RAYEN_OK; }
  const int n = p->n, k = p->k;
  uint32_t state = 0x2545F491u;
  auto uniform = [&state]() {
    state = state * 1664525u + 1013904223u;
    return (float)(state >> 8) * (1.0f / 8388608.0f) - 1.0f;
  };
  std::vector<float> hv, hg;
  for (int b = 0; b < 384; ++b) for (int j = 0; j < n; ++j) hv.push_back(1.5f * uniform());
  for (int b = 0; b < 384; ++b) for (int j = 0; j < n; ++j) hv.push_back(12.0f * uniform());
  for (int b = 0; b < 256; ++b) for (int j = 0; j < n; ++j) hv.push_back(96.0f * uniform());
  const int64_t B0 = (int64_t)(hv.size() / (size_t)n);
  for (int64_t b = 0; b < B0; ++b)
    for (int j = 0; j < k; ++j) hg.push_back(b % 5 == 4 ? std::ldexp(uniform(), -(int)((state >> 3) % 16u)) : uniform());
  std::vector<double> hvd(hv.begin(), hv.end()), hgd(hg.begin(), hg.end());
  bool own_g64 = false;
  int rc = RAYEN_OK;
  if (!p->g64.built) {
    rc = generic_build<double>(p, &p->g64);
    if (rc != RAYEN_OK) { generic_free<double>(&p->g64); return rc; }
    own_g64 = !(p->prepared & RAYEN_PREPARE_F64);
    if (!own_g64) p->device_bytes += p->g64.bytes;
  }
  const size_t nv = (size_t)B0 * n, ng = (size_t)B0 * k;
  std::vector<double> kap64((size_t)B0), gt(nv);
  std::vector<float> kap32((size_t)B0), gf(2 * nv);
  hipStream_t st = nullptr;
  float *dv = nullptr, *dg = nullptr, *dk = nullptr, *dgf = nullptr;
  double *dvd = nullptr, *dgd = nullptr, *dkd = nullptr, *dyd = nullptr, *dgt = nullptr;
  int32_t* dact = nullptr;
  bool ok = hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess &&
            hipMalloc(&dv, nv * sizeof(float)) == hipSuccess && hipMalloc(&dg, ng * sizeof(float)) == hipSuccess &&
            hipMalloc(&dk, (size_t)B0 * sizeof(float)) == hipSuccess && hipMalloc(&dgf, 2 * nv * sizeof(float)) == hipSuccess &&
            hipMalloc(&dvd, nv * sizeof(double)) == hipSuccess && hipMalloc(&dgd, ng * sizeof(double)) == hipSuccess &&
            hipMalloc(&dkd, (size_t)B0 * sizeof(double)) == hipSuccess && hipMalloc(&dyd, ng * sizeof(double)) == hipSuccess &&
            hipMalloc(&dgt, nv * sizeof(double)) == hipSuccess && hipMalloc(&dact, (size_t)B0 * 2 * sizeof(int32_t)) == hipSuccess &&
            hipMemsetAsync(dgf, 0, 2 * nv * sizeof(float), st) == hipSuccess &&
            hipMemcpyAsync(dv, hv.data(), nv * sizeof(float), hipMemcpyHostToDevice, st) == hipSuccess &&
            hipMemcpyAsync(dg, hg.data(), ng * sizeof(float), hipMemcpyHostToDevice, st) == hipSuccess &&
            hipMemcpyAsync(dvd, hvd.data(), nv * sizeof(double), hipMemcpyHostToDevice, st) == hipSuccess &&
            hipMemcpyAsync(dgd, hgd.data(), ng * sizeof(double), hipMemcpyHostToDevice, st) == hipSuccess;
  if (ok) {
    rc = generic_forward<double>(p, p->g64, dvd, B0, n, dyd, k, dkd, dact, nullptr, 0, st);
    if (rc == RAYEN_OK)
      rc = generic_backward<double>(p, p->g64, dvd, B0, n, dkd, dact, dgd, k, dgt, n, 0, st);
    ok = rc == RAYEN_OK &&
         hipMemcpyAsync(kap64.data(), dkd, (size_t)B0 * sizeof(double), hipMemcpyDeviceToHost, st) == hipSuccess &&
         hipStreamSynchronize(st) == hipSuccess;
    if (ok) {
      for (int64_t b = 0; b < B0; ++b) kap32[(size_t)b] = (float)kap64[(size_t)b];
      ok = hipMemcpyAsync(dk, kap32.data(), (size_t)B0 * sizeof(float), hipMemcpyHostToDevice, st) == hipSuccess;
    }
    if (ok) {
      rc = dense ? mfma_bwdd_backward(p, p->mbd32, dv, B0, n, dk, dact, dg, k, dgf, n, st)
                 : mfma_bwdp_backward(p, p->mbp32, dv, B0, n, dk, dact, dg, k, dgf, n, st);
      if (rc == RAYEN_OK)
        rc = dense ? mfma_backward(p, p->mb32, dv, B0, n, dk, dact, dg, k, dgf + nv, n, 0, nullptr, 0, st)
                   : mfma_bwdg_backward(p, p->mbg32, dv, B0, n, dk, dact, dg, k, dgf + nv, n, 0, nullptr, 0, st);
      ok = rc == RAYEN_OK &&
           hipMemcpyAsync(gf.data(), dgf, 2 * nv * sizeof(float), hipMemcpyDeviceToHost, st) == hipSuccess &&
           hipMemcpyAsync(gt.data(), dgt, nv * sizeof(double), hipMemcpyDeviceToHost, st) == hipSuccess &&
           hipStreamSynchronize(st) == hipSuccess;
    }
  }
  for (void* ptr : {(void*)dv, (void*)dg, (void*)dk, (void*)dgf, (void*)dvd, (void*)dgd, (void*)dkd, (void*)dyd,
                    (void*)dgt, (void*)dact})
    if (ptr) (void)hipFree(ptr);
  if (st) (void)hipStreamDestroy(st);
  if (own_g64) generic_free<double>(&p->g64);
  if (!ok) return rc != RAYEN_OK ? rc : RAYEN_E_ALLOC;
  double worst[2] = {0.0, 0.0};
  bool broken[2] = {false, false};
  for (int64_t b = 0; b < B0; ++b) {
    // (kappa within rounding of 1: fp32 and fp64 may disagree about "clipped" -- a kink, not an accuracy question)
    if (std::fabs(kap64[(size_t)b] - 1.0) < 1e-5) continue;
    double d[2] = {0.0, 0.0}, size = 1e-30;
    for (int i = 0; i < n; ++i) {
      const double t = gt[(size_t)b * n + i];
      for (int f = 0; f < 2; ++f) {
        const double e = std::fabs((double)gf[f * nv + (size_t)b * n + i] - t);
        if (std::isfinite(e)) d[f] = std::fmax(d[f], e);
        else if (std::isfinite(t)) broken[f] = true;
      }
      if (std::isfinite(t)) size = std::fmax(size, std::fabs(t));
    }
    for (int f = 0; f < 2; ++f) worst[f] = std::fmax(worst[f], d[f] / size);
  }
  for (int f = 0; f < 2; ++f)
    if (broken[f]) worst[f] = std::numeric_limits<double>::infinity();
  p->check_bwd_pair = worst[0];
  p->check_bwd_exact = worst[1];
  (dense ? p->mbd32_state : p->mbp32_state) = (worst[0] <= 4e-6 || (std::isfinite(worst[1]) && worst[0] <= 1.5 * worst[1])) ? 1 : 2;
  return RAYEN_OK;
}

int rayen_pack_create(const RayenPackDesc* desc, RayenPack** out) {
  if (desc == nullptr || out == nullptr) return RAYEN_E_BAD_ARG;
  *out = nullptr;
  if (desc->abi_version != RAYEN_ABI_VERSION) return RAYEN_E_ABI;
  int rc = check_table(desc);
  if (rc) return rc;
  if (desc->prepare < 0 || desc->prepare > 15 || desc->fp32_mode < 0 || desc->fp32_mode > 4) return RAYEN_E_BAD_ARG;
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) return RAYEN_E_NO_DEVICE;
  if (!device_is_gfx950(dev)) return RAYEN_E_NO_DEVICE;
  RayenPack* p = new (std::nothrow) RayenPack();
  if (p == nullptr) return RAYEN_E_ALLOC;
  p->device = dev;
  p->k = desc->k;
  p->n = desc->n;
  p->n_rows = desc->n_rows;
  p->out_identity = desc->out_identity ? 1 : 0;
  {
    // RayenPackDesc.fp32_mode (include/rayen_hip.h); RAYEN_FP32_MODE=0..4 overrides it.  (RAYEN_SPLIT_BF16, the
    // switch of earlier versions: 0 = exact-fp32 kernels only, 1 = bf16 triples where accepted, 2 = bf16 triples
    // without the measurement.)
    p->fp32_mode = desc->fp32_mode;
    const char* old_env = std::getenv("RAYEN_SPLIT_BF16");
    if (old_env != nullptr && old_env[0] >= '0' && old_env[0] <= '2') p->fp32_mode = old_env[0] == '0' ? 1 : (old_env[0] == '1' ? 4 : 2);
    const char* env = std::getenv("RAYEN_FP32_MODE");
    if (env != nullptr && env[0] >= '0' && env[0] <= '4') p->fp32_mode = env[0] - '0';
  }
  {
    // RAYEN_PREPARE_INWARD_BIAS: eps = 2^-20; RAYEN_INWARD_BIAS=<eps> (0 ... 1e-3; 0 = off) overrides the descriptor
    p->inward_bias = (desc->prepare & RAYEN_PREPARE_INWARD_BIAS) ? 0x1p-20 : 0.0;
    const char* env = std::getenv("RAYEN_INWARD_BIAS");
    if (env != nullptr && env[0] != '\0') {
      char* end = nullptr;
      const double eps = std::strtod(env, &end);
      if (end != env && eps >= 0.0 && eps <= 1e-3) p->inward_bias = eps;
    }
  }
  p->W.assign(desc->W, desc->W + (size_t)desc->n_rows * desc->n);
  p->y0.assign(desc->y0, desc->y0 + desc->k);
  p->NA_E.assign((size_t)desc->k * desc->n, 0.0);
  if (p->out_identity) {
    for (int i = 0; i < desc->k; ++i) p->NA_E[(size_t)i * desc->n + i] = 1.0;
  } else {
    p->NA_E.assign(desc->NA_E, desc->NA_E + (size_t)desc->k * desc->n);
  }
  p->segs.assign(desc->segments, desc->segments + desc->n_segments);
  rc = build_images(p, desc->prepare & 7);
  if (rc == RAYEN_OK) rc = fp32_selfcheck(p);
  if (rc == RAYEN_OK) rc = bwd32_selfcheck(p);
  if (rc != RAYEN_OK) { rayen_pack_destroy(p); return rc; }
  *out = p;
  return RAYEN_OK;
}

void rayen_pack_destroy(RayenPack* p) {
  if (p == nullptr) return;
  int prev = -1;
  const bool switched = hipGetDevice(&prev) == hipSuccess && prev != p->device &&
                        hipSetDevice(p->device) == hipSuccess;
  generic_free<float>(&p->g32);
  generic_free<double>(&p->g64);
  if (p->m32) mfma_free(p->m32);
  if (p->m64) mfma64_free(p->m64);
  if (p->mb32) mfma_bwd_free(p->mb32);
  if (p->mb64) mfma64_bwd_free(p->mb64);
  if (p->mbg32) mfma_bwdg_free(p->mbg32);
  if (p->mbp32) mfma_bwdp_free(p->mbp32);
  if (p->mbd32) mfma_bwdd_free(p->mbd32);
  if (p->mbg64) mfma64_bwdg_free(p->mbg64);
  if (p->sp32) mfma_split_free(p->sp32);
  if (p->pr32) mfma_pair_free(p->pr32);
  if (p->pr32m) mfma_pair_free(p->pr32m);
  if (p->ws8_32) mfma_pair_ws8_free(p->ws8_32);
  if (p->wide) wide_free(p->wide);
  if (p->q32) lmi_quad_free(p->q32);
  if (p->w32) lmi_wave_free(p->w32);
  if (p->w64) lmi_wave_free(p->w64);
  if (p->q64) lmi_quad_free(p->q64);
  if (switched) (void)hipSetDevice(prev);
  delete p;
}

int rayen_pack_info(const RayenPack* p, RayenPackInfo* info) {
  if (p == nullptr || info == nullptr) return RAYEN_E_BAD_ARG;
  std::memset(info, 0, sizeof(*info));
  info->k = p->k;
  info->n = p->n;
  info->n_rows = p->n_rows;
  info->n_segments = (int32_t)p->segs.size();
  info->device = p->device;
  info->mfma_f32 = (p->pr32 != nullptr && p->pr32_state == 1) ? 3
                   : (p->sp32 != nullptr && p->sp32_state == 1) ? 2 : (mfma_eligible(p) ? 1 : 0);
  info->mfma_f64 = mfma64_eligible(p) ? 1 : 0;
  info->prepared = p->prepared;
  info->fp32_check_split = p->check_split;
  info->fp32_check_exact = p->check_exact;
  info->fp32_check_pair = p->check_pair;
  info->bwd_f32 = !(p->prepared & 4) ? 0
                  : (p->q32 != nullptr && lmi_quad_bwd_serves_f32(p, p->q32)) ? 4
                  : (p->mbd32 != nullptr && p->mbd32_state == 1) ? 7
                  : p->mb32 != nullptr ? 1
                  : (p->mbp32 != nullptr && p->mbp32_state == 1) ? 3
                  : p->mbg32 != nullptr ? 2
                  : (p->w32 != nullptr && (lmi_wave_serves_f32(p->w32) || lmi_block_bwd_serves_f32(p->w32)) &&
                     !generic_backward_serves<float>(p, image_of<float>(p))) ? 5 : 0;
  info->bwd32_check_pair = p->check_bwd_pair;
  info->bwd32_check_exact = p->check_bwd_exact;
  int lmi_words = 0;
  for (const RayenSegment& g : p->segs)
    if (g.type == RAYEN_SEG_LMI && g.nrows + 4 * g.dim > lmi_words) lmi_words = g.nrows + 4 * g.dim;
  GenericImage<float> probe;
  probe.lmi_words = lmi_words;
  info->generic_block = generic_block_for<float>(p, probe);
  info->device_bytes = p->device_bytes;
  return RAYEN_OK;
}

int rayen_ray_project_generic_f32(const RayenPack* p, const float* v, int64_t B, int64_t ldv, float* y,
                                  int64_t ldy, float* kappa, int32_t* active, int32_t* nan_flag,
                                  void* stream) {
  g_last_forward = RAYEN_KERNEL_LANE;
  return project_generic<float>(p, v, B, ldv, y, ldy, kappa, active, nan_flag, stream);
}

int rayen_ray_project_from_products_f32(const RayenPack* p, const float* T, int64_t ldt, const float* v, int64_t B,
                                        int64_t ldv, float* y, int64_t ldy, float* kappa, int32_t* active,
                                        int32_t* nan_flag, void* stream) {
  g_last_forward = RAYEN_KERNEL_PRODUCTS;
  return project_from_products<float>(p, T, ldt, v, B, ldv, y, ldy, kappa, active, nan_flag, stream);
}

int rayen_ray_project_from_products_f64(const RayenPack* p, const double* T, int64_t ldt, const double* v, int64_t B,
                                        int64_t ldv, double* y, int64_t ldy, double* kappa, int32_t* active,
                                        int32_t* nan_flag, void* stream) {
  g_last_forward = RAYEN_KERNEL_PRODUCTS;
  return project_from_products<double>(p, T, ldt, v, B, ldv, y, ldy, kappa, active, nan_flag, stream);
}

#define RAYEN_BWD_COEFF(NAME, T)                                                                                      \
  int NAME(const RayenPack* p, const T* Tm, int64_t ldt, const T* v, int64_t B, int64_t ldv, const T* kappa,             \
           const int32_t* active, const T* grad_y, int64_t ldg, T* C, int64_t ldc, T* gs, void* stream) {                \
    if (p == nullptr || B < 0 || ldv < p->n || ldg < p->k || ldt < (int64_t)p->n_rows + (p->out_identity ? 0 : p->k) ||  \
        ldc < (int64_t)p->n_rows + (p->out_identity ? 0 : p->k) || (p->out_identity && gs == nullptr && B > 0))          \
      return RAYEN_E_BAD_ARG;                                                                                            \
    if (B > 0 && (!Tm || !v || !kappa || !active || !grad_y || !C)) return RAYEN_E_BAD_ARG;                              \
    int dev = -1;                                                                                                        \
    if (hipGetDevice(&dev) != hipSuccess) return RAYEN_E_NO_DEVICE;                                                      \
    if (dev != p->device) return RAYEN_E_DEVICE_MISMATCH;                                                                \
    if (p->wide == nullptr && lmi_products_pack(p)) {                                                                    \
      if constexpr (sizeof(T) == 4)                                                                                      \
        return lmi_block_bwd_coefficients_f32(p, p->w32, (const float*)Tm, ldt, (const float*)v, B, ldv,                 \
                                              (const float*)kappa, active, (const float*)grad_y, ldg, (float*)C, ldc,    \
                                              (float*)gs, static_cast<hipStream_t>(stream));                             \
      else                                                                                                               \
        return lmi_block_bwd_coefficients_f64(p, p->w64, (const double*)Tm, ldt, (const double*)v, B, ldv,               \
                                              (const double*)kappa, active, (const double*)grad_y, ldg, (double*)C, ldc, \
                                              (double*)gs, static_cast<hipStream_t>(stream));                            \
    }                                                                                                                    \
    return wide_bwd_coefficients<T>(p, p->wide, Tm, ldt, v, B, ldv, kappa, active, grad_y, ldg, C, ldc, gs,              \
                                    static_cast<hipStream_t>(stream));                                                   \
  }
RAYEN_BWD_COEFF(rayen_ray_project_bwd_coefficients_f32, float)
RAYEN_BWD_COEFF(rayen_ray_project_bwd_coefficients_f64, double)
#undef RAYEN_BWD_COEFF

int64_t rayen_products_rows(const RayenPack* p) {
  if (p == nullptr || (p->wide == nullptr && !lmi_products_pack(p))) return 0;
  return (int64_t)p->n_rows + (p->out_identity ? 0 : p->k);
}

// per precision (ABI v8): an LMI pack may serve the products route in fp32 only (r in (212, 304]); forward AND backward
int rayen_products_served(const RayenPack* p, int f64) {
  if (p == nullptr) return 0;
  if (p->wide != nullptr) return 1;
  if (!lmi_products_pack(p)) return 0;
  return f64 ? (p->w64 != nullptr && lmi_block_products_serves_f64(p->w64)) : (p->w32 != nullptr && lmi_block_products_serves_f32(p->w32));
}

static int project_f32(const RayenPack* p, const float* v, int64_t B, int64_t ldv, float* y, int64_t ldy,
                       float* kappa, int32_t* active, int32_t* nan_flag, void* stream, int old_mode) {
  if (p == nullptr || B < 0 || (B > 0 && v == nullptr) || ldv < p->n + old_mode ||
      (y != nullptr && ldy < p->k))
    return RAYEN_E_BAD_ARG;
  const int rc = check_ready<float>(p, false);
  if (rc) return rc;
  if (p->pr32 != nullptr && p->pr32_state == 1 && y != nullptr && !old_mode) {
    if (pair_schedule() == 3 && mfma_pair_wl_serves(p, p->pr32, v, B, ldv, y, ldy)) {
      g_last_forward = RAYEN_KERNEL_PAIR_WL;
      return mfma_pair_wl_forward(p, p->pr32, v, B, ldv, y, ldy, kappa, active, nan_flag, static_cast<hipStream_t>(stream));
    }
    if (pair_schedule() == 2 && mfma_pair_ws8_serves(p, p->pr32, p->ws8_32, v, B, ldv, y, ldy)) {
      g_last_forward = RAYEN_KERNEL_PAIR_WS;
      return mfma_pair_ws8_forward(p, p->pr32, p->ws8_32, v, B, ldv, y, ldy, kappa, active, nan_flag, static_cast<hipStream_t>(stream));
    }
    if (pair_schedule() >= 1 && mfma_pair_io_serves(p, p->pr32, v, B, ldv, y, ldy)) {
      g_last_forward = RAYEN_KERNEL_PAIR_IO;
      return mfma_pair_io_forward(p, p->pr32, v, B, ldv, y, ldy, kappa, active, nan_flag, static_cast<hipStream_t>(stream));
    }
    // Batches between two groups per CU and one group per resident wave (32 768 <= B < 131 072 on this chip): too small
    // for the trickled rows to have interior rounds, and the plain kernel leaves SIMDs with one wave or none -- there the
    // W-stationary kernel is the fastest of the three bit-identical schedules (12.7 against 18.2 us at B = 32 768,
    // 20.5 / 22.6 at 65 536, 27.2 / 28.4 at 98 304: profiles/bench/r04_midbatch_schedules.txt).
    if (pair_schedule() == 1 && mfma_pair_ws8_serves(p, p->pr32, p->ws8_32, v, B, ldv, y, ldy)) {
      g_last_forward = RAYEN_KERNEL_PAIR_WS;
      return mfma_pair_ws8_forward(p, p->pr32, p->ws8_32, v, B, ldv, y, ldy, kappa, active, nan_flag, static_cast<hipStream_t>(stream));
    }
    g_last_forward = RAYEN_KERNEL_PAIR;
    return mfma_pair_forward(p, p->pr32, v, B, ldv, y, ldy, kappa, active, nan_flag, static_cast<hipStream_t>(stream));
  }
  if (p->sp32 != nullptr && p->sp32_state == 1 && y != nullptr && !old_mode) {
    g_last_forward = RAYEN_KERNEL_TRIPLE;
    return mfma_split_forward(p, p->sp32, v, B, ldv, y, ldy, kappa, active, nan_flag,
                              static_cast<hipStream_t>(stream));
  }
  if (p->m32 != nullptr && y != nullptr) {
    g_last_forward = RAYEN_KERNEL_MFMA;
    return mfma_forward(p, p->m32, v, B, ldv, y, ldy, kappa, active, nan_flag, old_mode,
                        static_cast<hipStream_t>(stream));
  }
  // four lanes per sample pay off while one lane per sample cannot fill the chip (B/64 waves on
  // 1024 SIMDs x 2); beyond that the lane-per-sample kernel has the higher throughput in fp32
  if (y != nullptr && !old_mode && B <= 65536 && p->q32 != nullptr) {
    g_last_forward = RAYEN_KERNEL_LMI_QUAD;
    return lmi_quad_forward_f32(p, p->q32, v, B, ldv, y, ldy, kappa, active, nan_flag,
                                static_cast<hipStream_t>(stream));
  }
  g_last_forward = RAYEN_KERNEL_LANE;
  const int rcg = project_generic<float>(p, v, B, ldv, y, ldy, kappa, active, nan_flag, stream, old_mode);
  if (rcg == RAYEN_E_UNSUPPORTED && p->w32 != nullptr && y != nullptr && old_mode && !p->mixed32 && lmi_block_serves_f32(p->w32)) {
    g_last_forward = RAYEN_KERNEL_LMI_BLOCK;       // (the RAYEN_old head: only the workgroup-per-sample kernel has it)
    return lmi_block_forward_f32(p, p->w32, v, B, ldv, y, ldy, kappa, active, nan_flag, static_cast<hipStream_t>(stream), nullptr, 1, 1);
  }
  if (rcg == RAYEN_E_UNSUPPORTED && p->w32 != nullptr && y != nullptr && !old_mode) {   // (nothing was launched)
    if (lmi_block_preferred(lmi_block_serves_f32(p->w32), lmi_wave_serves_f32(p->w32), lmi_dim(p), false)) {
      g_last_forward = RAYEN_KERNEL_LMI_BLOCK;
      return lmi_block_forward_f32(p, p->w32, v, B, ldv, y, ldy, kappa, active, nan_flag, static_cast<hipStream_t>(stream));
    }
    g_last_forward = RAYEN_KERNEL_LMI_WAVE;
    return lmi_wave_forward_f32(p, p->w32, v, B, ldv, y, ldy, kappa, active, nan_flag, static_cast<hipStream_t>(stream));
  }
  return rcg;
}

int rayen_ray_project_f32(const RayenPack* p, const float* v, int64_t B, int64_t ldv, float* y,
                          int64_t ldy, float* kappa, int32_t* active, int32_t* nan_flag, void* stream) {
  return project_f32(p, v, B, ldv, y, ldy, kappa, active, nan_flag, stream, 0);
}

int rayen_ray_project_old_f32(const RayenPack* p, const float* v, int64_t B, int64_t ldv, float* y,
                              int64_t ldy, float* kappa, int32_t* active, int32_t* nan_flag, void* stream) {
  return project_f32(p, v, B, ldv, y, ldy, kappa, active, nan_flag, stream, 1);
}

int rayen_mapper_fusable(const RayenPack* p, int32_t in_dim) {
  if (p == nullptr || check_ready<float>(p, false) != RAYEN_OK) return 0;
  if (p->pr32 != nullptr && p->pr32_state == 1)   // a split-operand kernel serves the pack: its own fused form or none
    return mfma_pair_mapper_image_bytes(p, p->pr32, in_dim) > 0 ? 2 : 0;
  if (p->sp32 != nullptr && p->sp32_state == 1)
    return mfma_split_mapper_image_bytes(p, p->sp32, in_dim) > 0 ? 2 : 0;
  return (p->m32 != nullptr && mfma_mapper_fusable(p, p->m32, in_dim)) ? 1 : 0;
}

int64_t rayen_mapper_image_bytes(const RayenPack* p, int32_t in_dim) {
  if (p == nullptr) return 0;
  if (p->pr32 != nullptr && p->pr32_state == 1) return mfma_pair_mapper_image_bytes(p, p->pr32, in_dim);
  if (p->sp32 == nullptr || p->sp32_state != 1) return 0;
  return mfma_split_mapper_image_bytes(p, p->sp32, in_dim);
}

int rayen_mapper_prepare_f32(const RayenPack* p, const float* Wm, int64_t ldw, int32_t in_dim, const float* bias,
                             void* image, void* stream) {
  if (p == nullptr || Wm == nullptr || image == nullptr || in_dim <= 0 || ldw < in_dim) return RAYEN_E_BAD_ARG;
  const int rc = check_ready<float>(p, false);
  if (rc) return rc;
  if (p->pr32 != nullptr && p->pr32_state == 1)
    return mfma_pair_mapper_prepare(p, p->pr32, Wm, ldw, in_dim, bias, image, static_cast<hipStream_t>(stream));
  if (p->sp32 == nullptr || p->sp32_state != 1) return RAYEN_E_UNSUPPORTED;
  return mfma_split_mapper_prepare(p, p->sp32, Wm, ldw, in_dim, bias, image, static_cast<hipStream_t>(stream));
}

int rayen_ray_project_mapped_image_f32(const RayenPack* p, const float* x, int64_t B, int64_t ldx, int32_t in_dim,
                                       const void* image, float* v_out, int64_t ldvo, float* y, int64_t ldy,
                                       float* kappa, int32_t* active, int32_t* nan_flag, void* stream) {
  if (p == nullptr || B < 0 || in_dim <= 0 || ldx < in_dim || image == nullptr || y == nullptr || ldy < p->k ||
      (B > 0 && x == nullptr) || (v_out != nullptr && ldvo < p->n))
    return RAYEN_E_BAD_ARG;
  const int rc = check_ready<float>(p, false);
  if (rc) return rc;
  if (p->pr32 != nullptr && p->pr32_state == 1) {
    // (round 6: the W-in-LDS schedule with the mapper's image next to W's -- on the pack's own image, shared tiles included)
    if (pair_schedule() == 3 && mfma_pair_wl_serves_mapped(p, p->pr32, x, B, ldx, in_dim, v_out, ldvo, y, ldy)) {
      g_last_forward = RAYEN_KERNEL_PAIR_WL;
      return mfma_pair_wl_forward_mapped(p, p->pr32, x, B, ldx, in_dim, image, v_out, ldvo, y, ldy, kappa, active, nan_flag,
                                         static_cast<hipStream_t>(stream));
    }
    return mfma_pair_forward_mapped(p, p->pr32m != nullptr ? p->pr32m : p->pr32, x, B, ldx, in_dim, image, v_out, ldvo, y, ldy,
                                    kappa, active, nan_flag, static_cast<hipStream_t>(stream));
  }
  if (p->sp32 == nullptr || p->sp32_state != 1) return RAYEN_E_UNSUPPORTED;
  return mfma_split_forward_mapped(p, p->sp32, x, B, ldx, in_dim, image, v_out, ldvo, y, ldy, kappa, active, nan_flag,
                                   static_cast<hipStream_t>(stream));
}

int rayen_ray_project_mapped_f32(const RayenPack* p, const float* x, int64_t B, int64_t ldx, int32_t in_dim,
                                 const float* Wm, int64_t ldw, const float* bias, float* v_out,
                                 int64_t ldvo, float* y, int64_t ldy, float* kappa, int32_t* active,
                                 int32_t* nan_flag, void* stream) {
  if (p == nullptr || B < 0 || in_dim <= 0 || ldx < in_dim || ldw < in_dim || Wm == nullptr || y == nullptr ||
      ldy < p->k || (B > 0 && x == nullptr) || (v_out != nullptr && ldvo < p->n))
    return RAYEN_E_BAD_ARG;
  const int rc = check_ready<float>(p, false);
  if (rc) return rc;
  if (p->m32 == nullptr) return RAYEN_E_UNSUPPORTED;
  return mfma_forward_mapped(p, p->m32, x, B, ldx, in_dim, Wm, ldw, bias, v_out, ldvo, y, ldy, kappa,
                             active, nan_flag, static_cast<hipStream_t>(stream));
}

int rayen_ray_project_generic_f64(const RayenPack* p, const double* v, int64_t B, int64_t ldv, double* y,
                                  int64_t ldy, double* kappa, int32_t* active, int32_t* nan_flag,
                                  void* stream) {
  g_last_forward = RAYEN_KERNEL_LANE;
  return project_generic<double>(p, v, B, ldv, y, ldy, kappa, active, nan_flag, stream);
}

static int project_f64(const RayenPack* p, const double* v, int64_t B, int64_t ldv, double* y, int64_t ldy,
                       double* kappa, int32_t* active, int32_t* nan_flag, void* stream, int old_mode) {
  if (p == nullptr || B < 0 || (B > 0 && v == nullptr) || ldv < p->n + old_mode ||
      (y != nullptr && ldy < p->k))
    return RAYEN_E_BAD_ARG;
  const int rc = check_ready<double>(p, false);
  if (rc) return rc;
  if (p->m64 != nullptr && y != nullptr) {
    g_last_forward = RAYEN_KERNEL_MFMA;
    return mfma64_forward(p, p->m64, v, B, ldv, y, ldy, kappa, active, nan_flag, old_mode,
                          static_cast<hipStream_t>(stream));
  }
  if (y != nullptr && !old_mode && p->q64 != nullptr) {
    g_last_forward = RAYEN_KERNEL_LMI_QUAD;
    return lmi_quad_forward_f64(p, p->q64, v, B, ldv, y, ldy, kappa, active, nan_flag,
                                static_cast<hipStream_t>(stream));
  }
  g_last_forward = RAYEN_KERNEL_LANE;
  const int rcg = project_generic<double>(p, v, B, ldv, y, ldy, kappa, active, nan_flag, stream, old_mode);
  if (rcg == RAYEN_E_UNSUPPORTED && p->w64 != nullptr && y != nullptr && old_mode && !p->mixed64 && lmi_block_serves_f64(p->w64)) {
    g_last_forward = RAYEN_KERNEL_LMI_BLOCK;
    return lmi_block_forward_f64(p, p->w64, v, B, ldv, y, ldy, kappa, active, nan_flag, static_cast<hipStream_t>(stream), nullptr, 1, 1);
  }
  if (rcg == RAYEN_E_UNSUPPORTED && p->w64 != nullptr && y != nullptr && !old_mode &&
      lmi_block_preferred(lmi_block_serves_f64(p->w64), lmi_wave_serves_f64(p->w64), lmi_dim(p), true)) {
    g_last_forward = RAYEN_KERNEL_LMI_BLOCK;
    return lmi_block_forward_f64(p, p->w64, v, B, ldv, y, ldy, kappa, active, nan_flag, static_cast<hipStream_t>(stream));
  }
  if (rcg == RAYEN_E_UNSUPPORTED && p->w64 != nullptr && y != nullptr && !old_mode) {
    g_last_forward = RAYEN_KERNEL_LMI_WAVE;
    return lmi_wave_forward_f64(p, p->w64, v, B, ldv, y, ldy, kappa, active, nan_flag, static_cast<hipStream_t>(stream));
  }
  return rcg;
}

int rayen_ray_project_f64(const RayenPack* p, const double* v, int64_t B, int64_t ldv, double* y,
                          int64_t ldy, double* kappa, int32_t* active, int32_t* nan_flag, void* stream) {
  return project_f64(p, v, B, ldv, y, ldy, kappa, active, nan_flag, stream, 0);
}

int rayen_ray_project_old_f64(const RayenPack* p, const double* v, int64_t B, int64_t ldv, double* y,
                              int64_t ldy, double* kappa, int32_t* active, int32_t* nan_flag, void* stream) {
  return project_f64(p, v, B, ldv, y, ldy, kappa, active, nan_flag, stream, 1);
}

int rayen_ray_project_bwd_f32(const RayenPack* p, const float* v, int64_t B, int64_t ldv,
                              const float* kappa, const int32_t* active, const float* grad_y,
                              int64_t ldg, float* grad_v, int64_t ldgv, void* stream) {
  return project_bwd<float>(p, v, B, ldv, kappa, active, grad_y, ldg, grad_v, ldgv, stream);
}

int64_t rayen_bwd_workspace_bytes_f32(const RayenPack* p, int64_t B) {
  if (p == nullptr || B <= 0 || check_ready<float>(p, true) != RAYEN_OK) return 0;
  if (p->q32 != nullptr && lmi_quad_bwd_serves_f32(p, p->q32)) return 0;
  // (the f16-pair backwards stream the batch in order: nothing to sort.  The dense-form one declines rows that are not 16-byte
  // aligned -- the bucketed walk then runs unsorted, same results)
  if (p->mbd32 != nullptr && p->mbd32_state == 1 && dense_pairs_backward_enabled() &&
      mfma_bwdd_serves(p, p->mbd32, nullptr, B, 4, nullptr, 4, nullptr, 4))
    return 0;
  if (p->mb32 != nullptr) return mfma_bwd_workspace_bytes(p, p->mb32, B);
  if (p->mbp32 != nullptr && p->mbp32_state == 1) return 0;
  return p->mbg32 != nullptr ? mfma_bwdg_workspace_bytes(p, p->mbg32, B) : 0;
}

int64_t rayen_bwd_workspace_bytes_f64(const RayenPack* p, int64_t B) {
  if (p == nullptr || B <= 0 || check_ready<double>(p, true) != RAYEN_OK) return 0;
  if (p->q64 != nullptr && lmi_quad_bwd_serves_f64(p, p->q64)) return 0;
  return p->mb64 != nullptr ? mfma64_bwd_workspace_bytes(p, p->mb64, B) : 0;
}

int rayen_ray_project_bwd_ws_f64(const RayenPack* p, const double* v, int64_t B, int64_t ldv,
                                 const double* kappa, const int32_t* active, const double* grad_y,
                                 int64_t ldg, double* grad_v, int64_t ldgv, void* workspace, int64_t workspace_bytes,
                                 void* stream) {
  return project_bwd<double>(p, v, B, ldv, kappa, active, grad_y, ldg, grad_v, ldgv, stream, 0, false, workspace,
                             workspace_bytes);
}

int rayen_ray_project_bwd_ws_f32(const RayenPack* p, const float* v, int64_t B, int64_t ldv,
                                 const float* kappa, const int32_t* active, const float* grad_y,
                                 int64_t ldg, float* grad_v, int64_t ldgv, void* workspace, int64_t workspace_bytes,
                                 void* stream) {
  return project_bwd<float>(p, v, B, ldv, kappa, active, grad_y, ldg, grad_v, ldgv, stream, 0, false, workspace,
                            workspace_bytes);
}

int rayen_ray_project_bwd_generic_f32(const RayenPack* p, const float* v, int64_t B, int64_t ldv,
                                      const float* kappa, const int32_t* active, const float* grad_y,
                                      int64_t ldg, float* grad_v, int64_t ldgv, void* stream) {
  return project_bwd<float>(p, v, B, ldv, kappa, active, grad_y, ldg, grad_v, ldgv, stream, 0, true);
}

int rayen_ray_project_bwd_generic_f64(const RayenPack* p, const double* v, int64_t B, int64_t ldv,
                                      const double* kappa, const int32_t* active, const double* grad_y,
                                      int64_t ldg, double* grad_v, int64_t ldgv, void* stream) {
  return project_bwd<double>(p, v, B, ldv, kappa, active, grad_y, ldg, grad_v, ldgv, stream, 0, true);
}

int rayen_ray_project_bwd_f64(const RayenPack* p, const double* v, int64_t B, int64_t ldv,
                              const double* kappa, const int32_t* active, const double* grad_y,
                              int64_t ldg, double* grad_v, int64_t ldgv, void* stream) {
  return project_bwd<double>(p, v, B, ldv, kappa, active, grad_y, ldg, grad_v, ldgv, stream);
}

int rayen_ray_project_old_bwd_f32(const RayenPack* p, const float* v, int64_t B, int64_t ldv,
                                  const float* kappa, const int32_t* active, const float* grad_y,
                                  int64_t ldg, float* grad_v, int64_t ldgv, void* stream) {
  return project_bwd<float>(p, v, B, ldv, kappa, active, grad_y, ldg, grad_v, ldgv, stream, 1);
}

int rayen_ray_project_old_bwd_f64(const RayenPack* p, const double* v, int64_t B, int64_t ldv,
                                  const double* kappa, const int32_t* active, const double* grad_y,
                                  int64_t ldg, double* grad_v, int64_t ldgv, void* stream) {
  return project_bwd<double>(p, v, B, ldv, kappa, active, grad_y, ldg, grad_v, ldgv, stream, 1);
}

}  // extern "C"
