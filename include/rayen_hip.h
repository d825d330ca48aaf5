/*
 * rayen_hip.h -- C ABI of the MI355X (gfx950) ray-shooting projection.
 *
 * The reference (leggedrobotics/rayen, pure Python/PyTorch) has no FFI layer; its
 * boundary for this path is the Python method ConstraintModule.forward
 * (rayen/constraint_module.py:520-533) -> forwardForRAYEN (:468-474) ->
 * computeKappa (:351-458) -> getyFromz (:512-514).  This header is the C-ABI a
 * binding for that path attaches to: plain pointers and sizes, no torch types.
 * INTEGRATION.md shows the ctypes stub a maintainer of the reference would add.
 *
 * Mathematical contract (homogeneity identity):
 *
 *     y = y0 + NA_E v / max(1, kappa(v)),     kappa(v) = max_c kappa_c(v) >= 0
 *
 * which equals constraint_module.py:468-474 (normalise, kappa(v_bar), clip the
 * step at 1/kappa, lift with NA_E, add yp) for every v, including v = 0.
 *
 * All constants are handed over ONCE, in fp64 and on the HOST, as a matrix W
 * [n_rows x n] whose rows are the linear functionals of v that computeKappa
 * needs, plus a segment table that says how each group of rows is reduced:
 *
 *   LIN       rows D_i = A_p,i / (b_p,i - A_p,i z0)           kappa = relu(max_i D_i v)      (:38, :353)
 *   QUAD_SYM  aux row phi NA_E ; rows G = NA_E' delta NA_E    kappa = phi.v + sqrt(v'Gv)     (:99-122, :374)
 *   QUAD_FAC  aux row phi NA_E ; rows U with U'U = G          kappa = phi.v + ||U v||        (same, low-rank form)
 *   SOC       aux rows c'NA_E, (M'beta)'NA_E ; rows M NA_E    larger root of a'x^2+b'x+c'=0  (:383-399, :339-348)
 *   LMI       rows = packed lower triangle of -L'F_a L . NA_E  kappa = relu(lambda_max)      (:43-52, :401-449)
 *
 * The library uploads W to the current HIP device and builds whatever device
 * images its kernels want (fragment-ordered for MFMA, row-blocked for the
 * generic path); those layouts are private.
 *
 * Conventions: every entry point returns 0 on success or a negative RAYEN_E_*
 * code; nothing throws across the ABI; no per-call allocation; launches are
 * asynchronous on the caller's stream (no host sync); tensor memory is owned by
 * the caller and must live on the pack's device; a pack is immutable after
 * rayen_pack_create returns, so it may be shared by threads and streams and
 * captured into a HIP graph from its first call.
 */
#ifndef RAYEN_HIP_H
#define RAYEN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RAYEN_ABI_VERSION 8

enum {
  RAYEN_OK = 0,
  RAYEN_E_BAD_ARG = -1,       /* null pointer, negative size, inconsistent table */
  RAYEN_E_ABI = -2,           /* desc->abi_version != RAYEN_ABI_VERSION */
  RAYEN_E_NO_DEVICE = -3,     /* no HIP device / wrong architecture (needs gfx950) */
  RAYEN_E_ALLOC = -4,         /* device allocation or upload failed in pack_create */
  RAYEN_E_LAUNCH = -5,        /* hipGetLastError() after a launch */
  RAYEN_E_UNSUPPORTED = -6,   /* shape outside what the kernels handle (see rayen_pack_info) */
  RAYEN_E_DEVICE_MISMATCH = -7, /* pack lives on another device than the current one */
  RAYEN_E_NOT_PREPARED = -8   /* precision / direction excluded by RayenPackDesc.prepare */
};

/* RayenPackDesc.prepare: the kernel families whose device images rayen_pack_create builds.  0 = everything
 * (fp32 + fp64, forward + backward); neither precision bit set = both precisions (RAYEN_PREPARE_FWD_ONLY alone =
 * forward only, fp32 and fp64).  A call into a family that was left out returns RAYEN_E_NOT_PREPARED. */
enum {
  RAYEN_PREPARE_ALL = 0,
  RAYEN_PREPARE_F32 = 1,
  RAYEN_PREPARE_F64 = 2,
  RAYEN_PREPARE_FWD_ONLY = 4,
  /* (ABI v8) not a family: the fp32 images evaluate (1 + 2^-20) kappa, so clipped samples stop 9.5e-7 (relative to the
   * step) short of the boundary instead of ON it, where half of the fp32 roundings fall outside -- SURVEY.md section 7
   * "fp32 feasibility", CM:374.  Measured on 65 536 rows (profiles/bench/r06_inward_bias.txt): config 3 rows with a residual
   * > 0: 31 594 -> 1; outputs move by at most 1.3e-6 of a row (the parity bar is 1e-5).  fp64 is untouched.
   * RAYEN_INWARD_BIAS=<eps> in the environment of rayen_pack_create overrides (0 = off, at most 1e-3). */
  RAYEN_PREPARE_INWARD_BIAS = 8
};

enum {
  RAYEN_SEG_LIN = 0,
  RAYEN_SEG_QUAD_SYM = 1,
  RAYEN_SEG_QUAD_FAC = 2,
  RAYEN_SEG_SOC = 3,
  RAYEN_SEG_LMI = 4
};

typedef struct RayenSegment {
  int32_t type;     /* RAYEN_SEG_* */
  int32_t row0;     /* first row of the group in W */
  int32_t nrows;    /* LIN m | QUAD_SYM n | QUAD_FAC rank | SOC r_M | LMI r(r+1)/2 */
  int32_t aux_row;  /* QUAD_*: row of phi NA_E | SOC: row of c'NA_E, next row is (M'beta)'NA_E | else -1 */
  int32_t dim;      /* LMI: r | else 0 */
  int32_t reserved;
  double f0;        /* SOC: tau = c'y0 + d */
  double f1;        /* SOC: a' = ||M y0 + s||^2 - tau^2  (< 0 for an interior y0) */
} RayenSegment;

typedef struct RayenPackDesc {
  int32_t abi_version;          /* RAYEN_ABI_VERSION */
  int32_t k;                    /* ambient dimension (rows of NA_E, length of y) */
  int32_t n;                    /* subspace dimension (columns of W, used columns of v) */
  int32_t n_rows;               /* rows of W */
  int32_t n_segments;
  int32_t out_identity;         /* 1: NA_E is the k x k identity (no equality constraints) */
  const double* W;              /* HOST [n_rows, n] row-major */
  const RayenSegment* segments; /* HOST [n_segments] */
  const double* NA_E;           /* HOST [k, n] row-major; may be NULL when out_identity */
  const double* y0;             /* HOST [k] */
  int32_t prepare;              /* RAYEN_PREPARE_* bits; 0 = all families */
  int32_t fp32_mode;            /* which family serves the fp32 forward of a set without LMI, n <= 64.  0 (default): the
                                   fastest one the creation-time accuracy measurement accepts -- f16-pair kernel, then
                                   bf16-triple kernel, then the exact-fp32 MFMA kernels | 1 = exact-fp32 MFMA kernels only |
                                   2 = bf16-triple kernel without the measurement | 3 = f16-pair kernel without the
                                   measurement | 4 = as 0 but never the f16-pair kernel */
} RayenPackDesc;

typedef struct RayenPackInfo {
  int32_t k, n, n_rows, n_segments;
  int32_t device;               /* HIP device ordinal the pack lives on */
  int32_t mfma_f32;             /* fp32 forward: 0 lane-per-sample kernels | 1 fp32 MFMA kernel | 2 bf16-triple kernel (six
                                   bf16 MFMA products per fp32 product) | 3 f16-pair kernel (three f16 MFMA products per
                                   fp32 product, operands carried to 22 bits); 2 and 3 are fp32-grade on the probes */
  int32_t generic_block;        /* fp32 generic path: workgroup size with v staged in LDS; 0 = v read from global memory */
  int32_t mfma_f64;             /* 1: the fp64 MFMA path serves this pack */
  int64_t device_bytes;         /* bytes of device memory the pack holds */
  int32_t prepared;             /* RAYEN_PREPARE_F32 | RAYEN_PREPARE_F64 | 4 (backward) */
  int32_t bwd_f32;              /* fp32 backward: 0 lane-per-sample kernel | 1 fp32 MFMA kernel (NA_E = I, dense forms) | 2 fp32
                                   MFMA kernel, general shapes | 3 f16-pair kernel (packed low-rank quadratics, n <= 32; accepted
                                   by a creation-time measurement like the forward's) | 4 the four-lanes-per-sample LMI kernel | 5 the
                                   wave-per-sample LMI kernel (matrices the lane kernels cannot hold) | 7 (round 6) f16-pair kernel for
                                   dense forms at n = k = 64, forms resident in LDS, one launch, no workspace (accepted by the same
                                   measurement; batches below a group per resident wave and RAYEN_old stay on 1) */
  double fp32_check_split;      /* worst row error (relative to the row's size) against fp64 on the creation-time probe */
  double fp32_check_exact;      /* directions: bf16-triple kernel, exact-fp32 kernel, */
  double fp32_check_pair;       /* f16-pair kernel; -1 = not measured */
  double bwd32_check_pair;      /* worst gradient-row error against the fp64 lane backward on the creation-time probe: */
  double bwd32_check_exact;     /* f16-pair backward, the exact-fp32 kernel it replaces; -1 = not measured */
} RayenPackInfo;

typedef struct RayenPack RayenPack;

int rayen_abi_version(void);
const char* rayen_strerror(int code);

/* Diagnostics (ABI v4): the kernel family that served the calling thread's most recent forward call
 * (rayen_ray_project_f32 / _f64 / _old_* / _generic_*); thread-local, no device work.  The reference has one code
 * path (rayen/constraint_module.py:351-474); here the shape of a call (alignment, batch, leading dimensions) can
 * select between instruction streams that compute the same values, and tests / benchmarks want to know which. */
enum {
  RAYEN_KERNEL_NONE = 0,
  RAYEN_KERNEL_LANE = 1,      /* lane-per-sample kernels (rayen_generic.hip) */
  RAYEN_KERNEL_MFMA = 2,      /* exact fp32 / fp64 matrix-core kernels */
  RAYEN_KERNEL_TRIPLE = 3,    /* bf16 triples */
  RAYEN_KERNEL_PAIR = 4,      /* f16 pairs */
  RAYEN_KERNEL_PAIR_IO = 5,   /* f16 pairs, rows of v and y trickled through LDS under the tile walk */
  RAYEN_KERNEL_LMI_QUAD = 6,  /* four lanes per sample (one LMI + linear rows) */
  RAYEN_KERNEL_LMI_WAVE = 7,  /* one wave per sample, the matrix in LDS (one LMI beyond ~30 x 30 + linear rows) */
  RAYEN_KERNEL_PAIR_WS = 8,   /* f16 pairs, W-stationary (ABI v6): the tiles of W resident in the registers of a workgroup's eight
                                 waves, the batch streamed through a shared B-operand image in LDS */
  RAYEN_KERNEL_PRODUCTS = 9,  /* wide sets (ABI v7): the epilogue over products T = v W_ext' of a library GEMM */
  RAYEN_KERNEL_PAIR_WL = 11,  /* f16 pairs, the image of W resident in LDS, four waves per SIMD on groups of 32 samples (round 6) */
  RAYEN_KERNEL_LMI_BLOCK = 10 /* one workgroup per sample, the packed lower triangle in LDS (one LMI to ~280 x 280 + linear rows; round 5) */
};
int rayen_last_forward_kernel(void);

/* Tuning / A-B switch (ABI v4, process-wide): which SCHEDULE of the f16-pair forward serves the calls whose shape
 * allows it -- 3 (default since ABI v8 / round 6): the image of W resident in LDS (NA_E = I, n = k = 32 or 64, the image
 * within 160 KiB, rows 16-byte aligned and within 4 GiB; every batch size), else as 1 | 1 (the default of ABI v4-v7): rows of v and y trickled through LDS
 * under the tile walk for batches that give every resident wave a group (B >= 131 072 on MI355X), the W-stationary kernel
 * for 32 768 <= B < 131 072 where it serves the pack (ABI v7), else the plain kernel | 0: always the plain kernel |
 * 2 (ABI v6): the W-stationary kernel wherever it serves, else as 1.  All compute the same values bit for bit.  Initial
 * value from the environment variable RAYEN_PAIR_IO.  mode outside 0..3 only queries.  Returns the previous setting. */
int rayen_pair_schedule(int mode);

/* Multi-GPU step (ABI v4, process-wide): leave `cus` compute units out of the persistent grids of the projection
 * kernels, for a collective that runs beside them -- the all-gather of y that BASELINE.json's multi-GPU layout adds
 * (RCCL's kernels need CUs; a grid that fills every SIMD with resident waves serialises them behind the projection).
 * 0 (default) = every CU.  Initial value from RAYEN_RESERVE_CUS.  cus < 0 only queries.  Returns the previous
 * setting.  The reference has no counterpart (it has no parallelism of any kind). */
int rayen_reserve_cus(int cus);

/* Upload the constants to the CURRENT HIP device and build EVERY device image the entry points below will
 * read (all kernel families selected by desc->prepare), including the one-time accuracy measurement that
 * decides which fp32 forward family serves the pack (RayenPackInfo.mfma_f32).  This is the only call that
 * allocates device memory or synchronises; it must not run while a stream of this thread is being captured.
 * Host arrays are copied; they may be freed after the call returns. */
int rayen_pack_create(const RayenPackDesc* desc, RayenPack** out);
void rayen_pack_destroy(RayenPack* pack);
int rayen_pack_info(const RayenPack* pack, RayenPackInfo* info);

/* Forward: v [B, ldv] row-major (first n columns read) -> y [B, ldy] (first k
 * columns written).  Optional outputs (NULL to skip):
 *   kappa  [B]    kappa(v) of the UN-normalised direction
 *   active [B,2]  (segment index or -1 when kappa == 0, W row of the active
 *                 linear constraint or 0)  -- what the backward needs
 *   nan_flag [1]  set to 1 (never cleared) when any written y is NaN; the
 *                 fused replacement of constraint_module.py:531's full-tensor check
 * y may be NULL to compute kappa only (the computeKappa helper, :351).
 * `stream` is a hipStream_t (NULL = the null stream).  Calls are asynchronous,
 * allocate nothing and touch no other stream. */
int rayen_ray_project_f32(const RayenPack* pack, const float* v, int64_t B, int64_t ldv,
                          float* y, int64_t ldy, float* kappa, int32_t* active,
                          int32_t* nan_flag, void* stream);
int rayen_ray_project_f64(const RayenPack* pack, const double* v, int64_t B, int64_t ldv,
                          double* y, int64_t ldy, double* kappa, int32_t* active,
                          int32_t* nan_flag, void* stream);

/* Same contract, forced through the generic (non-MFMA) kernels; used by the
 * parity tests to cover both implementations on every shape. */
int rayen_ray_project_generic_f32(const RayenPack* pack, const float* v, int64_t B, int64_t ldv,
                                  float* y, int64_t ldy, float* kappa, int32_t* active,
                                  int32_t* nan_flag, void* stream);
int rayen_ray_project_generic_f64(const RayenPack* pack, const double* v, int64_t B, int64_t ldv,
                                  double* y, int64_t ldy, double* kappa, int32_t* active,
                                  int32_t* nan_flag, void* stream);

/* Wide sets (ABI v7): n beyond what the matrix-core kernels keep in registers (64 split-operand / 128 exact fp32).
 * There T = v W_ext' is a PLAIN GEMM and belongs to the vendor library (hipBLASLt / rocBLAS: the host side calls it on
 * the caller's stream -- torch.mm in rayen_amd/ops.py); this entry is the rest of constraint_module.py:351-458 +
 * 468-474 as ONE bandwidth-bound kernel over T: a wave per sample reduces its row of T segment by segment (max /
 * phi.v + sqrt / cone root), takes kappa = relu(max) and writes y = y0 + (NA_E v) / max(1, kappa).
 *   W_ext = [W ; NA_E]   (rayen_products_rows() rows: n_rows, + k when the set has equality constraints; the caller
 *                         builds it from the RayenPackDesc it created the pack with)
 *   T [B, ldt] row-major with ldt >= rayen_products_rows(); v, y, kappa, active, nan_flag as in rayen_ray_project_*.
 * Packs with an LMI segment: [linear rows] + ONE LMI that the workgroup-per-sample kernels hold are served (round 5: S(v) is
 * read from the LMI's rows of T, and the backward leaves (2 - [i = j]) x_i x_j there for the caller's second GEMM); any other
 * pack with an LMI is not (RAYEN_E_UNSUPPORTED; rayen_products_rows() == 0). */
int64_t rayen_products_rows(const RayenPack* pack);
/* (ABI v8) 1 when the products route serves this pack at the given precision (f64 = 0: fp32, 1: fp64), forward and backward:
 * a pack with an LMI of 213 ... 304 rows has the route in fp32 only -- ask before running the GEMM. */
int rayen_products_served(const RayenPack* pack, int f64);
int rayen_ray_project_from_products_f32(const RayenPack* pack, const float* T, int64_t ldt, const float* v, int64_t B,
                                        int64_t ldv, float* y, int64_t ldy, float* kappa, int32_t* active,
                                        int32_t* nan_flag, void* stream);
int rayen_ray_project_from_products_f64(const RayenPack* pack, const double* T, int64_t ldt, const double* v, int64_t B,
                                        int64_t ldv, double* y, int64_t ldy, double* kappa, int32_t* active,
                                        int32_t* nan_flag, void* stream);

/* Backward of the wide route: grad_v = C W_ext (+ gs), with this entry writing the COEFFICIENTS C [B, ldc >= rows] --
 * row b is zero outside sample b's active segment; s g on the NA_E rows -- and, for sets without equality constraints
 * (NA_E = I), gs [B, k] = s g; the caller finishes with one vendor GEMM.  T is the forward's product matrix (recomputed
 * by the caller: one more GEMM), kappa / active the forward's outputs. */
int rayen_ray_project_bwd_coefficients_f32(const RayenPack* pack, const float* T, int64_t ldt, const float* v, int64_t B,
                                           int64_t ldv, const float* kappa, const int32_t* active, const float* grad_y,
                                           int64_t ldg, float* C, int64_t ldc, float* gs, void* stream);
int rayen_ray_project_bwd_coefficients_f64(const RayenPack* pack, const double* T, int64_t ldt, const double* v, int64_t B,
                                           int64_t ldv, const double* kappa, const int32_t* active, const double* grad_y,
                                           int64_t ldg, double* C, int64_t ldc, double* gs, void* stream);

/* Backward of y w.r.t. v (vector-Jacobian product):
 *   grad_v = s N'g - [kappa > 1] s^2 (g . N v) grad kappa(v),   s = 1/max(1,kappa)
 * with grad kappa taken on the active constraint only, which is what autograd
 * produces for constraint_module.py:351-474 (max -> argmax, relu, eigvalsh). */
int rayen_ray_project_bwd_f32(const RayenPack* pack, const float* v, int64_t B, int64_t ldv,
                              const float* kappa, const int32_t* active,
                              const float* grad_y, int64_t ldg,
                              float* grad_v, int64_t ldgv, void* stream);
int rayen_ray_project_bwd_f64(const RayenPack* pack, const double* v, int64_t B, int64_t ldv,
                              const double* kappa, const int32_t* active,
                              const double* grad_y, int64_t ldg,
                              double* grad_v, int64_t ldgv, void* stream);
/* The same backward with a caller-provided scratch buffer (device memory, 16-byte aligned, contents irrelevant
 * before and after the call): rayen_bwd_workspace_bytes_f32/_f64() is what the pack can use for a batch of B (0: nothing).
 * With it, packs made of several dense quadratic / cone forms first group the samples by the constraint that set
 * kappa (two small launches in the workspace) and evaluate, per group, only that constraint's form instead of all of
 * them (config 3: 0.17 -> 0.09 ms).  Results are the same as without; workspace = NULL is rayen_ray_project_bwd_f32/_f64. */
int64_t rayen_bwd_workspace_bytes_f32(const RayenPack* pack, int64_t B);
int64_t rayen_bwd_workspace_bytes_f64(const RayenPack* pack, int64_t B);
int rayen_ray_project_bwd_ws_f64(const RayenPack* pack, const double* v, int64_t B, int64_t ldv,
                                 const double* kappa, const int32_t* active,
                                 const double* grad_y, int64_t ldg,
                                 double* grad_v, int64_t ldgv, void* workspace, int64_t workspace_bytes, void* stream);
int rayen_ray_project_bwd_ws_f32(const RayenPack* pack, const float* v, int64_t B, int64_t ldv,
                                 const float* kappa, const int32_t* active,
                                 const float* grad_y, int64_t ldg,
                                 float* grad_v, int64_t ldgv, void* workspace, int64_t workspace_bytes, void* stream);
/* Same contract, forced through the lane-per-sample backward (rayen_ray_project_bwd_f32/_f64 pick the
 * matrix-core backward when the pack allows it); used by the tests to cover both. */
int rayen_ray_project_bwd_generic_f32(const RayenPack* pack, const float* v, int64_t B, int64_t ldv,
                                      const float* kappa, const int32_t* active,
                                      const float* grad_y, int64_t ldg,
                                      float* grad_v, int64_t ldgv, void* stream);
int rayen_ray_project_bwd_generic_f64(const RayenPack* pack, const double* v, int64_t B, int64_t ldv,
                                      const double* kappa, const int32_t* active,
                                      const double* grad_y, int64_t ldg,
                                      double* grad_v, int64_t ldgv, void* stream);

/* The RAYEN_old head (rayen/constraint_module.py:460-466, forwardForRAYENOld): the input carries one
 * more column, beta = v[:, n] (so ldv >= n + 1), and the step is 1/(exp(beta) + kappa(v_bar)) along
 * v_bar = v/||v||:   y = y0 + NA_E v / (||v|| exp(beta) + kappa(v))   (y = y0 when v = 0).
 * kappa / active / nan_flag as above.  The backward writes n + 1 columns (ldgv >= n + 1):
 *   grad_v = s N'g - s^2 (g . N v) (exp(beta) v/||v|| + grad kappa(v)),  grad_beta = -s^2 (g . N v) ||v|| exp(beta). */
int rayen_ray_project_old_f32(const RayenPack* pack, const float* v, int64_t B, int64_t ldv,
                              float* y, int64_t ldy, float* kappa, int32_t* active,
                              int32_t* nan_flag, void* stream);
int rayen_ray_project_old_f64(const RayenPack* pack, const double* v, int64_t B, int64_t ldv,
                              double* y, int64_t ldy, double* kappa, int32_t* active,
                              int32_t* nan_flag, void* stream);
int rayen_ray_project_old_bwd_f32(const RayenPack* pack, const float* v, int64_t B, int64_t ldv,
                                  const float* kappa, const int32_t* active,
                                  const float* grad_y, int64_t ldg,
                                  float* grad_v, int64_t ldgv, void* stream);
int rayen_ray_project_old_bwd_f64(const RayenPack* pack, const double* v, int64_t B, int64_t ldv,
                                  const double* kappa, const int32_t* active,
                                  const double* grad_y, int64_t ldg,
                                  double* grad_v, int64_t ldgv, void* stream);

/* The module's mapper fused in front of the projection (rayen/constraint_module.py:259-263 creates
 * mapper = nn.Linear(input_dim, n); :525 applies it before forwardForRAYEN):
 *     v = Wm x + bias,   y = y0 + NA_E v / max(1, kappa(v))
 * in ONE launch; v is kept in registers and reaches memory only when v_out != NULL (the backward
 * needs it).  x [B, ldx] (first in_dim columns read).  rayen_mapper_fusable() says which form serves
 * this pack and input width:
 *   0  none: the caller runs its GEMM followed by rayen_ray_project_f32;
 *   1  rayen_ray_project_mapped_f32: Wm [n, ldw] row-major (= Linear.weight, rows 16-byte aligned) and
 *      bias [n] or NULL are read in place (exact-fp32 MFMA family; in_dim a multiple of 4, <= 64);
 *   2  rayen_ray_project_mapped_image_f32 (packs served by the split-operand kernel, no equality
 *      constraints, in_dim <= n rounded up to 32): Wm and bias are first converted into a caller-owned image of
 *      rayen_mapper_image_bytes() bytes (16-byte aligned device memory) by rayen_mapper_prepare_f32 -- one
 *      small asynchronous launch, to be repeated whenever the weights change -- and the projection reads
 *      that image.  The library keeps no per-mapper state: packs stay immutable. */
int rayen_mapper_fusable(const RayenPack* pack, int32_t in_dim);
int rayen_ray_project_mapped_f32(const RayenPack* pack, const float* x, int64_t B, int64_t ldx,
                                 int32_t in_dim, const float* Wm, int64_t ldw, const float* bias,
                                 float* v_out, int64_t ldvo, float* y, int64_t ldy, float* kappa,
                                 int32_t* active, int32_t* nan_flag, void* stream);
int64_t rayen_mapper_image_bytes(const RayenPack* pack, int32_t in_dim);
int rayen_mapper_prepare_f32(const RayenPack* pack, const float* Wm, int64_t ldw, int32_t in_dim,
                             const float* bias, void* image, void* stream);
int rayen_ray_project_mapped_image_f32(const RayenPack* pack, const float* x, int64_t B, int64_t ldx,
                                       int32_t in_dim, const void* image, float* v_out, int64_t ldvo,
                                       float* y, int64_t ldy, float* kappa, int32_t* active,
                                       int32_t* nan_flag, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RAYEN_HIP_H */
