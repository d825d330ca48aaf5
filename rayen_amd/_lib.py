"""ctypes binding of the C ABI in ``include/rayen_hip.h`` (``librayen_hip.so``).

There is deliberately no fallback: if the shared library is missing or an entry
point fails, the error is raised to the caller.
"""
from __future__ import annotations

import ctypes
import os

from ._build import LIBRARY

ABI_VERSION = 8

SEG_LIN, SEG_QUAD_SYM, SEG_QUAD_FAC, SEG_SOC, SEG_LMI = range(5)
E_UNSUPPORTED = -6      # RAYEN_E_UNSUPPORTED (include/rayen_hip.h)
PREPARE_ALL, PREPARE_F32, PREPARE_F64, PREPARE_FWD_ONLY, PREPARE_INWARD_BIAS = 0, 1, 2, 4, 8

# every symbol include/rayen_hip.h declares
EXPORTS = (
    "rayen_abi_version", "rayen_strerror", "rayen_pack_create", "rayen_pack_destroy",
    "rayen_pack_info", "rayen_ray_project_f32", "rayen_ray_project_f64",
    "rayen_ray_project_generic_f32", "rayen_ray_project_generic_f64", "rayen_ray_project_bwd_f32",
    "rayen_ray_project_bwd_f64", "rayen_ray_project_old_f32", "rayen_ray_project_old_f64",
    "rayen_ray_project_old_bwd_f32", "rayen_ray_project_old_bwd_f64",
    "rayen_mapper_fusable", "rayen_ray_project_mapped_f32", "rayen_ray_project_bwd_generic_f32",
    "rayen_ray_project_bwd_generic_f64", "rayen_mapper_image_bytes", "rayen_mapper_prepare_f32",
    "rayen_ray_project_mapped_image_f32", "rayen_bwd_workspace_bytes_f32", "rayen_ray_project_bwd_ws_f32",
    "rayen_bwd_workspace_bytes_f64", "rayen_ray_project_bwd_ws_f64", "rayen_last_forward_kernel",
    "rayen_pair_schedule", "rayen_reserve_cus",
    "rayen_products_rows", "rayen_products_served", "rayen_ray_project_from_products_f32", "rayen_ray_project_from_products_f64",
    "rayen_ray_project_bwd_coefficients_f32", "rayen_ray_project_bwd_coefficients_f64",
)
KERNEL_NONE, KERNEL_LANE, KERNEL_MFMA, KERNEL_TRIPLE, KERNEL_PAIR, KERNEL_PAIR_IO, KERNEL_LMI_QUAD, KERNEL_LMI_WAVE, KERNEL_PAIR_WS, KERNEL_PRODUCTS, KERNEL_LMI_BLOCK, KERNEL_PAIR_WL = range(12)


class RayenSegment(ctypes.Structure):
    _fields_ = [("type", ctypes.c_int32), ("row0", ctypes.c_int32), ("nrows", ctypes.c_int32),
                ("aux_row", ctypes.c_int32), ("dim", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("f0", ctypes.c_double), ("f1", ctypes.c_double)]


class RayenPackDesc(ctypes.Structure):
    _fields_ = [("abi_version", ctypes.c_int32), ("k", ctypes.c_int32), ("n", ctypes.c_int32),
                ("n_rows", ctypes.c_int32), ("n_segments", ctypes.c_int32),
                ("out_identity", ctypes.c_int32),
                ("W", ctypes.POINTER(ctypes.c_double)),
                ("segments", ctypes.POINTER(RayenSegment)),
                ("NA_E", ctypes.POINTER(ctypes.c_double)),
                ("y0", ctypes.POINTER(ctypes.c_double)),
                ("prepare", ctypes.c_int32), ("fp32_mode", ctypes.c_int32)]


class RayenPackInfo(ctypes.Structure):
    _fields_ = [("k", ctypes.c_int32), ("n", ctypes.c_int32), ("n_rows", ctypes.c_int32),
                ("n_segments", ctypes.c_int32), ("device", ctypes.c_int32),
                ("mfma_f32", ctypes.c_int32), ("generic_block", ctypes.c_int32),
                ("mfma_f64", ctypes.c_int32), ("device_bytes", ctypes.c_int64),
                ("prepared", ctypes.c_int32), ("bwd_f32", ctypes.c_int32),
                ("fp32_check_split", ctypes.c_double), ("fp32_check_exact", ctypes.c_double),
                ("fp32_check_pair", ctypes.c_double),
                ("bwd32_check_pair", ctypes.c_double), ("bwd32_check_exact", ctypes.c_double)]


class RayenError(RuntimeError):
    def __init__(self, code, what):
        super().__init__(f"{what} failed with code {code}: {strerror(code)}")
        self.code = code


_lib = None


def library_path():
    return LIBRARY


def load():
    """Load ``librayen_hip.so`` (once) and declare the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIBRARY):
        raise RuntimeError(
            f"{LIBRARY} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). Tensors on a HIP device are never routed anywhere else.")
    lib = ctypes.CDLL(LIBRARY)
    p, i64, i32p = ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p
    lib.rayen_abi_version.restype = ctypes.c_int
    lib.rayen_abi_version.argtypes = []
    lib.rayen_last_forward_kernel.restype = ctypes.c_int
    lib.rayen_last_forward_kernel.argtypes = []
    lib.rayen_pair_schedule.restype = ctypes.c_int
    lib.rayen_pair_schedule.argtypes = [ctypes.c_int]
    lib.rayen_reserve_cus.restype = ctypes.c_int
    lib.rayen_reserve_cus.argtypes = [ctypes.c_int]
    lib.rayen_strerror.restype = ctypes.c_char_p
    lib.rayen_strerror.argtypes = [ctypes.c_int]
    lib.rayen_pack_create.restype = ctypes.c_int
    lib.rayen_pack_create.argtypes = [ctypes.POINTER(RayenPackDesc), ctypes.POINTER(ctypes.c_void_p)]
    lib.rayen_pack_destroy.restype = None
    lib.rayen_pack_destroy.argtypes = [p]
    lib.rayen_pack_info.restype = ctypes.c_int
    lib.rayen_pack_info.argtypes = [p, ctypes.POINTER(RayenPackInfo)]
    fwd = [p, p, i64, i64, p, i64, p, i32p, i32p, p]
    for name in ("rayen_ray_project_f32", "rayen_ray_project_f64", "rayen_ray_project_generic_f32",
                 "rayen_ray_project_generic_f64", "rayen_ray_project_old_f32", "rayen_ray_project_old_f64"):
        getattr(lib, name).restype = ctypes.c_int
        getattr(lib, name).argtypes = fwd
    for name in ("rayen_ray_project_from_products_f32", "rayen_ray_project_from_products_f64"):
        getattr(lib, name).restype = ctypes.c_int
        getattr(lib, name).argtypes = [p, p, i64, p, i64, i64, p, i64, p, i32p, i32p, p]
    for name in ("rayen_ray_project_bwd_coefficients_f32", "rayen_ray_project_bwd_coefficients_f64"):
        getattr(lib, name).restype = ctypes.c_int
        getattr(lib, name).argtypes = [p, p, i64, p, i64, i64, p, i32p, p, i64, p, i64, p, p]
    lib.rayen_products_rows.restype = ctypes.c_int64
    lib.rayen_products_rows.argtypes = [p]
    lib.rayen_products_served.restype = ctypes.c_int
    lib.rayen_products_served.argtypes = [p, ctypes.c_int]
    bwd = [p, p, i64, i64, p, i32p, p, i64, p, i64, p]
    for name in ("rayen_ray_project_bwd_f32", "rayen_ray_project_bwd_f64", "rayen_ray_project_bwd_generic_f32",
                 "rayen_ray_project_bwd_generic_f64",
                 "rayen_ray_project_old_bwd_f32", "rayen_ray_project_old_bwd_f64"):
        getattr(lib, name).restype = ctypes.c_int
        getattr(lib, name).argtypes = bwd
    lib.rayen_mapper_fusable.restype = ctypes.c_int
    lib.rayen_mapper_fusable.argtypes = [p, ctypes.c_int32]
    lib.rayen_ray_project_mapped_f32.restype = ctypes.c_int
    lib.rayen_ray_project_mapped_f32.argtypes = [p, p, i64, i64, ctypes.c_int32, p, i64, p, p, i64, p, i64,
                                                 p, i32p, i32p, p]
    for name in ("rayen_bwd_workspace_bytes_f32", "rayen_bwd_workspace_bytes_f64"):
        getattr(lib, name).restype = ctypes.c_int64
        getattr(lib, name).argtypes = [p, i64]
    for name in ("rayen_ray_project_bwd_ws_f32", "rayen_ray_project_bwd_ws_f64"):
        getattr(lib, name).restype = ctypes.c_int
        getattr(lib, name).argtypes = [p, p, i64, i64, p, i32p, p, i64, p, i64, p, i64, p]
    lib.rayen_mapper_image_bytes.restype = ctypes.c_int64
    lib.rayen_mapper_image_bytes.argtypes = [p, ctypes.c_int32]
    lib.rayen_mapper_prepare_f32.restype = ctypes.c_int
    lib.rayen_mapper_prepare_f32.argtypes = [p, p, i64, ctypes.c_int32, p, p, p]
    lib.rayen_ray_project_mapped_image_f32.restype = ctypes.c_int
    lib.rayen_ray_project_mapped_image_f32.argtypes = [p, p, i64, i64, ctypes.c_int32, p, p, i64, p, i64,
                                                       p, i32p, i32p, p]
    if lib.rayen_abi_version() != ABI_VERSION:
        raise RuntimeError(f"librayen_hip.so ABI {lib.rayen_abi_version()} != binding ABI {ABI_VERSION}")
    _lib = lib
    return lib


def strerror(code):
    return load().rayen_strerror(int(code)).decode()


def check(code, what):
    if code != 0:
        raise RayenError(code, what)
