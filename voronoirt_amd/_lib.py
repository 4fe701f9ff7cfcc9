"""Loader and ctypes prototypes of libvrt_hip.so (the C ABI of include/voronoirt.h).

There is no CPU fallback anywhere in this package: if the shared library is missing the import
fails loudly, and every compute call on a machine without a HIP device raises `VrtError`.
"""
from __future__ import annotations

import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libvrt_hip.so")
if os.environ.get("VRT_LIB_PATH"):      # host-only sanitizer build of the same sources (tools/asan_host.sh)
    LIB_PATH = os.environ["VRT_LIB_PATH"]

c_i64 = ctypes.c_int64
c_dbl = ctypes.c_double
p_i64 = ctypes.POINTER(ctypes.c_int64)
p_i32 = ctypes.POINTER(ctypes.c_int32)
p_dbl = ctypes.POINTER(ctypes.c_double)
p_int = ctypes.POINTER(ctypes.c_int)
vp = ctypes.c_void_p

VRT_OK, VRT_EINVAL, VRT_EGRID, VRT_ENODEVICE, VRT_ENOMEM, VRT_EIO = 0, -1, -2, -3, -4, -5
ALPHA_SITE, ALPHA_SITE_LAM, ALPHA_ANGLE_SITE_LAM, ALPHA_ANGLE_NATIVE, ALPHA_SITE_LAM_NATIVE = 0, 1, 2, 3, 4

class LineCaseStruct(ctypes.Structure):
    """vrt_line_case of include/voronoirt.h"""
    _fields_ = [("nlam", c_i64), ("lambda_", p_dbl), ("blocks", c_i64 * 6), ("lambda0", c_dbl), ("c0", c_dbl),
                ("velocity", p_dbl), ("doppler_width", p_dbl), ("gamma_static", p_dbl), ("gamma_unsold", p_dbl),
                ("alpha_cont", p_dbl), ("eps", p_dbl), ("temperature", p_dbl), ("atom_density", p_dbl), ("B0", p_dbl),
                ("lte_populations", p_dbl), ("C", p_dbl), ("planck2", p_dbl), ("sigma_bf1", p_dbl), ("sigma_bf2", p_dbl),
                ("strength_const", c_dbl), ("Bij", c_dbl), ("Bji", c_dbl), ("sigma_bb_const", c_dbl),
                ("hc_over_kB", c_dbl), ("pref_ij", c_dbl), ("pref_ji", c_dbl)]


# name -> (restype, argtypes): every symbol include/voronoirt.h declares
PROTOTYPES = {
    "vrt_last_error": (ctypes.c_char_p, []),
    "vrt_version": (ctypes.c_int, []),
    "vrt_device_count": (ctypes.c_int, []),
    "vrt_grid_create": (ctypes.c_int, [c_i64, p_dbl, p_i64, c_i64, p_dbl, ctypes.c_int,
                                       ctypes.POINTER(vp)]),
    "vrt_grid_create_from_file": (ctypes.c_int, [ctypes.c_char_p, c_i64, p_dbl, p_dbl,
                                                 ctypes.c_int, ctypes.POINTER(vp)]),
    "vrt_grid_destroy": (None, [vp]),
    "vrt_tessellate": (ctypes.c_int, [c_i64, p_dbl, p_dbl, c_i64, p_i64, p_i64]),
    "vrt_write_neighbours_file": (ctypes.c_int, [ctypes.c_char_p, c_i64, p_i64, c_i64]),
    "vrt_grid_n": (c_i64, [vp]),
    "vrt_grid_max_neighbours": (c_i64, [vp]),
    "vrt_grid_num_layer_offsets": (c_i64, [vp, ctypes.c_int]),
    "vrt_grid_get_layers": (ctypes.c_int, [vp, ctypes.c_int, p_i64]),
    "vrt_grid_get_perm": (ctypes.c_int, [vp, ctypes.c_int, p_i64]),
    "vrt_grid_get_delaunay_lines": (ctypes.c_int, [vp, p_dbl]),
    "vrt_direction": (None, [c_dbl, c_dbl, p_dbl]),
    "vrt_plan_create": (ctypes.c_int, [vp, c_i64, p_dbl, ctypes.c_int, ctypes.POINTER(vp)]),
    "vrt_plan_create_ex": (ctypes.c_int, [vp, c_i64, p_dbl, p_int, ctypes.c_int,
                                          ctypes.POINTER(vp)]),
    "vrt_plan_destroy": (None, [vp]),
    "vrt_plan_num_levels": (c_i64, [vp]),
    "vrt_plan_num_nodes": (c_i64, [vp]),
    "vrt_plan_get_upwind": (ctypes.c_int, [vp, c_i64, p_i64, p_dbl, p_dbl, p_dbl]),
    "vrt_plan_execute": (ctypes.c_int, [vp, c_i64, c_i64, p_dbl, p_dbl, ctypes.c_int, p_dbl, p_dbl,
                                        p_dbl, p_dbl, p_dbl]),
    "vrt_plan_execute_dev": (ctypes.c_int, [vp, c_i64, c_i64, vp, vp, ctypes.c_int, vp, vp, p_dbl,
                                            vp, vp, vp]),
    "vrt_plan_execute_dev_f32": (ctypes.c_int, [vp, c_i64, c_i64, vp, vp, ctypes.c_int, vp, vp, p_dbl,
                                                vp, vp, vp]),
    "vrt_plan_native_plane_count": (c_i64, [vp, c_i64]),
    "vrt_plan_to_native_dev": (ctypes.c_int, [vp, c_i64, c_i64, vp, vp, vp, vp]),
    "vrt_plan_from_native_dev": (ctypes.c_int, [vp, ctypes.c_int, c_i64, c_i64, vp, vp, vp]),
    "vrt_plan_j_from_native_dev": (ctypes.c_int, [vp, c_i64, c_i64, vp, vp, vp, vp]),
    "vrt_plan_execute_native_dev": (ctypes.c_int, [vp, c_i64, vp, vp, vp, ctypes.c_int, vp, vp, p_dbl, vp, vp, vp]),
    "vrt_plan_to_native_dev_f32": (ctypes.c_int, [vp, c_i64, c_i64, vp, vp, vp, vp]),
    "vrt_plan_from_native_dev_f32": (ctypes.c_int, [vp, ctypes.c_int, c_i64, c_i64, vp, vp, vp]),
    "vrt_plan_j_from_native_dev_f32": (ctypes.c_int, [vp, c_i64, c_i64, vp, vp, vp, vp]),
    "vrt_plan_execute_native_dev_f32": (ctypes.c_int, [vp, c_i64, vp, vp, vp, ctypes.c_int, vp, vp, p_dbl, vp, vp, vp]),
    "vrt_plan_check": (ctypes.c_int, [vp]),
    "vrt_lambda_update_native_dev": (ctypes.c_int, [vp, c_i64, vp, vp, vp, vp, vp, vp, p_dbl, vp]),
    "vrt_rates_populations_native_dev": (ctypes.c_int, [vp, c_i64, p_dbl, p_i64, vp, vp, p_dbl, c_dbl, c_dbl, vp, vp,
                                                        c_dbl, p_dbl, p_dbl, vp, vp, c_dbl, c_dbl, c_dbl, vp, vp, vp,
                                                        vp, vp]),
    "vrt_grid_get_storage_order": (ctypes.c_int, [vp, ctypes.c_int, p_i64]),
    "vrt_plan_native_alpha_count": (c_i64, [vp, c_i64]),
    "vrt_plan_native_pair_block": (ctypes.c_int, [vp]),
    "vrt_plan_native_pair_block_f32": (ctypes.c_int, [vp]),
    "vrt_plan_alpha_to_native_dev": (ctypes.c_int, [vp, c_i64, c_i64, vp, vp, vp]),
    "vrt_line_opacity_dev": (ctypes.c_int, [vp, c_i64, p_dbl, c_dbl, c_dbl, vp, vp, vp, vp, vp, vp, vp]),
    "vrt_line_opacity_dev_f32": (ctypes.c_int, [vp, c_i64, p_dbl, c_dbl, c_dbl, vp, vp, vp, vp, vp, vp, vp]),
    "vrt_plan_alpha_to_native_dev_f32": (ctypes.c_int, [vp, c_i64, c_i64, vp, vp, vp]),
    "vrt_plan_execute_line": (ctypes.c_int, [vp, c_i64, c_i64, p_dbl, c_dbl, c_dbl, p_dbl, p_dbl, p_dbl, p_dbl, p_dbl,
                                             p_dbl, p_dbl, p_dbl, p_dbl, p_dbl]),
    "vrt_line_terms_dev": (ctypes.c_int, [vp, vp, vp, vp, c_dbl, c_dbl, c_dbl, vp, vp, vp]),
    "vrt_lambda_create": (ctypes.c_int, [vp, ctypes.POINTER(LineCaseStruct), p_dbl, ctypes.POINTER(vp)]),
    "vrt_lambda_iterate": (ctypes.c_int, [vp, p_dbl]),
    "vrt_lambda_get": (ctypes.c_int, [vp, p_dbl, p_dbl, p_dbl, p_dbl, p_dbl]),
    "vrt_lambda_destroy": (None, [vp]),
    "vrt_multi_create": (ctypes.c_int, [ctypes.c_int, p_int, c_i64, p_dbl, p_i64, c_i64, p_dbl, c_i64, p_dbl, p_int,
                                        ctypes.c_int, ctypes.POINTER(vp)]),
    "vrt_multi_execute": (ctypes.c_int, [vp, c_i64, c_i64, p_dbl, p_dbl, ctypes.c_int, p_dbl, p_dbl, p_dbl, p_dbl]),
    "vrt_multi_set_shard": (ctypes.c_int, [vp, ctypes.c_char_p]),
    "vrt_multi_last_shard": (ctypes.c_int, [vp]),
    "vrt_multi_uses_rccl": (ctypes.c_int, [vp]),
    "vrt_multi_destroy": (None, [vp]),
    "vrt_multi_execute_line": (ctypes.c_int, [vp, c_i64, c_i64, p_dbl, c_dbl, c_dbl, p_dbl, p_dbl, p_dbl, p_dbl, p_dbl, p_dbl,
                                              p_dbl, p_dbl, p_dbl, p_dbl]),
    "vrt_multi_lambda_create": (ctypes.c_int, [vp, ctypes.POINTER(LineCaseStruct), p_dbl, ctypes.POINTER(vp)]),
    "vrt_multi_lambda_iterate": (ctypes.c_int, [vp, p_dbl]),
    "vrt_multi_lambda_get": (ctypes.c_int, [vp, p_dbl, p_dbl, p_dbl, p_dbl, p_dbl]),
    "vrt_multi_lambda_destroy": (None, [vp]),
    "vrt_rates_populations_dev": (ctypes.c_int, [vp, c_i64, c_i64, p_dbl, p_i64, vp, p_dbl, c_dbl, c_dbl, vp, vp,
                                                 c_dbl, p_dbl, p_dbl, vp, vp, c_dbl, c_dbl, c_dbl, vp, vp, vp,
                                                 vp, vp]),
    "vrt_plan_last_sweep_timing": (ctypes.c_int, [vp, p_dbl, p_i64]),
    "vrt_plan_last_path": (ctypes.c_int, [vp]),
    "vrt_plan_set_option": (ctypes.c_int, [vp, ctypes.c_char_p, ctypes.c_char_p]),
    "vrt_grid_set_option": (ctypes.c_int, [vp, ctypes.c_char_p, ctypes.c_char_p]),
    "vrt_schedule_build": (ctypes.c_int, [vp, ctypes.c_int, p_i64, ctypes.c_int,
                                          ctypes.POINTER(vp)]),
    "vrt_schedule_num_nodes": (c_i64, [vp]),
    "vrt_schedule_num_levels": (c_i64, [vp]),
    "vrt_schedule_get": (ctypes.c_int, [vp, p_i64, p_i32, p_i64]),
    "vrt_schedule_destroy": (None, [vp]),
    "vrt_layer_schedule": (ctypes.c_int, [vp, ctypes.c_int, p_i64, ctypes.c_int,
                                          ctypes.POINTER(ctypes.c_uint32), p_i32, p_i64]),
    "vrt_layer_sorted_slots": (ctypes.c_int, [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_uint32), p_i64, p_i64]),
    "vrt_patch_schedule_build": (ctypes.c_int, [vp, ctypes.c_int, p_i64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                ctypes.POINTER(vp), p_i64]),
    "vrt_patch_schedule_get": (ctypes.c_int, [vp, p_i32, p_i32, p_i32, p_i32, p_i64, p_i32,
                                              ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]),
    "vrt_patch_schedule_get_deps": (ctypes.c_int, [vp, p_i64, p_i32]),
    "vrt_patch_schedule_get_layers": (ctypes.c_int, [vp, ctypes.POINTER(ctypes.c_uint32), p_i32, p_i64]),
    "vrt_patch_schedule_destroy": (None, [vp]),
    "vrt_lambda_update_dev": (ctypes.c_int, [vp, c_i64, c_i64, vp, vp, vp, vp, vp, p_dbl, vp]),
    "vrt_short_characteristics": (ctypes.c_int, [c_i64, c_i64, c_i64, p_dbl, p_dbl, p_dbl, c_i64, p_dbl,
                                                 p_int, p_dbl, c_i64, p_dbl, c_i64, p_dbl, ctypes.c_int,
                                                 ctypes.c_int, p_dbl]),
    "vrt_regular_create": (ctypes.c_int, [c_i64, c_i64, c_i64, p_dbl, p_dbl, p_dbl, ctypes.c_int,
                                          ctypes.POINTER(vp)]),
    "vrt_regular_destroy": (None, [vp]),
    "vrt_regular_execute_dev": (ctypes.c_int, [vp, c_i64, p_dbl, p_int, vp, c_i64, vp, c_i64, c_i64, vp,
                                               ctypes.c_int, vp, vp]),
    "vrt_regular_last_solve_ms": (ctypes.c_int, [vp, p_dbl]),
    "vrt_delaunay_up": (ctypes.c_int, [vp, p_dbl, p_dbl, p_dbl, c_i64, p_dbl, ctypes.c_int, p_dbl]),
    "vrt_delaunay_down": (ctypes.c_int, [vp, p_dbl, p_dbl, p_dbl, c_i64, p_dbl, ctypes.c_int, p_dbl]),
}


class VrtError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libvrt_hip error {code}: {message}")
        self.code = code
        self.message = message


_lib = None


def load() -> ctypes.CDLL:
    """dlopen libvrt_hip.so.  torch (if installed) is imported first so that this library and
    torch share ONE HIP runtime (both resolve the soname libamdhip64.so.7; whichever is loaded
    first wins, and torch must get its own bundled copy)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m voronoirt_amd.build` "
            "(hipcc --offload-arch=gfx950).  voronoirt_amd has no CPU fallback.")
    if os.environ.get("VRT_NO_TORCH") != "1":     # (the host-only sanitizer drivers: torch under TSan takes minutes to import)
        try:
            import torch  # noqa: F401  (plumbing only: device memory, streams, torch.distributed)
        except Exception:  # pragma: no cover - torch is optional for the C ABI itself
            pass
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        msg = load().vrt_last_error()
        raise VrtError(rc, msg.decode() if msg else "")
