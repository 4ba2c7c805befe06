"""ctypes binding of libgsplat_hip.so -- the C-ABI declared in include/gsplat.h.

The library is the product: if it is missing or fails to load this module raises
(there is no CPU or PyTorch fallback on this path).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GSPLAT_LIB: another build of the same library (kernel experiments); default = the in-tree build
LIB_PATH = os.environ.get("GSPLAT_LIB") or os.path.join(_HERE, "lib", "libgsplat_hip.so")

_vp, _i, _ll, _f, _sz, _d = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_size_t, C.c_double


class GsAdamGroup(C.Structure):
    """struct GsAdamGroup of include/gsplat.h"""
    _fields_ = [("param", _vp), ("grad", _vp), ("exp_avg", _vp), ("exp_avg_sq", _vp), ("rows", _ll),
                ("grad_row_stride", _ll), ("row_len", _i), ("reserved", _i), ("lr", _d)]


class GsPlyLayout(C.Structure):
    """struct GsPlyLayout of include/gsplat.h"""
    _fields_ = [("num_vertices", _ll), ("data_offset", _ll), ("stride_bytes", _i), ("sh_rest_count", _i),
                ("off_xyz", _i * 3), ("off_f_dc", _i * 3), ("off_opacity", _i), ("off_scale", _i * 3),
                ("off_rot", _i * 4), ("off_f_rest", _i * 72)]


class GsBwdView(C.Structure):
    """struct GsBwdView of include/gsplat.h (one view of gs_preprocess_bwd_views)"""
    _fields_ = [("viewmatrix", _vp), ("projmatrix", _vp), ("campos", _vp), ("radii", _vp), ("grad_rows", _vp),
                ("sh_jac", _vp), ("g_means2D", _vp), ("H", C.c_int32), ("W", C.c_int32)]


class GsFwdView(C.Structure):
    """struct GsFwdView of include/gsplat.h (one view of gs_preprocess_fwd_views)"""
    _fields_ = [("viewmatrix", _vp), ("projmatrix", _vp), ("campos", _vp), ("splat", _vp), ("radii", _vp),
                ("tiles_touched", _vp), ("depth_keys", _vp), ("ids", _vp), ("rects", _vp), ("sh_jac", _vp),
                ("H", C.c_int32), ("W", C.c_int32), ("tile_row_begin", C.c_int32), ("tile_row_end", C.c_int32)]


MAX_BWD_VIEWS = 8      # GS_MAX_BWD_VIEWS: views per launch of gs_preprocess_bwd_views (more are taken in chunks)

# name -> (restype, argtypes); mirrors include/gsplat.h one to one
PROTOTYPES = {
    "gs_version": (_i, []),
    "gs_last_error": (C.c_char_p, []),
    "gs_source_sha": (C.c_char_p, []),
    "gs_tile_size": (_i, []),
    "gs_grad_row_floats": (_i, []),
    "gs_sh_jac_row_floats": (_i, []),
    "gs_preprocess_fwd": (_i, [_i, _i, _i, _i, _i] + [_vp] * 16 + [_i, _i, _vp, _vp]),
    "gs_preprocess_bwd": (_i, [_i, _i, _i, _i, _i] + [_vp] * 20 + [_i, _vp, _vp]),
    "gs_preprocess_fwd_views": (_i, [_i, C.POINTER(GsFwdView), _i, _i, _i] + [_vp] * 7 + [_vp]),
    "gs_preprocess_bwd_views": (_i, [_i, C.POINTER(GsBwdView), _i, _i, _i] + [_vp] * 14 + [_i, _vp]),
    "gs_scan_workspace_bytes": (_sz, [_ll]),
    "gs_scan_tiles": (_i, [_ll, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "gs_duplicate_with_keys": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _ll, _i, _i, _vp]),
    "gs_sort_workspace_bytes": (_sz, [_ll]),
    "gs_sort_pairs": (_i, [_ll, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    "gs_sort_pairs_static_max": (_ll, []),
    "gs_sort_pairs_static": (_i, [_ll, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gs_identify_tile_ranges": (_i, [_ll, _vp, _i, _vp, _vp, _vp]),
    "gs_render_fwd": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _i, _i, _vp, _sz, _vp]),
    "gs_l1_ssim_workspace_bytes": (_sz, [_i, _i, _i]),
    "gs_l1_ssim": (_i, [_i, _i, _i, _vp, _vp, _f, _f, _vp, _vp, _vp, _sz, _vp]),
    "gs_render_bwd": (_i, [_i, _i] + [_vp] * 10 + [_f, _i, _i, _vp]),
    "gs_ply_parse_header": (_i, [_vp, _sz, _sz, C.POINTER(GsPlyLayout)]),
    "gs_ply_write_header": (_ll, [_vp, _sz, _ll, _i, C.POINTER(GsPlyLayout)]),
    "gs_ply_unpack": (_i, [C.POINTER(GsPlyLayout)] + [_vp] * 12),
    "gs_ply_pack": (_i, [C.POINTER(GsPlyLayout)] + [_vp] * 8),
    "gs_nearest_dist2_workspace_bytes": (_sz, [_ll]),
    "gs_nearest_dist2": (_i, [_ll, _vp, _vp, _vp, _sz, _vp, _vp]),
    "gs_adam_densify_step": (_i, [_i, C.POINTER(GsAdamGroup), _d, _d, _d, _ll, _ll, _i] + [_vp] * 6),
}

_lib = None
ABI_VERSION = 303     # GS_ABI_VERSION of the include/gsplat.h these prototypes were written against


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C gaussian-splatting_amd/csrc` (hipcc --offload-arch=gfx950). "
            "This back-end has no CPU/PyTorch fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    got = int(lib.gs_version())
    if got != ABI_VERSION:
        raise ImportError(f"{LIB_PATH} reports ABI version {got}, this binding was written against {ABI_VERSION} "
                          "(include/gsplat.h GS_ABI_VERSION): rebuild the library from this tree")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().gs_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed ({rc}): {msg}")
