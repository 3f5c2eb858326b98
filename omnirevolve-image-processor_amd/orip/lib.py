"""ctypes binding of liborip.so (include/orip.h).  Fails loudly when the HIP library is missing: the
product has no CPU path (the CPU restatement under oracle/ is test infrastructure and is never imported here)."""
from __future__ import annotations

import ctypes as C
import os

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(_PKG, "liborip.so")
# ORIP_LIB_VARIANTS=1: the variants build (`make -C csrc variants`: the same sources plus the replaced kernel variants, for the agreement tests)
if os.environ.get("ORIP_LIB_VARIANTS") == "1" and os.path.exists(os.path.join(_PKG, "liborip_variants.so")):
    LIB_PATH = os.path.join(_PKG, "liborip_variants.so")

# One hardware queue per lane (layer pipelines, raster stages, stage 10): HIP multiplexes its streams onto GPU_MAX_HW_QUEUES
# (default 4) hardware queues and kernels sharing a queue run one after the other, so a short kernel of one layer would wait
# behind a long walk of another.  Must be in the environment before the HIP runtime initialises (bench.py sets it before torch).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

MAX_LAYERS = 16
SLOT_CONTOURS, SLOT_SCALED, SLOT_SORTED, SLOT_LINES_INTRA, SLOT_LINES_CROSS = range(5)
TAPS_INTRA, TAPS_CROSS = 0, 1


class Params08(C.Structure):
    _fields_ = [("tap_diam", C.c_double), ("tap_max_dim", C.c_double), ("min_keep", C.c_double), ("tap_max_per", C.c_double),
                ("tap_max_v", C.c_int32), ("sample_step", C.c_double), ("tail_len_px", C.c_double), ("col_rad", C.c_double),
                ("grid_stride", C.c_double), ("max_jump", C.c_double), ("post_on", C.c_int32), ("post_brush", C.c_int32),
                ("post_step", C.c_double), ("post_eps", C.c_double), ("post_minlen", C.c_int32), ("W", C.c_int32), ("H", C.c_int32),
                ("brush_forbid", C.c_int32)]


class Params10(C.Structure):
    _fields_ = [("tap_diam", C.c_double), ("min_keep", C.c_double), ("tap_max_per", C.c_double), ("tap_max_v", C.c_int32),
                ("max_jump", C.c_double), ("D_lines", C.c_double), ("D_taps", C.c_double), ("step_px", C.c_double),
                ("W", C.c_int32), ("H", C.c_int32)]


_vp, _i64, _i32, _f64, _f32, _cp = C.c_void_p, C.c_int64, C.c_int, C.c_double, C.c_float, C.c_char_p
_P = C.POINTER

# name -> (restype, argtypes); every symbol include/orip.h declares
SIGNATURES = {
    "orip_create": (_i32, [_i32, _P(_vp)]), "orip_destroy": (None, [_vp]), "orip_last_error": (_cp, [_vp]), "orip_sync": (_i32, [_vp]),
    "orip_prof_reset": (_i32, [_vp]), "orip_prof_get": (_i32, [_vp, _cp, _P(_f64), _P(_i64)]), "orip_prof_enable": (_i32, [_vp, _i32]),
    "orip_resize_area": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _i32]),
    "orip_set_image": (_i32, [_vp, _vp, _i32, _i32]), "orip_lab_of": (_i32, [_vp, _vp, _i64, _vp]),
    "orip_kmeans_fit": (_i32, [_vp, _vp, _i64, _i32, _i32, _i32, _f64, _vp, _P(_f64)]),
    "orip_kmeans_fit_rgb": (_i32, [_vp, _vp, _i64, _i32, _i32, _i32, _f64, _vp, _P(_f64)]), "orip_assign_palette": (_i32, [_vp, _vp, _i32, _vp, _vp]),
    "orip_extract_layers": (_i32, [_vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "orip_get_labels": (_i32, [_vp, _vp]), "orip_get_mask": (_i32, [_vp, _i32, _vp]), "orip_set_masks": (_i32, [_vp, _vp, _i32, _i32, _i32]),
    "orip_keep_layers": (_i32, [_vp, _vp, _i32]),
    "orip_detect_edges": (_i32, [_vp, _i32, _i32, _i32, _i32, _i32, _i32]),
    "orip_get_edges": (_i32, [_vp, _i32, _vp]), "orip_set_edges": (_i32, [_vp, _vp, _i32, _i32, _i32]),
    "orip_find_contours": (_i32, [_vp]), "orip_contours_prepare": (_i32, [_vp]), "orip_contours_reserve": (_i32, [_vp, _i32]), "orip_contours_layer": (_i32, [_vp, _i32]),
    "orip_dedup_cross_begin": (_i32, [_vp, _P(Params10)]), "orip_dedup_cross_layer": (_i32, [_vp, _i32]), "orip_dedup_cross_layer_from": (_i32, [_vp, _i32, _i32]), "orip_dedup_cross_layer_deferred": (_i32, [_vp, _i32, _i32]), "orip_get_skeleton": (_i32, [_vp, _i32, _vp]),
    "orip_polys_size": (_i32, [_vp, _i32, _i32, _P(_i64), _P(_i64)]), "orip_get_polys": (_i32, [_vp, _i32, _i32, _vp, _vp]),
    "orip_set_polys": (_i32, [_vp, _i32, _i32, _i64, _vp, _vp]),
    "orip_taps_size": (_i32, [_vp, _i32, _i32, _P(_i64)]), "orip_get_taps": (_i32, [_vp, _i32, _i32, _vp]),
    "orip_set_taps": (_i32, [_vp, _i32, _i32, _i64, _vp]), "orip_set_layer_count": (_i32, [_vp, _i32]),
    "orip_scale_vectors": (_i32, [_vp, _i32, _f32, _f32, _f32, _f32]), "orip_sort_contours": (_i32, [_vp, _i32]),
    "orip_dedup_layer": (_i32, [_vp, _i32, _P(Params08)]), "orip_layer_front": (_i32, [_vp, _i32, _f32, _f32, _f32, _f32, _i32, _vp]), "orip_dedup_cross": (_i32, [_vp, _vp, _i32, _P(Params10)]),
    "orip_plot_order": (_i32, [_vp, _i32, _f64, _P(_i64)]), "orip_get_ops": (_i32, [_vp, _i32, _vp]),
    "orip_preview_cover": (_i32, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "orip_has_variants": (_i32, []),
    "orip_stream_codes": (_i32, [_vp, _vp, _i64, _P(_i64)]), "orip_stream_codes_fetch": (_i32, [_vp, _vp, _vp]),
    "orip_comm_unique_id": (_i32, [_vp]), "orip_comm_init": (_i32, [_vp, _vp, _i32, _i32]), "orip_comm_destroy": (_i32, [_vp]),
    "orip_bcast_layer": (_i32, [_vp, _i32, _i32]),
}
COMM_ID_BYTES = 128

_lib = None


def load():
    """Load liborip.so and declare every entry point.  Raises (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"liborip.so not found at {LIB_PATH}: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)   # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def has_variants() -> bool:
    """True when the loaded library carries the replaced kernel variants (include/orip.h: orip_has_variants)."""
    return bool(load().orip_has_variants())
