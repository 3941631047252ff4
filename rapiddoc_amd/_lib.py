"""ctypes binding of librapiddoc_mi355.so (include/rapiddoc_mi355.h).  No fallback: if the library is
missing or no MI355X is visible, importing callers get a loud error."""
from __future__ import annotations

import ctypes as C
from pathlib import Path

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "librapiddoc_mi355.so"
_lib = None

SYMBOLS = {
    "rd_version": (C.c_char_p, []),
    "rd_create": (C.c_void_p, [C.c_int, C.c_char_p]),
    "rd_create_error": (C.c_char_p, []),
    "rd_destroy": (None, [C.c_void_p]),
    "rd_last_error": (C.c_char_p, [C.c_void_p]),
    "rd_load_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "rd_query_workspace": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    "rd_det_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rd_rec_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rd_rec_num_classes": (C.c_int, [C.c_void_p]),
    "rd_rec_token_dim": (C.c_int, [C.c_void_p]),
    "rd_rec_backbone_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rd_rec_backbone_forward_lines": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rd_rec_tail_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_size_t, C.c_void_p]),
    "rd_rec_seq_len": (C.c_int, [C.c_int]),
    "rd_backbone_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.c_void_p, C.c_size_t, C.c_void_p]),
    "rd_formula_encoder_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rd_formula_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int32), C.c_void_p]),
    "rd_formula_max_new_tokens": (C.c_int, [C.c_void_p]),
    "rd_preproc_resize_norm": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "rd_preproc_resize_norm_batch": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "rd_crop_resize_norm_batch": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_int, C.c_void_p, C.c_void_p]),
    "rd_line_crops_batch": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "rd_line_warp_batch": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]),
    "rd_line_resize_norm_batch": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "rd_ctc_collapse": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "rd_ctc_collapse_lines": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                        C.c_void_p]),
    "rd_db_postprocess": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "rd_db_boxes_workspace": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "rd_db_boxes_device": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int,
                                     C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "rd_rec_chunk_cost": (C.c_double, [C.c_int, C.c_int, C.c_int]),
    "rd_rec_plan_chunks": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "rd_layout_postprocess": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rd_layout_postprocess_select": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rd_find_external_contours": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "rd_contour_area": (C.c_double, [C.c_void_p, C.c_int]),
    "rd_arc_length": (C.c_double, [C.c_void_p, C.c_int, C.c_int]),
    "rd_approx_poly_dp": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p]),
    "rd_min_area_rect_points": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "rd_polygon_area": (C.c_double, [C.c_void_p, C.c_int]),
    "rd_polygon_intersection_area": (C.c_double, [C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "rd_fill_poly": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]),
    "rd_msdeform_attn": (C.c_int, [C.c_int] + [C.c_void_p] * 6 + [C.c_int] * 7 + [C.c_void_p]),
    "rd_topk_rows": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rd_encoder_layer_workspace": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "rd_encoder_layer": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p] + [C.c_int] * 6 + [C.c_void_p] * 12 + [C.c_float, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rd_set_precision": (C.c_int, [C.c_void_p, C.c_char_p]),
    "rd_range_status": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rd_plan_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "rd_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "rd_profile_json": (C.c_char_p, [C.c_void_p]),
}


class NativeLibraryError(RuntimeError):
    pass


def load():
    """Load the HIP library (once).  Raises NativeLibraryError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise NativeLibraryError(
            f"{LIB_PATH} is missing - build it with `python -m rapiddoc_amd.build` (hipcc, gfx950). "
            "rapiddoc_amd has no CPU fallback.")
    # PyTorch-ROCm bundles its own HIP runtime; it must be in the process BEFORE this library is opened so that
    # both resolve to ONE libamdhip64 (shared streams / device memory).  Opening ours first would pull in
    # /opt/rocm's copy and leave the process with two runtimes.
    import torch  # noqa: F401
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError = a symbol of include/rapiddoc_mi355.h is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
