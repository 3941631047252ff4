"""Polygon branch of the layout post-process: the detector's instance masks -> `polygon_points` per box, and the polygon-masked
region crop.  Host mirror of

    extract_polygon_points_by_masks   rapid_doc/model/layout/rapid_layout_self/model_handler/pp_doclayout/post_process.py:425-535
    mask2polygon / extract_custom_vertices                                                                        :261-423
    convert_polygon_to_quad                                                                                       :536-563
    calculate_polygon_overlap_ratio                                                                               :681-719
    crop_img's polygon mask           rapid_doc/utils/model_utils.py:109-118

The reference makes its OpenCV / shapely calls from Python; here each of those calls is one C function of the library
(rapiddoc_amd/csrc/polygon_ops.cpp, declared in include/rapiddoc_mi355.h) and the numpy arithmetic around them stays numpy, written
with the same numpy calls in the same order so that a vertex angle or a direction vector is the same float64 the reference gets
(np.linalg.norm / np.dot go through BLAS, whose rounding a hand-vectorised form would not reproduce).

Pinned: everything in this file, against the reference's own functions run with the library's primitives standing in for `cv2` and
`shapely` (tests/golden/make_golden_polygon.py -> tests/test_layout_polygon.py).  Unpinned: the primitives themselves (cv2 / shapely
are not installed here); they have known-answer tests (tests/test_polygon_ops.py).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np

from . import _lib

# ---------------------------------------------------------------------------------------------------------------------
# the primitives (one C call each)
# ---------------------------------------------------------------------------------------------------------------------


def _pts_i32(pts) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(pts).reshape(-1, 2), dtype=np.int32)


def find_external_contours(mask: np.ndarray) -> List[np.ndarray]:
    """cv2.findContours(mask, RETR_EXTERNAL, CHAIN_APPROX_SIMPLE)[0] as a list of int32 [n, 2] arrays."""
    lib = _lib.load()
    m = np.ascontiguousarray(mask, dtype=np.uint8)
    assert m.ndim == 2
    h, w = m.shape
    if h == 0 or w == 0:
        return []
    n_c, n_p = C.c_int32(0), C.c_int32(0)
    cap_p, cap_c = 4 * (h + w) + 64, 64
    while True:
        pts = np.empty((cap_p, 2), np.int32)
        counts = np.empty(cap_c, np.int32)
        rc = lib.rd_find_external_contours(m.ctypes.data, h, w, pts.ctypes.data, cap_p, counts.ctypes.data, cap_c, C.byref(n_c), C.byref(n_p))
        if rc == 0:
            break
        if rc != 2:
            raise ValueError("rd_find_external_contours: bad arguments")
        cap_p, cap_c = max(cap_p, n_p.value), max(cap_c, n_c.value)
    out, o = [], 0
    for c in counts[:n_c.value].tolist():
        out.append(pts[o:o + c].copy())
        o += c
    return out


def contour_area(contour) -> float:
    p = _pts_i32(contour)
    return float(_lib.load().rd_contour_area(p.ctypes.data, len(p)))


def arc_length(contour, closed: bool) -> float:
    p = _pts_i32(contour)
    return float(_lib.load().rd_arc_length(p.ctypes.data, len(p), int(bool(closed))))


def approx_poly_dp(contour, epsilon: float, closed: bool) -> np.ndarray:
    p = _pts_i32(contour)
    out = np.empty((max(len(p), 1), 2), np.int32)
    n = C.c_int32(0)
    if _lib.load().rd_approx_poly_dp(p.ctypes.data, len(p), float(epsilon), int(bool(closed)), out.ctypes.data, C.byref(n)) != 0:
        raise ValueError("rd_approx_poly_dp: bad arguments")
    return out[:n.value].copy()


def min_area_rect_points(points) -> np.ndarray:
    """cv2.boxPoints(cv2.minAreaRect(points)): float32 [4, 2]."""
    p = np.ascontiguousarray(np.asarray(points, dtype=np.float32).reshape(-1, 2))
    out = np.empty((4, 2), np.float32)
    if _lib.load().rd_min_area_rect_points(p.ctypes.data, len(p), out.ctypes.data) != 0:
        raise ValueError("rd_min_area_rect_points: bad arguments")
    return out


def polygon_area(poly) -> float:
    p = np.ascontiguousarray(np.asarray(poly, dtype=np.float64).reshape(-1, 2))
    return float(_lib.load().rd_polygon_area(p.ctypes.data, len(p)))


def polygon_intersection_area(a, b) -> float:
    pa = np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1, 2))
    pb = np.ascontiguousarray(np.asarray(b, dtype=np.float64).reshape(-1, 2))
    return float(_lib.load().rd_polygon_intersection_area(pa.ctypes.data, len(pa), pb.ctypes.data, len(pb)))


def fill_poly(mask: np.ndarray, polygon, value: int = 1) -> np.ndarray:
    """cv2.fillPoly(mask, [polygon], value) in place on a C-contiguous u8 [h, w] array."""
    assert mask.dtype == np.uint8 and mask.ndim == 2 and mask.flags.c_contiguous
    p = _pts_i32(polygon)
    if mask.size and _lib.load().rd_fill_poly(mask.ctypes.data, mask.shape[0], mask.shape[1], p.ctypes.data, len(p), int(value)) != 0:
        raise ValueError("rd_fill_poly: bad arguments")
    return mask


def resize_nearest(img: np.ndarray, width: int, height: int) -> np.ndarray:
    """cv2.resize(img, (width, height), interpolation=cv2.INTER_NEAREST): source index floor(dst * src / dst_size), clamped."""
    sh, sw = img.shape[:2]
    xs = np.minimum(np.floor(np.arange(width) * (1.0 / (width / sw))).astype(np.int64), sw - 1)
    ys = np.minimum(np.floor(np.arange(height) * (1.0 / (height / sh))).astype(np.int64), sh - 1)
    return np.ascontiguousarray(img[ys][:, xs])


# ---------------------------------------------------------------------------------------------------------------------
# overlap of two polygons (post_process.py:681-719); shapely's union of two valid polygons = area + area - intersection
# ---------------------------------------------------------------------------------------------------------------------
def polygon_overlap_ratio(polygon1, polygon2, mode: str = "union") -> float:
    a1, a2 = polygon_area(polygon1), polygon_area(polygon2)
    inter = polygon_intersection_area(polygon1, polygon2)
    if mode == "union":
        return inter / (a1 + a2 - inter)
    if mode == "small":
        return inter / min(a1, a2)
    if mode == "large":
        return inter / max(a1, a2)
    raise ValueError(f"Unknown mode: {mode}")


# ---------------------------------------------------------------------------------------------------------------------
# vertex selection on the approximated contour (post_process.py:261-396)
# ---------------------------------------------------------------------------------------------------------------------
def custom_vertices(polygon, max_allowed_dist, sharp_angle_thresh: float = 45, max_dist_ratio: float = 0.3) -> list:
    """Keeps the convex vertices, and the concave ones that sit in a run of >= 2 neighbouring concave vertices and open wider than
    120 degrees; long gaps between kept vertices get evenly spaced intermediate vertices back; a convex vertex of ~45 degrees is
    pushed outwards along its bisector.  Returns a list of (x, y) tuples like the reference."""
    poly = np.array(polygon)
    n = len(poly)
    limit = max_allowed_dist * max_dist_ratio
    convex, angle, arms = [], [], []
    for i in range(n):
        prev, cur, nxt = poly[(i - 1) % n], poly[i], poly[(i + 1) % n]
        e_in, e_out = cur - prev, nxt - cur
        convex.append(e_in[0] * e_out[1] - e_in[1] * e_out[0] < 0)
        v1, v2 = prev - cur, nxt - cur
        u1, u2 = v1 / np.linalg.norm(v1), v2 / np.linalg.norm(v2)
        angle.append(np.degrees(np.arccos(np.clip(np.dot(u1, u2), -1.0, 1.0))))
        arms.append((v1, v2))

    concave = [i for i in range(n) if not convex[i]]
    in_runs: List[int] = []                             # members of runs of >= 2 consecutive concave indices
    if concave:
        run = [concave[0]]
        for a, b in zip(concave, concave[1:]):
            if b - a == 1:
                run.append(b)
            else:
                if len(run) >= 2:
                    in_runs += run
                run = [b]
        if len(run) >= 2:
            in_runs += run
    preserved = set()
    if len(concave) >= 2 and concave[0] == 0 and concave[-1] == n - 1:
        # concave at both ends of the index range: the reference keeps the runs only if both ends are themselves in a run
        if 0 in in_runs and n - 1 in in_runs:
            preserved.update(in_runs)
    else:
        preserved.update(in_runs)

    kept = [i for i in range(n) if convex[i] or (i in preserved and angle[i] >= 120)]
    chosen: List[int] = []
    for k, cur in enumerate(kept):
        nxt = kept[(k + 1) % len(kept)]
        chosen.append(cur)
        dist = np.linalg.norm(poly[cur] - poly[nxt])
        if dist > limit:
            between = list(range(cur + 1, nxt)) if nxt > cur else list(range(cur + 1, n)) + list(range(0, nxt))
            if between:
                need = int(np.ceil(dist / limit)) - 1
                if len(between) <= need:
                    chosen += between
                else:
                    step = len(between) / need
                    chosen += [between[int(j * step)] for j in range(need)]

    out = []
    for i in sorted(set(chosen)):
        if convex[i] and abs(angle[i] - sharp_angle_thresh) < 1:
            v1, v2 = arms[i]
            direction = v1 / np.linalg.norm(v1) + v2 / np.linalg.norm(v2)
            direction /= np.linalg.norm(direction)
            reach = (np.linalg.norm(v1) + np.linalg.norm(v2)) / 2
            out.append(tuple(poly[i] + direction * reach))
        else:
            out.append(tuple(poly[i]))
    return out


def mask_to_polygon(mask: np.ndarray, max_allowed_dist, epsilon_ratio: float = 0.004, extract_custom: bool = True):
    """post_process.py:399-423: largest external contour -> approxPolyDP at 0.4 % of its perimeter -> custom_vertices."""
    contours = find_external_contours(mask)
    if not contours:
        return None
    largest = max(contours, key=contour_area)
    approx = approx_poly_dp(largest, epsilon_ratio * arc_length(largest, True), True)
    points = np.atleast_2d(approx)
    return custom_vertices(points, max_allowed_dist) if extract_custom else points


def polygon_to_quad(polygon) -> Optional[np.ndarray]:
    """post_process.py:536-563: min-area rectangle, corners ordered by angle around their centre, rolled to start at the corner
    with the smallest x + y."""
    if polygon is None or len(polygon) < 3:
        return None
    pts = np.array(polygon, dtype=np.float32)
    if pts.ndim == 1:
        pts = pts.reshape(-1, 2)
    quad = min_area_rect_points(pts)
    centre = quad.mean(axis=0)
    quad = quad[np.argsort(np.arctan2(quad[:, 1] - centre[1], quad[:, 0] - centre[0]))]
    return np.roll(quad, -int(np.argmin(quad[:, 0] + quad[:, 1])), axis=0)


def _mask_windows(boxes: np.ndarray, mask_hw: Sequence[int], scale_ratio: Sequence[float]):
    """All boxes at once: integer page rectangles [n, 4] (x0, y0, x1, y1: the float corners truncated), and the window of the detector's
    mask grid under each (columns c0:c1, rows r0:r1) - page pixels times input size / page size / 4, rounded half to even like the
    reference's round(), clipped to the grid."""
    hm, wm = mask_hw
    page = boxes[:, 2:6].astype(np.int32)
    fx, fy = scale_ratio[0] / 4, scale_ratio[1] / 4
    cols = np.clip(np.rint(page[:, [0, 2]] * fx).astype(np.int64), 0, wm)
    rows = np.clip(np.rint(page[:, [1, 3]] * fy).astype(np.int64), 0, hm)
    return page, cols, rows


def _auto_shape(polygon, quad, rect: np.ndarray, previous):
    """'auto' mode for one box whose mask gave a polygon (post_process.py:499-529): the min-area quad - or the box itself when that
    quad covers >= 95 % of box U quad - is taken when it agrees with the polygon (IoU >= 0.8, always measured against the min-area
    quad) and the box barely touches what the PREVIOUS box returned (< 1 % of the smaller of the two); else the polygon."""
    if quad is None:
        return polygon
    min_quad = quad.tolist()
    candidate = rect if polygon_overlap_ratio(rect.tolist(), min_quad, mode="union") >= 0.95 else quad
    agrees = polygon_overlap_ratio(polygon.tolist() if isinstance(polygon, np.ndarray) else polygon, min_quad, mode="union") >= 0.8
    clear = previous is None or polygon_overlap_ratio(previous.tolist(), rect.tolist(), mode="small") < 0.01
    return candidate if agrees and clear else polygon


def polygons_from_masks(boxes: np.ndarray, masks: np.ndarray, scale_ratio: Sequence[float], layout_shape_mode: str) -> list:
    """post_process.py:425-533 (`extract_polygon_points_by_masks`).  boxes [n, 6] float32 (page pixels), masks [n, hm, wm] (the
    detector's mask grid, 1/4 of its input size), scale_ratio = input size / page size per axis.  One entry per box: the box rectangle
    (float32 [4, 2]) where no usable mask exists or in 'rect' mode, else the polygon / quad, or None.

    Own structure (round 6): the box -> mask-window arithmetic runs over all boxes at once (`_mask_windows`), the boxes that have
    something to trace are known before the loop, and the per-box decision of the 'auto' mode is `_auto_shape`; the loop itself stays
    sequential because that decision looks at what the previous box returned.  One quirk of the reference is part of its output and kept:
    the width that `custom_vertices` may bridge is that of the "widest box" computed as column 4 minus column 3 (x_max - y_min)."""
    if layout_shape_mode not in ("rect", "poly", "quad", "auto"):
        raise ValueError("layout_shape_mode must be one of ['rect', 'poly', 'quad', 'auto']")
    n = len(boxes)
    page, cols, rows = _mask_windows(boxes, masks.shape[1:], scale_ratio)
    wh = page[:, 2:4] - page[:, 0:2]
    rects = np.stack([page[:, [0, 1]], page[:, [2, 1]], page[:, [2, 3]], page[:, [0, 3]]], axis=1).astype(np.float32)      # [n, 4, 2]
    shapes: list = [rects[i] for i in range(n)]
    if layout_shape_mode == "rect" or n == 0:
        return shapes
    widest = max(boxes[:, 4] - boxes[:, 3])
    traceable = [i for i in range(n) if wh[i, 0] > 0 and wh[i, 1] > 0 and rows[i, 1] > rows[i, 0] and cols[i, 1] > cols[i, 0]
                 and masks[i, rows[i, 0]:rows[i, 1], cols[i, 0]:cols[i, 1]].sum() != 0]
    for i in traceable:
        w, h = int(wh[i, 0]), int(wh[i, 1])
        window = masks[i, rows[i, 0]:rows[i, 1], cols[i, 0]:cols[i, 1]]
        polygon = mask_to_polygon(resize_nearest(window.astype(np.uint8), w, h), wh[i, 0] if wh[i, 0] > widest * 0.6 else widest)
        if polygon is not None:
            if len(polygon) < 4:
                continue                                            # a degenerate trace: the rectangle stays
            if len(polygon) > 0:
                polygon = polygon + page[i, 0:2]
        if layout_shape_mode == "poly":
            shapes[i] = polygon
            continue
        quad = polygon_to_quad(polygon)
        if layout_shape_mode == "quad":
            shapes[i] = quad if quad is not None else rects[i]
        else:
            shapes[i] = _auto_shape(polygon, quad, rects[i], shapes[i - 1] if i > 0 else None)
    return shapes


def polygon_keep_mask(crop_hw: Sequence[int], polygon_points, x_min: int, y_min: int) -> np.ndarray:
    """crop_img's polygon branch (model_utils.py:109-118): which pixels of the region crop - the page slice
    [y_min:y_max, x_min:x_max], `crop_hw` = its (height, width) - lie in the polygon; the caller whites out the others.
    The polygon's coordinates are truncated to integers first (np.array(..., dtype=np.int32))."""
    poly = np.array(polygon_points, dtype=np.int32)
    if poly.ndim == 1:
        poly = poly.reshape((-1, 2))
    poly = poly.reshape((-1, 2)) - np.array([x_min, y_min])
    mask = np.zeros((int(crop_hw[0]), int(crop_hw[1])), dtype=np.uint8)
    fill_poly(mask, poly, 1)
    return mask.astype(bool)
