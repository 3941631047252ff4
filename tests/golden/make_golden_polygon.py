#!/usr/bin/env python3
"""Golden vectors of the layout post-process's POLYGON branch, minted by the reference's own code (build container only).

    python tests/golden/make_golden_polygon.py            # writes tests/golden/layout_polygon.json

What runs is rapid_doc/model/layout/rapid_layout_self/model_handler/pp_doclayout/post_process.py, loaded by file path, unmodified:
`PPPostProcess.__call__` with masks (threshold / NMS / filters / merge / sort carrying the masks along, :20-243),
`extract_polygon_points_by_masks` (:425-535), `mask2polygon` / `extract_custom_vertices` (:261-423), `convert_polygon_to_quad`
(:536-563), `calculate_polygon_overlap_ratio` (:681-719), `restructured_boxes` (:566-608).
The module imports `cv2` and `shapely`, which are not installed here.  They are stood in for by thin modules whose functions call
this repo's C primitives (rapiddoc_amd/csrc/polygon_ops.cpp via rapiddoc_amd.layout_polygon) - so what the vectors pin is every
line the reference wrote around those calls, and NOT the arithmetic inside OpenCV / GEOS (that stays unpinned, see polygon_ops.cpp).
Inputs come from tests/golden/polygon_masks.py (seeded); the JSON holds the case parameters and the reference's outputs."""
import importlib.util
import json
import sys
import types
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
REF = Path("/root/reference")
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE.parents[1]))
import polygon_masks as PM  # noqa: E402
from rapiddoc_amd import layout_polygon as LP  # noqa: E402


def fake_cv2():
    cv2 = types.ModuleType("cv2")
    cv2.RETR_EXTERNAL, cv2.CHAIN_APPROX_SIMPLE, cv2.INTER_NEAREST = 0, 2, 0
    cv2.findContours = lambda img, mode, method: (tuple(c.reshape(-1, 1, 2) for c in LP.find_external_contours(img)), None)
    cv2.contourArea = lambda c: LP.contour_area(c)
    cv2.arcLength = lambda c, closed: LP.arc_length(c, closed)
    cv2.approxPolyDP = lambda c, eps, closed: LP.approx_poly_dp(c, eps, closed).reshape(-1, 1, 2)
    cv2.resize = lambda img, size, interpolation=None: LP.resize_nearest(img, int(size[0]), int(size[1]))
    cv2.minAreaRect = lambda pts: ("rect-of", np.array(pts, dtype=np.float32))
    cv2.boxPoints = lambda r: LP.min_area_rect_points(r[1])
    return cv2


def fake_shapely():
    class _Area:
        def __init__(self, a):
            self.area = a

    class Polygon:
        is_valid = True

        def __init__(self, pts):
            self.pts = [list(map(float, p)) for p in pts]
            self.area = LP.polygon_area(self.pts)

        def intersection(self, other):
            return _Area(LP.polygon_intersection_area(self.pts, other.pts))

        def union(self, other):
            return _Area(self.area + other.area - LP.polygon_intersection_area(self.pts, other.pts))

    shapely, geometry = types.ModuleType("shapely"), types.ModuleType("shapely.geometry")
    geometry.Polygon = Polygon
    shapely.geometry = geometry
    return shapely, geometry


def to_jsonable(poly):
    return None if poly is None else [[float(x), float(y)] for x, y in poly]


def main():
    sys.modules["cv2"] = fake_cv2()
    sys.modules["shapely"], sys.modules["shapely.geometry"] = fake_shapely()
    spec = importlib.util.spec_from_file_location(
        "ref_layout_post_poly", REF / "rapid_doc/model/layout/rapid_layout_self/model_handler/pp_doclayout/post_process.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)

    vertices = []
    for seed in range(40):
        poly = PM.random_polygon(seed)
        for dist in (60.0, 400.0):
            vertices.append({"seed": seed, "max_allowed_dist": dist, "n_in": len(poly),
                             "out": to_jsonable(ref.extract_custom_vertices(poly, dist))})
    print("extract_custom_vertices:", len(vertices), "cases; output sizes", sorted({len(v["out"]) for v in vertices}))

    tables = json.loads((HERE / "layout_tables.json").read_text())
    v2_merge = {int(k): v for k, v in tables["PP_DOCLAYOUTV2_layout_merge_bboxes_mode"].items()}
    post = []
    for ci, (mode, ncol, thr, merge, unclip) in enumerate((
            ("auto", 7, 0.3, v2_merge, [1.0, 1.0]), ("poly", 6, 0.3, None, None), ("quad", 6, 0.4, "large", [1.0, 1.0]),
            ("rect", 7, 0.3, v2_merge, [1.0, 1.0]), ("auto", 8, {3: 0.2, 7: 0.6}, None, 1.05), ("auto", 6, 0.3, None, None))):
        labels = [f"c{i}" for i in range(25)]
        labels[14], labels[5] = "image", "formula"
        boxes, masks, kinds = PM.make_case(7000 + ci, 48, ncol, 25)
        pp = ref.PPPostProcess(labels, thr, 0.5, layout_merge_bboxes_mode=merge, layout_unclip_ratio=unclip, scale_size=(PM.INPUT, PM.INPUT))
        out = pp(boxes.copy(), [PM.PAGE_W, PM.PAGE_H], masks.copy(), mode)
        rows = [] if isinstance(out, np.ndarray) else [
            {"cls_id": d["cls_id"], "label": d["label"], "score": d["score"], "coordinate": d["coordinate"], "order": d["order"],
             "polygon_points": to_jsonable(d.get("polygon_points")), "has_polygon": "polygon_points" in d} for d in out]
        post.append({"seed": 7000 + ci, "n": 48, "ncol": ncol, "mode": mode, "labels": labels,
                     "thresh": thr if not isinstance(thr, dict) else {str(k): v for k, v in thr.items()},
                     "merge": merge if not isinstance(merge, dict) else {str(k): v for k, v in merge.items()}, "unclip": unclip, "out": rows})
        sizes = [len(r["polygon_points"]) for r in rows if r["polygon_points"] is not None]
        print(f"post case {ci} ({mode}, {ncol} cols): {len(rows)} boxes kept, polygon sizes {sorted(set(sizes))}")
    (HERE / "layout_polygon.json").write_text(json.dumps({"custom_vertices": vertices, "post": post}))


if __name__ == "__main__":
    main()
