"""Host side of the layout stage: what surrounds the PP-DocLayout network in the reference.

* `LayoutPostProcess` ....... PPPostProcess (pp_doclayout/post_process.py:20-243) - C++ behind the C-ABI; with the detector's
  masks and a layout_shape_mode other than "rect", the polygon stage of layout_polygon.py runs between its sort and unclip steps
* label / threshold / merge tables of the shipped models: data captured from
  rapid_doc/model/layout/rapid_layout_self/utils/typings.py:14-140 (tests/golden/layout_tables.json)
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np

_MODES = {"union": 0, "large": 1, "small": 2}


class _Cfg(C.Structure):
    _fields_ = [("n_classes", C.c_int32), ("image_index", C.c_int32), ("formula_index", C.c_int32),
                ("thresh_is_dict", C.c_int32), ("thresh", C.c_void_p), ("layout_nms", C.c_int32),
                ("merge_kind", C.c_int32), ("merge_per_class", C.c_void_p), ("unclip_kind", C.c_int32),
                ("unclip", C.c_void_p), ("unclip_present", C.c_void_p)]


class LayoutPostProcess:
    """Same constructor arguments and result dicts as the reference's PPPostProcess."""

    def __init__(self, labels: Sequence[str], conf_thres: Union[float, Dict[int, float]] = 0.5, iou_thres: float = 0.5,
                 layout_nms: bool = True, layout_merge_bboxes_mode: Union[None, str, Dict[int, str]] = None,
                 layout_unclip_ratio=None, scale_size=None):
        from . import _lib
        self._lib = _lib.load()
        self.labels = list(labels)
        self.scale_size = scale_size
        n = len(self.labels)
        cfg = _Cfg()
        cfg.n_classes = n
        cfg.image_index = self.labels.index("image") if "image" in self.labels else -1
        cfg.formula_index = self.labels.index("formula") if "formula" in self.labels else -1
        if isinstance(conf_thres, dict):
            self._thresh = np.full(n, 0.5, np.float32)
            for k, v in conf_thres.items():
                if 0 <= int(k) < n:
                    self._thresh[int(k)] = v
            cfg.thresh_is_dict = 1
        else:
            self._thresh = np.array([conf_thres], np.float32)
            cfg.thresh_is_dict = 0
        cfg.thresh = self._thresh.ctypes.data
        cfg.layout_nms = 1 if layout_nms else 0
        self._merge = np.zeros(max(n, 1), np.int8)
        if layout_merge_bboxes_mode is None:
            cfg.merge_kind = 0
        elif isinstance(layout_merge_bboxes_mode, str):
            cfg.merge_kind = _MODES[layout_merge_bboxes_mode]
        else:
            cfg.merge_kind = 3
            for k, v in layout_merge_bboxes_mode.items():
                if 0 <= int(k) < n:
                    self._merge[int(k)] = _MODES[v]
        cfg.merge_per_class = self._merge.ctypes.data
        self._unclip = np.ones((max(n, 1), 2), np.float32)
        self._present = np.zeros(max(n, 1), np.uint8)
        if not layout_unclip_ratio:
            cfg.unclip_kind = 0
        elif isinstance(layout_unclip_ratio, dict):
            cfg.unclip_kind = 2
            for k, (rw, rh) in layout_unclip_ratio.items():
                if 0 <= int(k) < n:
                    self._unclip[int(k)] = (rw, rh)
                    self._present[int(k)] = 1
        else:
            r = (layout_unclip_ratio, layout_unclip_ratio) if isinstance(layout_unclip_ratio, float) else tuple(layout_unclip_ratio)
            assert len(r) == 2, "The length of `layout_unclip_ratio` should be 2."
            cfg.unclip_kind = 1
            self._unclip[0] = r
        cfg.unclip = self._unclip.ctypes.data
        cfg.unclip_present = self._present.ctypes.data
        self._cfg = cfg

    def __call__(self, boxes: np.ndarray, img_size: Tuple[int, int], masks=None, layout_shape_mode: Optional[str] = "auto"):
        """boxes [n, 6|7|8] float32, img_size (width, height), masks [n, hm, wm] or None.  Returns the reference's list of dicts
        {cls_id, label, score, coordinate, order(, polygon_points)}; `polygon_points` only with masks and a mode other than "rect"
        (post_process.py:40-41,213-218), and a box whose polygon came out as None is dropped (:600-603)."""
        if layout_shape_mode == "rect":
            masks = None
        b = np.ascontiguousarray(boxes, dtype=np.float32)
        if b.size == 0:
            return np.array([])
        n, ncol = b.shape
        polygons = None
        if masks is not None:
            from . import layout_polygon
            if self.scale_size is None:
                raise ValueError("scale_size (the detector's input size) is needed to place boxes on the mask grid")
            sel = np.zeros((n, 6), np.float32)
            src = np.zeros(n, np.int32)
            m = C.c_int32(0)
            rc = self._lib.rd_layout_postprocess_select(b.ctypes.data, n, ncol, int(img_size[0]), int(img_size[1]), C.byref(self._cfg),
                                                        sel.ctypes.data, src.ctypes.data, C.byref(m))
            if rc != 0:
                raise ValueError(f"The shape of boxes should be 6, 7 or 8 columns, instead of {ncol}")
            if m.value == 0:
                return np.array([])
            scale_ratio = [h / s for h, s in zip(self.scale_size, img_size)]
            polygons = layout_polygon.polygons_from_masks(sel[:m.value], np.asarray(masks)[src[:m.value]], scale_ratio, layout_shape_mode)
        out = np.zeros((n, 6), np.float32)
        order = np.zeros(n, np.int32)
        k = C.c_int32(0)
        rc = self._lib.rd_layout_postprocess(b.ctypes.data, n, ncol, int(img_size[0]), int(img_size[1]), C.byref(self._cfg),
                                             out.ctypes.data, order.ctypes.data, C.byref(k))
        if rc != 0:
            raise ValueError(f"The shape of boxes should be 6, 7 or 8 columns, instead of {ncol}")
        if k.value == 0:
            return np.array([]) if not self._any_survivor_possible(b) else []
        res = []
        for i in range(k.value):
            r = out[i]
            d = {"cls_id": int(r[0]), "label": self.labels[int(r[0])], "score": float(r[1]),
                 "coordinate": [float(r[2]), float(r[3]), float(r[4]), float(r[5])], "order": int(order[i])}
            if polygons is not None:
                if polygons[order[i] - 1] is None:
                    continue
                d["polygon_points"] = polygons[order[i] - 1]
            res.append(d)
        return res

    @staticmethod
    def _any_survivor_possible(b) -> bool:
        return False


# ------------------------------------------------------------------------------------------------------------------
# Region bookkeeping after the layout network (rectangle mode), restated from
#   rapid_doc/backend/utils/utils.py:109-173        filter_overlap_boxes
#   rapid_doc/utils/model_utils.py:90-124,162-196   crop_img / get_res_list_from_layout_res
# ------------------------------------------------------------------------------------------------------------------
_EXCLUSIVE_LABELS = {"image", "seal", "chart"}          # never dropped against a box of another label
OCR_CATEGORY_IDS = (0, 1, 2, 4, 6, 7, 9)                 # model_utils.py:177
TABLE_CATEGORY_ID = 5
CHECKBOX_CATEGORY_ID = 200                          # utils/enum_class.py:106 CategoryId.CheckBox
FORMULA_CATEGORY_IDS = (8, 13, 14)
IMAGE_CATEGORY_IDS = (3,)


def _rect(det) -> Tuple[float, float, float, float]:
    p = det["poly"]
    return p[0], p[1], p[4], p[5]


def _overlap_over_smaller(a, b) -> float:
    iw = max(0, min(a[2], b[2]) - max(a[0], b[0]))
    ih = max(0, min(a[3], b[3]) - max(a[1], b[1]))
    inter = float(iw) * float(ih)
    ref = min(abs((a[2] - a[0]) * (a[3] - a[1])), abs((b[2] - b[0]) * (b[3] - b[1])))
    return 0.0 if ref == 0 else inter / ref


def filter_overlap_boxes(layout_dets: Sequence[dict], use_custom_ocr: bool = False) -> List[dict]:
    """Drop 'reference' boxes, boxes thinner than 6 px, and - for pairs overlapping by > 0.7 of the smaller box - the
    smaller one (image / seal / chart are only compared with their own label; inline formulas are only touched in
    custom-OCR mode, where an overlap > 0.5 removes the formula).  With `polygon_points` on the first box of a pair, the pair is
    left alone when the POLYGONS overlap by less than 0.7 of the smaller one (utils.py:150-155)."""
    boxes = [dict(d) for d in layout_dets if d["original_label"] != "reference"]
    dropped = set()
    for i in range(len(boxes)):
        ri = _rect(boxes[i])
        if ri[2] - ri[0] < 6 or ri[3] - ri[1] < 6:
            dropped.add(i)
        li = boxes[i]["original_label"]
        for j in range(i + 1, len(boxes)):
            if i in dropped or j in dropped:
                continue
            rj = _rect(boxes[j])
            lj = boxes[j]["original_label"]
            ov = _overlap_over_smaller(ri, rj)
            if li == "inline_formula" or lj == "inline_formula":
                if not use_custom_ocr:
                    continue
                if ov > 0.5:
                    if li == "inline_formula":
                        dropped.add(i)
                    if lj == "inline_formula":
                        dropped.add(j)
                    continue
            if ov > 0.7:
                if boxes[i].get("polygon_points"):
                    from .layout_polygon import polygon_overlap_ratio
                    if polygon_overlap_ratio(boxes[i]["polygon_points"], boxes[j]["polygon_points"], "small") < 0.7:
                        continue
                if ({li, lj} & _EXCLUSIVE_LABELS) and li != lj:
                    continue
                ai = abs((ri[2] - ri[0]) * (ri[3] - ri[1]))
                aj = abs((rj[2] - rj[0]) * (rj[3] - rj[1]))
                dropped.add(j if ai >= aj else i)
    return [b for k, b in enumerate(boxes) if k not in dropped]


def split_regions(layout_dets: Sequence[dict]) -> Tuple[List[dict], List[dict], List[dict]]:
    """(ocr regions, table regions, formula regions with 'bbox') - get_res_list_from_layout_res (utils/model_utils.py:
    162-180; the image-in-table bookkeeping of :181-194 is `images_inside_tables`).  Like the reference it writes the integer 'bbox' INTO the formula
    detections (they are the caller's dicts: the field is part of the page's output)."""
    ocr, tables, formulas = [], [], []
    for d in layout_dets:
        cid = int(d["category_id"])
        if cid in FORMULA_CATEGORY_IDS:
            p = d["poly"]
            d["bbox"] = [int(p[0]), int(p[1]), int(p[4]), int(p[5])]
            formulas.append(d)
        elif cid in OCR_CATEGORY_IDS:
            ocr.append(d)
        elif cid == TABLE_CATEGORY_ID:
            tables.append(d)
    return ocr, tables, formulas


def images_inside_tables(layout_dets: Sequence[dict], overlap_threshold: float = 0.8) -> List[Tuple[dict, dict]]:
    """(image detection, table detection) pairs of get_res_list_from_layout_res' second loop (utils/model_utils.py:181-194): an image
    region (category 3) belongs to every table at least `overlap_threshold` of whose own integer-box area lies inside the table's
    integer box.  Image-major order, like the reference's loops."""
    def rect_area(d):
        p = d["poly"]
        x0, y0, x1, y1 = int(p[0]), int(p[1]), int(p[4]), int(p[5])
        return x0, y0, x1, y1, (x1 - x0) * (y1 - y0)
    images = [d for d in layout_dets if int(d["category_id"]) in IMAGE_CATEGORY_IDS]
    tables = [d for d in layout_dets if int(d["category_id"]) == TABLE_CATEGORY_ID]
    pairs = []
    for im in images:
        a = rect_area(im)
        for tb in tables:
            b = rect_area(tb)
            ix0, iy0, ix1, iy1 = max(a[0], b[0]), max(a[1], b[1]), min(a[2], b[2]), min(a[3], b[3])
            if ix1 <= ix0 or iy1 <= iy0:
                continue
            if (ix1 - ix0) * (iy1 - iy0) >= overlap_threshold * a[4]:
                pairs.append((im, tb))
    return pairs


def table_fill_images(table_det: dict, useful_list: Sequence[int]) -> List[dict]:
    """extract_table_fill_image's layout branch (utils/span_pre_proc.py:245-259; the PDF-image branch needs the PDF): the images
    region collection attached to this table (`layout_image_list`) get their page box (`ori_bbox`, `bbox`) and their corners in
    table-crop coordinates (`ocr_bbox`); the SAME dicts are returned, like the reference."""
    paste_x, paste_y, xmin, ymin = useful_list[:4]
    out = []
    for image in table_det.get("layout_image_list", []):
        p = image["poly"]
        bbox = [p[0], p[1], p[4], p[5]]
        image["ori_bbox"] = bbox
        x0, y0, x1, y1 = bbox[0] + paste_x - xmin, bbox[1] + paste_y - ymin, bbox[2] + paste_x - xmin, bbox[3] + paste_y - ymin
        image["bbox"] = bbox
        image["ocr_bbox"] = [[x0, y0], [x1, y0], [x1, y1], [x0, y1]]
        out.append(image)
    return out


def crop_geometry(det: dict, paste_x: int = 0, paste_y: int = 0) -> List[int]:
    """crop_img's `useful_list`: [paste_x, paste_y, xmin, ymin, xmax, ymax, new_w, new_h] (white margin of paste_x/y)."""
    p = det["poly"]
    x0, y0, x1, y1 = int(p[0]), int(p[1]), int(p[4]), int(p[5])
    return [paste_x, paste_y, x0, y0, x1, y1, x1 - x0 + 2 * paste_x, y1 - y0 + 2 * paste_y]


def _int_rect(det: dict):
    """Rounded bounding rectangle of a detection's poly (backend/utils/utils.py:175-182), None without a poly."""
    poly = det.get("poly")
    if not poly or len(poly) < 6:
        return None
    xs = [int(round(float(v))) for v in poly[0::2]]
    ys = [int(round(float(v))) for v in poly[1::2]]
    return min(xs), min(ys), max(xs), max(ys)


def expand_formula_crop(formula: dict, layout_dets: Sequence[dict], image_hw: Tuple[int, int], expand_px: int = 2) -> dict:
    """`_expand_formula_crop_res` (backend/utils/utils.py:189-243): the crop of a formula grows by `expand_px` on every side,
    clipped to the page and stopped at any other layout box that lies beside / above / below it and overlaps the grown
    extent in the other axis.  Returns the detection itself when nothing changes is possible, else a copy with the new
    `poly` / `bbox` and without `polygon_points`."""
    rect = _int_rect(formula) if expand_px > 0 else None
    if rect is None:
        return formula
    H, W = image_hw[:2]
    x0, y0, x1, y1 = rect
    ex0, ey0, ex1, ey1 = max(0, x0 - expand_px), max(0, y0 - expand_px), min(W, x1 + expand_px), min(H, y1 + expand_px)

    def overlap(a0, a1, b0, b1):
        return max(a0, b0) < min(a1, b1)

    for other in layout_dets:
        if other is formula:
            continue
        r = _int_rect(other)
        if r is None:
            continue
        ox0, oy0, ox1, oy1 = r
        if ox1 <= x0 and overlap(ey0, ey1, oy0, oy1):
            ex0 = max(ex0, ox1)
        if ox0 >= x1 and overlap(ey0, ey1, oy0, oy1):
            ex1 = min(ex1, ox0)
        if oy1 <= y0 and overlap(ex0, ex1, ox0, ox1):
            ey0 = max(ey0, oy1)
        if oy0 >= y1 and overlap(ex0, ex1, ox0, ox1):
            ey1 = min(ey1, oy0)
    if ex0 >= ex1 or ey0 >= ey1:
        return formula
    out = dict(formula)
    out["poly"] = [ex0, ey0, ex1, ey0, ex1, ey1, ex0, ey1]
    out["bbox"] = [ex0, ey0, ex1, ey1]
    out.pop("polygon_points", None)
    return out


# ------------------------------------------------------------------------------------------------------------------
# label -> CategoryId mapping and the per-box dict schema of RapidLayoutModel.batch_predict
# (rapid_doc/model/layout/rapid_layout.py:55-108,131-227).  The tables are data captured from the reference
# (rapiddoc_amd/data/layout_category_maps.json, regenerated by tests/golden/make_golden.py).
# ------------------------------------------------------------------------------------------------------------------
import json as _json
from pathlib import Path as _Path

_TABLES = _json.loads((_Path(__file__).resolve().parent / "data" / "layout_category_maps.json").read_text())
CATEGORY_ID: Dict[str, int] = _TABLES["category_id"]


def category_map(model_family: str, markdown_ignore_labels: Sequence[str] = ()) -> Dict[str, int]:
    """model_family: 'pp_doclayout' (S/M/L), 'pp_doclayout_plus' (plus-L) or 'pp_doclayoutv2' (V2/V3).  Labels in
    `markdown_ignore_labels` map to Abandon (rapid_layout.py:162-165)."""
    base = _TABLES["label_to_category"][model_family]
    return {k: (CATEGORY_ID["Abandon"] if k in markdown_ignore_labels else v) for k, v in base.items()}


def to_layout_dets(post: Sequence[dict], model_family: str, ordered: bool, markdown_ignore_labels: Sequence[str] = ()) -> List[dict]:
    """LayoutPostProcess output -> the `layout_dets` dicts BatchAnalyze consumes: category_id, original_label,
    original_order, poly [x0,y0,x1,y0,x1,y1,x0,y1], polygon_points, score rounded to 3 decimals.  polygon_points: [[x, y], ...]
    floats when EVERY box of the page has a polygon, else None for all of them (pp_doclayout/main.py:69-72, rapid_layout.py:93-98)."""
    cmap = category_map(model_family, markdown_ignore_labels)
    polys = [d.get("polygon_points") for d in post]
    if any(p is None for p in polys):
        polys = [None] * len(polys)
    out = []
    for i, d in enumerate(post):
        x0, y0, x1, y1 = d["coordinate"]
        pts = None if polys[i] is None else [[float(x), float(y)] for x, y in polys[i]]
        out.append({"category_id": cmap[d["label"]], "original_label": d["label"], "original_order": i if ordered else -1,
                    "poly": [x0, y0, x1, y0, x1, y1, x0, y1], "polygon_points": pts, "score": round(float(d["score"]), 3)})
    return out
