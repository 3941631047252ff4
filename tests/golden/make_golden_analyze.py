#!/usr/bin/env python3
"""Traces of the REFERENCE's own page-batch driver, run with recording stand-ins for the models (build container only).

    python tests/golden/make_golden_analyze.py        # writes tests/golden/analyze_trace_seed*.json

What runs is the reference's code, unmodified, imported from /root/reference:
    rapid_doc/backend/pipeline/batch_analyze.py:78-164      BatchAnalyze.__call__ (stage sequencing)
    rapid_doc/backend/pipeline/analyze_utils.py:105-292     _run_ocr_det_batch, _run_ocr_rec_postprocess
    rapid_doc/utils/ocr_utils.py:105-431,494-536            sorted_boxes, merge_det_boxes, update_det_boxes, get_ocr_result_list,
                                                            get_rotate_crop_image (crop SIZE arithmetic)
    rapid_doc/utils/model_utils.py:90-196                   crop_img, get_res_list_from_layout_res
    rapid_doc/backend/utils/utils.py                        filter_overlap_boxes, _expand_formula_crop_res
What is stood in for (none of it is arithmetic of the path under test):
  * the three models: a layout model that returns prepared detections, an OCR model whose `det_batch_predict` returns seeded
    boxes and whose `ocr(det=False)` returns a text / score that is a pure function of the crop's SHAPE, a formula model that
    names the crop's shape - every call is recorded (order, batch sizes, shapes, crc32 of the det canvases);
  * `cv2`, absent offline: `cvtColor` flips the channel order, `warpPerspective` returns a blank image of the REQUESTED SIZE
    (the fixture pins crop sizes and call order, not crop pixels - those are pinned to oracle/cv2_ops.py elsewhere), everything
    else is a mock that is never reached on this path;  `loguru`, `tqdm` and the other missing wheels are inert mocks;
  * the package `__init__` files of rapid_doc (they import every model wrapper and their wheels): modules are loaded file by file.
The committed JSON holds inputs (page seeds, layout detections, the boxes the stand-in detector returned), the recorded calls
and the reference's output dicts: data only.
"""
import importlib
import importlib.abc
import importlib.machinery
import importlib.util
import json
import os
import sys
import types
import zlib
from pathlib import Path
from unittest import mock

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parents[1]
REF = Path("/root/reference")
sys.path.insert(0, str(ROOT))


# ---------------------------------------------------------------------------------------------------------------------
# import machinery: rapid_doc.* file by file (no package __init__), missing third-party wheels as mocks
# ---------------------------------------------------------------------------------------------------------------------
class _MockLoader(importlib.abc.Loader):
    def create_module(self, spec):
        m = mock.MagicMock(name=spec.name)
        m.__name__, m.__path__, m.__spec__, m.__loader__ = spec.name, [], spec, self
        return m

    def exec_module(self, module):
        pass


class _RefFinder(importlib.abc.MetaPathFinder):
    # package __init__ files run normally, except the ones that pull in every model wrapper (and their wheels)
    SKIP_INIT = {"rapid_doc", "rapid_doc.backend.pipeline"}

    def __init__(self):
        self.mocked = set()

    def find_spec(self, name, path, target=None):
        top = name.split(".")[0]
        if top == "rapid_doc":
            rel = REF.joinpath(*name.split("."))
            if rel.is_dir():
                if name not in self.SKIP_INIT and (rel / "__init__.py").exists():
                    return importlib.util.spec_from_file_location(name, rel / "__init__.py", submodule_search_locations=[str(rel)])
                spec = importlib.machinery.ModuleSpec(name, None, is_package=True)      # namespace-like package, __init__ not run
                spec.submodule_search_locations = [str(rel)]
                return spec
            if rel.with_suffix(".py").exists():
                return importlib.util.spec_from_file_location(name, rel.with_suffix(".py"))
            return None
        if top in self.mocked:
            return importlib.machinery.ModuleSpec(name, _MockLoader(), is_package=True)
        return None


def _fake_cv2():
    cv2 = types.ModuleType("cv2")
    cv2.COLOR_RGB2BGR, cv2.COLOR_BGR2RGB, cv2.BORDER_REPLICATE, cv2.INTER_CUBIC = 4, 4, 1, 2
    cv2.cvtColor = lambda img, code: np.ascontiguousarray(np.asarray(img)[:, :, ::-1])
    cv2.getPerspectiveTransform = lambda src, dst: np.eye(3)
    cv2.warpPerspective = lambda img, M, size, borderMode=None, flags=None: np.zeros((int(size[1]), int(size[0]), 3), np.uint8)

    def fill_poly(img, pts, color):       # crop_img's polygon mask (model_utils.py:114): this repo's primitive under OpenCV's name,
        from rapiddoc_amd import layout_polygon      # as in make_golden_polygon.py - pins the code AROUND the call, not OpenCV's rasteriser
        for p_ in pts:
            layout_polygon.fill_poly(img, np.asarray(p_).reshape(-1, 2), int(color))
        return img
    cv2.fillPoly = fill_poly
    # get_rotate_image (utils/boxbase.py:312-326): the two quarter turns, as numpy does them (seeds 8 / 9)
    cv2.ROTATE_90_CLOCKWISE, cv2.ROTATE_90_COUNTERCLOCKWISE = 0, 2
    cv2.rotate = lambda img, code: np.ascontiguousarray(np.rot90(img, -1 if code == 0 else 1))

    def _missing(name):
        raise AttributeError(f"cv2.{name} is not stood in for: the traced path must not reach it")
    cv2.__getattr__ = _missing
    return cv2


def import_reference():
    finder = _RefFinder()
    sys.meta_path.insert(0, finder)
    sys.modules["cv2"] = _fake_cv2()
    if "shapely" not in sys.modules:                  # filter_overlap_boxes' polygon rule (utils.py:150-155): see make_golden_polygon.py
        sys.path.insert(0, str(HERE))
        import make_golden_polygon
        sys.modules["shapely"], sys.modules["shapely.geometry"] = make_golden_polygon.fake_shapely()
    # the model registry imports every model wrapper: replaced by the recording registry below
    reg = types.ModuleType("rapid_doc.backend.pipeline.model_init")
    reg.AtomModelSingleton = type("AtomModelSingleton", (), {})      # rebound per trace in main()
    sys.modules["rapid_doc.backend.pipeline.model_init"] = reg
    for _ in range(40):                               # wheels that are absent offline become inert mocks, one by one
        try:
            ba = importlib.import_module("rapid_doc.backend.pipeline.batch_analyze")
            return ba, reg
        except ModuleNotFoundError as e:
            top = (e.name or "").split(".")[0]
            if not top or top == "rapid_doc" or top in finder.mocked:
                raise
            finder.mocked.add(top)
            for k in [k for k in sys.modules if k.startswith("rapid_doc.") and k != "rapid_doc.backend.pipeline.model_init"]:
                del sys.modules[k]
    raise RuntimeError("could not import the reference driver")


# ---------------------------------------------------------------------------------------------------------------------
# recording stand-ins
# ---------------------------------------------------------------------------------------------------------------------
def rec_text_and_score(h, w, k):
    """What the stand-in recogniser 'reads' from a crop: a pure function of its shape and its position in the call."""
    score = ((h * 131 + w * 17 + k * 29) % 1000) / 1000.0
    return f"T{k}:{h}x{w}", score


def det_boxes_for(h, w, key):
    """Boxes the stand-in detector 'finds' in an h x w canvas: rows of text-line quads, some split into neighbours on one line
    (merge_det_boxes), some tilted (calculate_is_angle), one sliver narrower than 3 px (min_width filter)."""
    rng = np.random.default_rng(key)
    boxes = []
    y = 56
    while y + 30 < h - 50:
        bh = int(rng.integers(16, 30))
        x = 52 + int(rng.integers(0, 30))
        n_seg = int(rng.integers(1, 4))
        for s in range(n_seg):
            if x + 40 >= w - 52:
                break
            bw = int(rng.integers(30, max(31, (w - 104) // n_seg)))
            bw = min(bw, w - 52 - x)
            dy = int(rng.integers(-2, 3))
            q = np.array([[x, y + dy], [x + bw, y + dy], [x + bw, y + dy + bh], [x, y + dy + bh]], dtype=np.float32)
            r = rng.random()
            if r < 0.12:
                q[1, 1] += bh * 1.4; q[2, 1] += bh * 1.4          # tilted
            elif r < 0.18:
                q[1, 0] = q[2, 0] = x + 2                           # a sliver (width 2)
            boxes.append(q)
            x += bw + int(rng.integers(-6, 25))                     # overlapping / touching / separate neighbours
        y += bh + int(rng.integers(6, 30))
    rng.shuffle(boxes)
    return np.asarray(boxes, dtype=np.float32).reshape(-1, 4, 2)


class RecordingOcr:
    def __init__(self, trace, lang=None):
        self.trace, self.lang = trace, lang       # `lang`: only set (and recorded) in the multi-language trace

    def det_batch_predict(self, img_list, max_batch_size=8):
        imgs = [np.asarray(im) for im in img_list]
        call = {"batch_size": int(max_batch_size), "shapes": [list(im.shape) for im in imgs],
                "crc32": [zlib.crc32(np.ascontiguousarray(im).tobytes()) for im in imgs], "boxes": []}
        if self.lang:
            call["lang"] = self.lang
        out = []
        for im in imgs:
            b = det_boxes_for(im.shape[0], im.shape[1], zlib.crc32(np.ascontiguousarray(im).tobytes()) & 0xffff)
            call["boxes"].append(b.tolist())
            out.append((b, 0.0))
        self.trace["det_calls"].append(call)
        return out

    def ocr(self, img, det=True, rec=True, tqdm_enable=False, **kw):
        assert det is False
        crops = [np.asarray(c) for c in img]
        self.trace["rec_calls"].append(dict({"shapes": [list(c.shape[:2]) for c in crops]}, **({"lang": self.lang} if self.lang else {})))
        return [[rec_text_and_score(c.shape[0], c.shape[1], k) for k, c in enumerate(crops)]]


class RecordingFormula:
    def __init__(self, trace):
        self.trace = trace

    def batch_predict(self, images, batch_size=1, **kw):
        self.trace["formula_calls"].append({"batch_size": int(batch_size), "shapes": [list(np.asarray(i).shape[:2]) for i in images],
                                            "crc32": [zlib.crc32(np.ascontiguousarray(np.asarray(i)).tobytes()) for i in images]})
        return [f"\\\\frac{{{np.asarray(i).shape[0]}}}{{{np.asarray(i).shape[1]}}}" for i in images]


class RecordingLayout:
    def __init__(self, trace, dets):
        self.trace, self.dets = trace, dets

    def batch_predict(self, images, batch_size):
        self.trace["layout_calls"].append({"batch_size": int(batch_size), "shapes": [list(np.asarray(i).shape) for i in images]})
        return json.loads(json.dumps(self.dets))


# ---------------------------------------------------------------------------------------------------------------------
def layout_for_page(rng, H, W, polygons=False):
    """Prepared layout detections of one page (the dict schema of RapidLayoutModel.batch_predict, rapid_layout.py:58-107).
    `polygons`: every box also carries `polygon_points` (what a detector with a mask head yields in the default "auto" shape mode):
    the box with its corners cut, a slanted quadrilateral for formulas, and - for the two nearly coincident text boxes - the left and
    the right half, so that the polygon rule of filter_overlap_boxes keeps both."""
    dets = []

    def add(cat, label, x0, y0, x1, y1, score=0.9, order=None, shape="cut"):
        pts = None
        if polygons:
            w_, h_ = x1 - x0, y1 - y0
            if shape == "cut":
                c = min(w_, h_) * 0.25
                pts = [[x0 + c, y0], [x1 - c, y0], [x1, y0 + c], [x1, y1 - c], [x1 - c, y1], [x0 + c, y1], [x0, y1 - c], [x0, y0 + c]]
            elif shape == "slant":
                pts = [[x0 + 0.1 * w_, y0], [x1, y0], [x1 - 0.1 * w_, y1], [x0, y1]]
            elif shape == "left":
                pts = [[x0, y0], [x0 + 0.55 * w_, y0], [x0 + 0.55 * w_, y1], [x0, y1]]
            elif shape == "right":
                pts = [[x0 + 0.45 * w_, y0], [x1, y0], [x1, y1], [x0 + 0.45 * w_, y1]]
            pts = [[float(a), float(b)] for a, b in pts]
        dets.append({"category_id": cat, "original_label": label, "original_order": len(dets) if order is None else order,
                     "poly": [x0, y0, x1, y0, x1, y1, x0, y1], "polygon_points": pts, "score": round(float(score), 3)})
    y = 70.5
    add(0, "doc_title", 180.2, y, 1010.7, y + 58.4)                       # Title
    y += 90
    for k in range(int(rng.integers(2, 4))):                                # text paragraphs of different sizes / buckets
        hgt = float(rng.integers(150, 420))
        x0 = float(rng.choice([88.0, 96.4, 120.9]))
        x1 = float(rng.choice([1100.2, 1050.0, 600.5]))
        add(1, "text", x0, y, x1, y + hgt)
        if k == 0:      # an inline formula inside the first paragraph (whited out of the det canvas, cut out of its lines)
            add(13, "inline_formula", x0 + 200.3, y + 60.2, x0 + 330.8, y + 92.6, 0.8, shape="slant")
        y += hgt + 24.6
    add(14, "display_formula", 300.0, y, 900.0, y + 70.0, 0.85, shape="slant")   # an isolated formula: formula model only
    y += 95
    add(1, "text", 90.0, y, 560.0, y + 130.0, shape="left")
    add(1, "text", 96.0, y + 4.0, 552.0, y + 122.0, 0.55, shape="right")   # almost the same box, lower score: filter_overlap_boxes
    add(3, "image", 620.0, y, 1100.0, y + 260.0, 0.9)                      # a figure: no OCR
    add(2, "abandon", 500.0, H - 60.0, 700.0, H - 25.0, 0.7)               # footer: OCR region (category 2)
    return dets


def main():
    ba, reg = import_reference()
    from rapid_doc.backend.pipeline.model_list import AtomicModel
    from rapiddoc_amd.pages import synth_page

    # seeds 0-4 were cut with ocr_config["seal_enable"] = False; 5 and 6 run the reference's DEFAULT (the key absent -> True,
    # batch_analyze.py:62): 5 = the pages and layout of seed 0 (no seal region: the seal stage must leave the output alone), 6 = two seal
    # regions (one with a polygon) -> _run_seal_ocr (batch_analyze.py:415-470) with a recording seal OCR model
    # 7: checkbox detection on (checkbox_config["checkbox_enable"], batch_analyze.py:51,207-219) with a stand-in checkbox_predict - its
    #    hits become CheckBox detections and mask the OCR detector's input next to the formulas;
    # 8 / 9: USE_DOC_ORIENTATION_CLASSIFY=true (batch_analyze.py:66-67,113-125,153-161) with a recording ImgOrientationCls stand-in:
    #    8 = both pages sideways ("90", "270": turned upright before the layout model, polys mapped back), 9 = "0" and "180" (left alone)
    for seed, (n_pages, formula_enable, formula_level, polygons, langs, seal) in enumerate([
            (3, True, 0, False, None, None), (2, False, 0, False, None, None), (2, True, 1, False, None, None), (2, True, 0, True, None, None),
            (3, True, 0, False, ["ch", "en", "ch"], None),              # pages of two languages in one batch
            (3, True, 0, False, None, "default"), (2, True, 0, False, None, "regions"),
            (2, True, 0, False, None, "default"), (2, True, 0, False, None, "default"), (2, True, 0, False, None, "default")]):
        rng = np.random.default_rng(7000 + (0 if seed == 5 else seed))
        trace = {"det_calls": [], "rec_calls": [], "formula_calls": [], "layout_calls": []}
        page_ids = [int(rng.integers(0, 1000)) for _ in range(n_pages)]
        pages = [synth_page(i)[0] for i in page_ids]
        H, W = pages[0].shape[:2]
        rotate_labels = {8: ["90", "270"], 9: ["0", "180"]}.get(seed)
        sideways = seed == 8
        dets = [layout_for_page(rng, H, W, polygons) for _ in range(n_pages)]
        if sideways:                          # the pages come in lying on their side (one quarter turn); either label stands them up as H x W
            pages = [np.ascontiguousarray(np.rot90(p_, 1)) for p_ in pages]
        os.environ["USE_DOC_ORIENTATION_CLASSIFY"] = "true" if rotate_labels else "false"

        class OrientationCls:                # get_atom_model(ImgOrientationCls).predict(rgb page) -> "0" | "90" | "180" | "270"
            def predict(self, img):
                img = np.asarray(img)
                k = len(trace.setdefault("orientation_calls", []))
                trace["orientation_calls"].append({"shape": list(img.shape), "crc32": zlib.crc32(np.ascontiguousarray(img).tobytes())})
                return rotate_labels[k]

        def checkbox_predict(bgr):           # utils/checkbox_det_cls.py checkbox_predict(bgr page) -> [{'bbox', 'text', ...}]
            bgr = np.asarray(bgr)
            k = len(trace.setdefault("checkbox_calls", []))
            trace["checkbox_calls"].append({"shape": list(bgr.shape), "crc32": zlib.crc32(np.ascontiguousarray(bgr).tobytes())})
            # one hit inside the first text region of the page (it masks that region's det input), one far outside every region
            first = next(d for d in dets[k] if d["category_id"] in (0, 1, 2, 4, 6, 7, 9))
            x0, y0 = int(first["poly"][0]) + 12, int(first["poly"][1]) + 6
            return [{"bbox": [x0, y0, x0 + 14, y0 + 14], "text": "checked" if k == 0 else "unchecked", "score": 0.97},
                    {"bbox": [3, 3, 15, 15], "text": "unchecked", "score": 0.5}]
        ba.checkbox_predict = checkbox_predict
        if seal == "regions":
            def seal_det(x0, y0, x1, y1, pts, order):
                return {"category_id": 3, "original_label": "seal", "original_order": order, "poly": [x0, y0, x1, y0, x1, y1, x0, y1],
                        "polygon_points": pts, "score": 0.88}
            dets[0].append(seal_det(820.4, 1300.2, 1040.9, 1510.7, None, len(dets[0])))
            dets[1].append(seal_det(130.0, 1330.0, 330.0, 1520.0, [[230.0, 1330.0], [330.0, 1425.0], [230.0, 1520.0], [130.0, 1425.0]], len(dets[1])))
            dets[1].append(seal_det(900.3, 60.1, 1100.2, 180.8, None, len(dets[1])))
        ocr = RecordingOcr(trace)
        ocr_by_lang = {lg: RecordingOcr(trace, lg) for lg in dict.fromkeys(langs or [])}

        class SealOcr:                       # `get_atom_model(OCR, is_seal=True)`: ocr(bgr crop, det=True, rec=True) -> [[box, (text, score)], ...]
            def ocr(self, img, det=True, rec=True, **kw):
                assert det is True and rec is True and not kw
                img = np.asarray(img)
                k = len(trace.setdefault("seal_calls", []))
                trace["seal_calls"].append({"shape": list(img.shape), "crc32": zlib.crc32(np.ascontiguousarray(img).tobytes())})
                if k == 2:
                    return [None]            # nothing read: the region keeps no `text`
                box = [[1.0, 2.0], [30.0, 2.0], [30.0, 12.0], [1.0, 12.0]]
                return [[[box, (f"seal {k} line a {img.shape[0]}x{img.shape[1]}", 0.91)], None, [box], [box, ()], [box, ("", 0.4)],
                         [box, (f"line b", 0.2)]]]      # malformed / empty items are skipped one by one (batch_analyze.py:456-467)

        class Registry:                      # rapid_doc/backend/pipeline/model_init.py:57-88 AtomModelSingleton
            def get_atom_model(self, atom_model_name, **kw):
                if atom_model_name == AtomicModel.ImgOrientationCls:
                    assert rotate_labels and not kw
                    return OrientationCls()
                assert atom_model_name == AtomicModel.OCR, atom_model_name
                if kw.get("is_seal"):
                    assert set(kw) == {"is_seal"}
                    trace["seal_model_requests"] = trace.get("seal_model_requests", 0) + 1
                    return SealOcr()
                trace.setdefault("atom_model_requests", []).append({k: v for k, v in kw.items() if k in ("lang",)})
                return ocr_by_lang[kw["lang"]] if langs else ocr
        reg.AtomModelSingleton = Registry
        ba.AtomModelSingleton = Registry
        sys.modules["rapid_doc.backend.pipeline.analyze_utils"].AtomModelSingleton = Registry

        class Model:
            device = "cpu"
            layout_model = RecordingLayout(trace, dets)
            ocr_model = ocr
            formula_model = RecordingFormula(trace)
            table_model = None

        class Manager:
            def get_model(self, **kw):
                return Model()
        ocr_cfg = {"use_det_mode": "ocr", "Det.rec_batch_num": 3, "seal_enable": False}
        if seal:
            del ocr_cfg["seal_enable"]       # the reference's default: True
        analyzer = ba.BatchAnalyze(Manager(), batch_ratio=1, formula_enable=formula_enable, table_enable=False,
                                   layout_config={"batch_num": 2}, ocr_config=ocr_cfg,
                                   formula_config={"formula_level": formula_level, "batch_num": 4, "bbox_expand_px": 2},
                                   checkbox_config={"checkbox_enable": True} if seed == 7 else None)
        ba.clean_vram = lambda *a, **k: None
        from PIL import Image
        inputs = [(Image.fromarray(p), 2.0, True, (langs[i] if langs else "ch"), {}) for i, p in enumerate(pages)]
        out = analyzer(inputs)
        os.environ.pop("USE_DOC_ORIENTATION_CLASSIFY", None)
        if rotate_labels:
            assert [pd.get("rotate_label") for *_x, pd in inputs] == rotate_labels

        def clean(o):
            if isinstance(o, dict):
                return {k: clean(v) for k, v in o.items()}
            if isinstance(o, (list, tuple)):
                return [clean(v) for v in o]
            if isinstance(o, (np.floating, np.integer)):
                return o.item()
            return o
        fixture = {"seed": seed, "page_langs": langs, "page_ids": page_ids, "page_hw": [H, W], "formula_enable": formula_enable, "formula_level": formula_level,
                   "ocr_config": ocr_cfg, "layout_batch_num": 2, "formula_batch_num": 4, "layout_dets": dets,
                   "checkbox_enable": seed == 7, "rotate_labels": rotate_labels, "input_rot90": 1 if sideways else 0,
                   "trace": clean(trace), "output": clean(out)}
        (HERE / f"analyze_trace_seed{seed}.json").write_text(json.dumps(fixture))
        n_spans = [sum(1 for d in page if d["category_id"] in (15, 16)) for page in fixture["output"]]
        print(f"analyze trace seed {seed}: {n_pages} pages, det calls {[(c['shapes'][0][:2], len(c['shapes']), c['batch_size']) for c in trace['det_calls']]}, "
              f"rec crops {[len(c['shapes']) for c in trace['rec_calls']]}, formula crops {[len(c['shapes']) for c in trace['formula_calls']]}, spans per page {n_spans}")


if __name__ == "__main__":
    main()
